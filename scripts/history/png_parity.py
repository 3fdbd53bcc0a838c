import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import gomel_amd, gomel_oracle as go
from gomel_amd import codec
def rms(a): return float(np.sqrt(np.mean(a * a)))
# (1) the PNG fixture (chirp + AM tone + noise, 39 frames); (2) a 10 s version of the same signal through ToMel -> 8-bit PNG quantisation
png = os.path.join(ROOT, "tests", "golden", "chirp_192.png")
buf, samples, sr = codec.loadpng(png, True)
sig = go.config2_signal(seconds=10.0)
cfg = go.towav_config(GriffinLimIterations=60)
long_mel = go.to_mel(cfg, sig)
img = go.dumpimage_pixels(long_mel, 192, True, 1290.0, 44100.0)
long_buf, _, _ = go.loadpng_pixels(img, True)
for name, b in (("fixture PNG (39 frames)", buf), ("10 s utterance via PNG (342 frames)", long_buf)):
    for I in (2, 60):
        cfg = go.towav_config(GriffinLimIterations=I)
        m = gomel_amd.NewMel(); m.MelFmax, m.Window, m.NumMels, m.Resolut, m.GriffinLimIterations = 16000, 1280, 192, 4096, I
        x0 = go.x0_splitmix(1, go.out_len(cfg, len(b)))
        want = go.from_mel(cfg, b.copy(), x0)
        for path in (2, 1):
            m.engine().set_path(path)
            got = m.FromMel(b.copy(), x0=x0)
            print(name, "I", I, "path", "fused" if path == 2 else "generic", "rms err %.2e" % rms(got - want), "signal rms %.3f" % rms(want))
