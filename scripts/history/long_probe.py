"""Developer probe: one very long utterance (hours of audio) through the fused and the generic kernels -- index-range / planning check."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import gomel_amd
def rms(a): return float(np.sqrt(np.mean(a * a)))
for N, hop, K, F in ((1024, 256, 80, 300000), (4096, 1280, 192, 120000)):
    outs = {}
    mel = np.random.default_rng(1).uniform(np.log(1e-5), np.log(50.0), size=(F * K, 2))
    for path in (2, 1):
        m = gomel_amd.NewMel()
        m.NumMels, m.MelFmax, m.Window, m.Resolut, m.GriffinLimIterations = K, 8000.0 if N == 1024 else 16000.0, hop, N, 2
        e = m.engine(); e.set_path(path)
        t = time.perf_counter()
        outs[path] = e.from_mel(mel, seed=5)
        print("N %d F %d path %d: %.2f s, %.1f h of audio at 22.05/44.1 kHz, rms %.4f" % (N, F, path, time.perf_counter() - t, F * hop / (22050 if N == 1024 else 44100) / 3600, rms(outs[path])), flush=True)
        e.close()
    d = outs[2] - outs[1]
    print("   fused vs generic rms %.3e, max %.3e, finite %s" % (rms(d), float(np.abs(d).max()), bool(np.all(np.isfinite(outs[2])))), flush=True)
