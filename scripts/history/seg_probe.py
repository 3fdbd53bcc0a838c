import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import gomel_amd, gomel_oracle as go
def rms(a): return float(np.sqrt(np.mean(a * a)))
F, K, I = 400, 192, int(sys.argv[1]) if len(sys.argv) > 1 else 60
fields = dict(NumMels=192, MelFmax=16000.0, Window=1280, Resolut=4096, GriffinLimIterations=I)
mel = np.random.default_rng(3).uniform(math.log(1e-5), math.log(50.0), size=(F * K, 2))
cfg = go.MelConfig(**fields)
L = go.out_len(cfg, len(mel)); x0 = go.x0_splitmix(1003, L)
want = go.from_mel(cfg, mel.copy(), x0)
res = {}
for name, env, path in (("gen", None, 1), ("seg3", "3", 0), ("seg25", "25", 0), ("seg64", "64", 0), ("seg400", "400", 0)):
    if env: os.environ["GOMEL_SEG_FRAMES"] = env
    m = gomel_amd.NewMel()
    for k, v in fields.items(): setattr(m, k, v)
    e = m.engine()
    if path: e.set_path(path)
    res[name] = e.from_mel(mel, x0=x0)
    print(name, "vs oracle %.3e" % rms(res[name] - want), flush=True)
for a in ("seg3", "seg64", "seg400", "gen"):
    print(a, "vs seg25 %.3e" % rms(res[a] - res["seg25"]))
