import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import gomel_amd, gomel_oracle as go
def rms(a): return float(np.sqrt(np.mean(a * a)))
F, K, I = 400, 192, 60
fields = dict(NumMels=192, MelFmax=16000.0, Window=1280, Resolut=4096, GriffinLimIterations=I)
U = int(sys.argv[1]) if len(sys.argv) > 1 else 256
mels = [np.random.default_rng(u).uniform(math.log(1e-5), math.log(50.0), size=(F * K, 2)) for u in range(U)]
seeds = [1000 + u for u in range(U)]
m = gomel_amd.NewMel()
for k, v in fields.items(): setattr(m, k, v)
e = m.engine()
outs = e.from_mel_batch(mels, None, seeds)
cfg = go.MelConfig(**fields); L = go.out_len(cfg, F * K)
for u in (0, 3, 17, 100, 200, 255):
    if u >= U: continue
    want = go.from_mel(cfg, mels[u].copy(), go.x0_splitmix(seeds[u], L))
    d = outs[u] - want
    fr = np.abs(d[:(L // 1280) * 1280].reshape(-1, 1280)).max(axis=1)
    worst = np.argsort(fr)[-3:]
    print("utt", u, "rms vs oracle %.3e" % rms(d), "max %.3e" % np.abs(d).max(), "worst hop-blocks", worst.tolist(), ["%.1e" % fr[i] for i in worst], flush=True)
