"""Parity distribution at BASELINE config 3: every utterance of the batch vs the fp64 C restatement (same x0)."""
import os, sys, math, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import gomel_amd, gomel_oracle as go, oracle_c
U, F, K, I = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 400, 192, 60
fields = dict(NumMels=192, MelFmax=16000.0, Window=1280, Resolut=4096, GriffinLimIterations=I)
mels = [np.random.default_rng(u).uniform(math.log(1e-5), math.log(50.0), size=(F * K, 2)) for u in range(U)]
seeds = [1000 + u for u in range(U)]
m = gomel_amd.NewMel()
for k, v in fields.items(): setattr(m, k, v)
b = gomel_amd.Batch(m.engine(), [F * K] * U); b.upload(mels, None, seeds); b.run(); outs = b.download(); b.close()
cfg = go.MelConfig(**fields); L = go.out_len(cfg, F * K)
co = oracle_c.load()
t = time.time()
x0s = [go.x0_splitmix(s, L).astype(np.float32).astype(np.float64) for s in seeds]   # the device stores x0 in fp32
want, used = co.from_mel_batch(cfg, mels, x0s, threads=16)
print("oracle: %d threads, %.0f s" % (used, time.time() - t))
e = np.array([float(np.sqrt(np.mean((a - w) ** 2))) for a, w in zip(outs, want)])
s = np.array([float(np.sqrt(np.mean(w ** 2))) for w in want])
print("utterances", U, "signal rms %.3f" % s.mean())
print("rms error vs fp64 oracle: median %.2e  p90 %.2e  p99 %.2e  max %.2e  (# > 1e-4: %d)" % (np.median(e), np.percentile(e, 90), np.percentile(e, 99), e.max(), int((e > 1e-4).sum())))
