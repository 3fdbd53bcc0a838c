import os, sys, math, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import gomel_amd
U, F, K, I = 256, 400, 192, 60
mels = [np.random.default_rng(u).uniform(math.log(1e-5), math.log(50.0), size=(F * K, 2)) for u in range(U)]
seeds = [1000 + u for u in range(U)]
res = {}
for bits in (32, 64):
    m = gomel_amd.NewMel(); m.MelFmax, m.Window, m.NumMels, m.Resolut, m.GriffinLimIterations = 16000, 1280, 192, 4096, I
    e = m.engine(); e.set_precision(bits)
    b = gomel_amd.Batch(e, [F * K] * U); b.upload(mels, None, seeds)
    b.run(); b.sync()
    tm = b.run_timed()
    res[bits] = b.download(); b.close()
    print(bits, "bits: %.1f ms per batch -> %.2f M frames/s" % (tm["total_ms"], U * F / tm["total_ms"] / 1e3))
e = np.array([float(np.sqrt(np.mean((a - w) ** 2))) for a, w in zip(res[32], res[64])])
print("fp32 vs fp64 kernels, 256 utterances: median %.2e p90 %.2e p99 %.2e max %.2e (# > 1e-4: %d)" % (np.median(e), np.percentile(e, 90), np.percentile(e, 99), e.max(), int((e > 1e-4).sum())))
