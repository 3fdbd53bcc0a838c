import os, sys, time, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import gomel_amd
U, F, K = 256, 400, 192
m = gomel_amd.NewMel(); m.MelFmax, m.Window, m.NumMels, m.Resolut, m.GriffinLimIterations = 16000, 1280, 192, 4096, 60
e = m.engine()
mels = [np.random.default_rng(u).uniform(math.log(1e-5), math.log(50.0), size=(F * K, 2)) for u in range(U)]
outs = [np.zeros(4096 + (F - 1) * 1280) for _ in range(U)]
ref = None
for rep in range(4):
    t = time.perf_counter(); e.from_mel_batch(mels, None, list(range(U)), outs); dt = time.perf_counter() - t
    print("from_mel_batch %.1f ms -> %.0f frames/s" % (dt * 1e3, U * F / dt), "rms", float(np.sqrt(np.mean(outs[7] ** 2))))
    if ref is None: ref = [o.copy() for o in outs]
print("repeat identical:", all(np.array_equal(a, b) for a, b in zip(ref, outs)))
b = gomel_amd.Batch(e, [F * K] * U); b.upload(mels, None, list(range(U))); b.run(); one = b.download(); b.close()
print("max |chunked - single batch| rms:", max(float(np.sqrt(np.mean((a - c) ** 2))) for a, c in zip(one, outs)))
