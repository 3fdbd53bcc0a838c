import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import gomel_amd
F, K = 342, 192
mel = np.random.default_rng(0).uniform(np.log(1e-5), np.log(50.0), size=(F * K, 2))
for U in (1, 2, 3, 4, 6, 8, 16, 32):
    res = []
    for path in (2, 1):
        m = gomel_amd.NewMel()
        m.MelFmax, m.Window, m.NumMels, m.Resolut, m.GriffinLimIterations = 16000, 1280, 192, 4096, 60
        e = m.engine(); e.set_path(path)
        b = gomel_amd.Batch(e, [F * K] * U); b.upload([mel] * U, None, list(range(U)))
        for _ in range(3): b.run()
        b.sync()
        t = time.perf_counter()
        for _ in range(10): b.run()
        b.sync()
        res.append((time.perf_counter() - t) / 10 * 1e3)
        b.close()
    print("utts", U, "frames", U * F, "fused %.3f ms generic %.3f ms" % tuple(res), "-> fused faster" if res[0] < res[1] else "-> generic faster")
