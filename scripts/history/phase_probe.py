"""Developer probe: where one FromMel call's host-to-host time goes (create / upload / run / download / destroy)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import gomel_amd
K = 192
for F, I in ((342, 2), (342, 60), (39, 2)):
    m = gomel_amd.NewMel()
    m.MelFmax, m.Window, m.NumMels, m.Resolut, m.GriffinLimIterations = 16000, 1280, 192, 4096, I
    e = m.engine()
    mel = np.random.default_rng(0).uniform(np.log(1e-5), np.log(50.0), size=(F * K, 2))
    out = [np.zeros(e.out_len(len(mel)))]
    acc = np.zeros(5)
    R = 20
    for r in range(R + 3):
        t0 = time.perf_counter(); b = gomel_amd.Batch(e, [F * K])
        t1 = time.perf_counter(); b.upload([mel], None, [1])
        t2 = time.perf_counter(); b.run(); b.sync()
        t3 = time.perf_counter(); b.download(out)
        t4 = time.perf_counter(); b.close()
        t5 = time.perf_counter()
        if r >= 3: acc += np.diff([t0, t1, t2, t3, t4, t5])
    print("F %d I %d: create %.3f upload %.3f run %.3f download %.3f destroy %.3f ms" % ((F, I) + tuple(acc / R * 1e3)), flush=True)
    e.close()
