import os, sys, time, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import gomel_amd
U, F, K = 256, 400, 192
m = gomel_amd.NewMel(); m.MelFmax, m.Window, m.NumMels, m.Resolut, m.GriffinLimIterations = 16000, 1280, 192, 4096, 60
e = m.engine()
mels = [np.random.default_rng(u).uniform(math.log(1e-5), math.log(50.0), size=(F * K, 2)) for u in range(U)]
outs_buf = [np.zeros(4096 + (F - 1) * 1280) for _ in range(U)]
for rep in range(3):
    t0 = time.perf_counter(); b = gomel_amd.Batch(e, [F * K] * U); t1 = time.perf_counter()
    b.upload(mels, None, list(range(U))); t2 = time.perf_counter()
    b.run(); b.sync(); t3 = time.perf_counter()
    outs = b.download(outs_buf); t4 = time.perf_counter()
    b.close(); t5 = time.perf_counter()
    print("create %.1f ms  upload %.1f ms (%.1f GB/s)  run %.1f ms  download %.1f ms (%.1f GB/s)  destroy %.1f ms  total %.1f ms -> %.0f frames/s" % (
        (t1-t0)*1e3, (t2-t1)*1e3, U*F*K*16/1e9/(t2-t1), (t3-t2)*1e3, (t4-t3)*1e3, sum(o.nbytes for o in outs)/1e9/(t4-t3), (t5-t4)*1e3, (t5-t0)*1e3, U*F/(t5-t0)))
