"""Developer probe: host-to-host latency of ONE FromMel call (what the Go API does per utterance) against the device time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import gomel_amd
K = 192
for F, I in ((342, 60), (342, 2), (39, 2), (1000, 60)):
    m = gomel_amd.NewMel()
    m.MelFmax, m.Window, m.NumMels, m.Resolut, m.GriffinLimIterations = 16000, 1280, 192, 4096, I
    e = m.engine()
    mel = np.random.default_rng(0).uniform(np.log(1e-5), np.log(50.0), size=(F * K, 2))
    for _ in range(3): e.from_mel(mel, seed=1)
    t = time.perf_counter()
    for _ in range(20): e.from_mel(mel, seed=1)
    one = (time.perf_counter() - t) / 20 * 1e3
    b = gomel_amd.Batch(e, [F * K]); b.upload([mel], None, [1])
    for _ in range(3): b.run()
    b.sync()
    t = time.perf_counter()
    for _ in range(20): b.run()
    b.sync()
    wall = (time.perf_counter() - t) / 20 * 1e3
    tm = b.run_timed()
    print("F %d I %d: from_mel host-to-host %.3f ms | resident run %.3f ms | device total %.3f iter %.3f dom %.3f path %s" % (
        F, I, one, wall, tm["total_ms"], tm["iter_ms"], tm["dom_ms"], tm["dom_kernel"]), flush=True)
    b.close(); e.close()
