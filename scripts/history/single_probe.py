import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import gomel_amd
F, K = 342, 192
for path in (0, 1):
    for seg in (None, "3", "6", "12"):
        if path == 1 and seg: continue
        if seg: os.environ["GOMEL_SEG_FRAMES"] = seg
        m = gomel_amd.NewMel()
        m.MelFmax, m.Window, m.NumMels, m.Resolut, m.GriffinLimIterations = 16000, 1280, 192, 4096, 60
        e = m.engine()
        if path: e.set_path(1)
        mel = np.random.default_rng(0).uniform(np.log(1e-5), np.log(50.0), size=(F * K, 2))
        b = gomel_amd.Batch(e, [F * K]); b.upload([mel], None, [1])
        for _ in range(3): b.run()
        b.sync()
        t = time.perf_counter()
        for _ in range(20): b.run()
        b.sync()
        wall = (time.perf_counter() - t) / 20 * 1e3
        tm = b.run_timed()
        print("path", path, "seg", seg, "wall %.3f ms/utt" % wall, "device total %.3f iter %.3f dom %.3f" % (tm["total_ms"], tm["iter_ms"], tm["dom_ms"]))
        b.close()
