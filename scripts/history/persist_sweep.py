"""k_gl_persist: time per iteration against the number of frames (one utterance, Resolut 4096, hop 1280)."""
import sys
import numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gomel_amd
I = 60
for N, hop, K, fmax, Fs in ((4096, 1280, 192, 16000.0, (16, 64, 128, 200, 256, 257, 300, 342, 400, 512, 684, 1024)), (1024, 256, 80, 8000.0, (64, 256, 512, 1024, 2048, 4096))):
    m = gomel_amd.NewMel()
    m.NumMels, m.MelFmax, m.Window, m.Resolut, m.GriffinLimIterations = K, fmax, hop, N, I
    e = m.engine()
    for F in Fs:
        mel = np.random.default_rng(F).uniform(np.log(1e-5), np.log(50.0), size=(F * K, 2))
        b = gomel_amd.Batch(e, [F * K]); b.upload([mel], None, [0])
        for _ in range(3): b.run_timed()
        d = sorted(b.run_timed()["dom_ms"] for _ in range(15))
        tm = b.run_timed()
        print("N %d frames %4d: path %d  kernel %.3f ms (min %.3f)  %.2f us per iteration" % (N, F, tm["fast_path"], d[7], d[0], d[7] * 1e3 / I), flush=True)
        b.close()
    e.close()
