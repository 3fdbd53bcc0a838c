"""A/B of k_gl_fused_any against k_gl_fused at a hop both take (GOMEL_FORCE_ANY=1 selects the any-hop kernel), and of builds with
parts of the any-hop kernel compiled out (timing only: their results are wrong)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gomel_amd import _lib
if os.environ.get("GOMEL_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["GOMEL_LIB"])
import numpy as np
import gomel_amd
for N, hop, K in ((2048, 256, 80), (4096, 1280, 192), (1024, 256, 80)):
    m = gomel_amd.NewMel()
    m.NumMels, m.MelFmax, m.Window, m.Resolut, m.GriffinLimIterations = K, 8000.0, hop, N, 20
    e = m.engine(); e.set_path(2)
    base = np.random.default_rng(0).uniform(np.log(1e-5), np.log(50.0), size=(400 * K, 2))
    b = gomel_amd.Batch(e, [400 * K] * 256); b.upload([base] * 256, None, list(range(256)))
    for _ in range(2): b.run_timed()
    tm = b.run_timed()
    print("N %d hop %d: path %d  %.4f ms per iteration" % (N, hop, tm["fast_path"], tm["dom_ms"] / 20), flush=True)
    b.close(); e.close()
