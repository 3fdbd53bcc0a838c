"""k_gl_fused: launch time against the number of frames per launch (U utterances of 400 frames): the fixed cost per launch that
the chunked host-to-host pipeline pays 60 times per chunk."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gomel_amd import _lib
if os.environ.get("GOMEL_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["GOMEL_LIB"])      # A/B another build
import gomel_amd
G = {"T": dict(NumMels=192, MelFmax=16000.0, Window=1280, Resolut=4096), "S": dict(NumMels=80, MelFmax=8000.0, Window=256, Resolut=1024)}
for geom in ("T", "S"):
    m = gomel_amd.NewMel()
    for k, v in dict(G[geom], GriffinLimIterations=60).items(): setattr(m, k, v)
    e = m.engine(); e.set_path(2)
    base = np.random.default_rng(0).uniform(np.log(1e-5), np.log(50.0), size=(400 * m.NumMels, 2))
    pts = []
    for U in (4, 8, 16, 17, 32, 33, 64, 128, 256):
        b = gomel_amd.Batch(e, [400 * m.NumMels] * U); b.upload([base] * U, None, list(range(U)))
        for _ in range(2): b.run_timed()
        d = sorted(b.run_timed()["dom_ms"] / 60 for _ in range(5))[2]
        pts.append((U * 400, d)); b.close()
        print("%s %6d frames per launch: %.4f ms  (%.2f ns per frame)" % (geom, U * 400, d, d * 1e6 / (U * 400)), flush=True)
    (f1, t1), (f2, t2) = pts[4], pts[-1]
    bb = (t2 - t1) / (f2 - f1); a = t1 - bb * f1
    print("%s: t = %.1f us + %.2f ns per frame (through 12 800 and 102 400 frames)" % (geom, a * 1e3, bb * 1e6))
    e.close()
