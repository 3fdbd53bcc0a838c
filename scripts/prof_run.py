"""Profiling target: one cfg3-sized (or smaller) batch, a few GL iterations, nothing else.  usage: prof_run.py [utts] [frames] [iters] [geom]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from gomel_amd import _lib
if os.environ.get("GOMEL_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["GOMEL_LIB"])      # A/B another build
import gomel_amd
U = int(sys.argv[1]) if len(sys.argv) > 1 else 256
F = int(sys.argv[2]) if len(sys.argv) > 2 else 400
I = int(sys.argv[3]) if len(sys.argv) > 3 else 6
geom = sys.argv[4] if len(sys.argv) > 4 else "T"
G = {"T": dict(NumMels=192, MelFmax=16000.0, Window=1280, Resolut=4096), "S": dict(NumMels=80, MelFmax=8000.0, Window=256, Resolut=1024),
     "D": dict(NumMels=160, MelFmax=8000.0, Window=256, Resolut=2048)}[geom]
m = gomel_amd.NewMel()
for k, v in dict(G, GriffinLimIterations=I).items():
    setattr(m, k, v)
e = m.engine()
if os.environ.get("GOMEL_PATH"):
    e.set_path(int(os.environ["GOMEL_PATH"]))
if os.environ.get("GOMEL_FP64_ITERS"):                            # precision schedule: -1 = fp64 throughout (k_gl_fused_d), k = first k iterations
    e.set_precision_schedule(int(os.environ["GOMEL_FP64_ITERS"]))
rng = np.random.default_rng(0)
base = rng.uniform(np.log(1e-5), np.log(50.0), size=(F * m.NumMels, 2))
b = gomel_amd.Batch(e, [F * m.NumMels] * U)
b.upload([base] * U, None, list(range(U)))
for _ in range(int(os.environ.get("GOMEL_RUNS", "0"))):            # calls in front of the timed one, back to back (steady state of a power-limited chip)
    b.run()
tm = b.run_timed()
print(tm)
b.close()
