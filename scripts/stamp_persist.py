"""Diagnostic: where an iteration's cycles go inside k_gl_persist.  Needs a -DGOMEL_STAMPS build:
   python3 -m gomel_amd.build --force -o abtest/lib_stamps.so -DGOMEL_STAMPS
usage: stamp_persist.py abtest/lib_stamps.so"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gomel_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import numpy as np
import gomel_amd
m = gomel_amd.NewMel()
I = 60
G = {"T": dict(NumMels=192, MelFmax=16000.0, Window=1280, Resolut=4096), "S": dict(NumMels=80, MelFmax=8000.0, Window=256, Resolut=1024)}[sys.argv[2] if len(sys.argv) > 2 else "T"]
F = int(sys.argv[3]) if len(sys.argv) > 3 else 342
for k, v in dict(G, GriffinLimIterations=I).items(): setattr(m, k, v)
e = m.engine()
base = np.random.default_rng(0).uniform(np.log(1e-5), np.log(50.0), size=(F * m.NumMels, 2))
b = gomel_amd.Batch(e, [F * m.NumMels]); b.upload([base], None, [0])
L = _lib.lib()
L.gomel_debug_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
b.run(); b.sync()
L.gomel_debug_stamps(None, 1)
tm = b.run_timed()
acc = (ctypes.c_ulonglong * 21)()
L.gomel_debug_stamps(acc, 1)
a = np.array(list(acc), dtype=np.float64)
names = ["gather: sc1 loads until every neighbour's quads are fresh, sums", "DPP + sentinel stores", "(load rounds per gather, not ticks)", "window + forward FFT", "spectral step", "inverse FFT",
         "window + drain + 16-B sc1 stores", "loop"]
rounds = a[2]; a[2] = 0; tot = a[:8].sum(); wi = a[20]
print(tm)
print("load rounds per wave-iteration: %.2f" % (rounds / wi))
print("ticks per wave-iteration: %.0f  (kernel %.3f us per iteration)" % (tot / wi, tm["dom_ms"] * 1e3 / I))
for k in range(8):
    print("  %-40s %7.0f  %5.1f %%" % (names[k], a[k] / wi, 100 * a[k] / tot))
b.close()
