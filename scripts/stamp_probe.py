"""Diagnostic: where a frame's cycles go inside k_gl_fused.  Needs a -DGOMEL_STAMPS build of the library:
   python3 -m gomel_amd.build --force -o abtest/lib_stamps.so -DGOMEL_STAMPS
usage: stamp_probe.py abtest/lib_stamps.so [T|D|S]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gomel_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import numpy as np
import gomel_amd
geom = sys.argv[2] if len(sys.argv) > 2 else "T"
G = {"T": dict(NumMels=192, MelFmax=16000.0, Window=1280, Resolut=4096), "S": dict(NumMels=80, MelFmax=8000.0, Window=256, Resolut=1024),
     "D": dict(NumMels=160, MelFmax=8000.0, Window=256, Resolut=2048)}[geom]
U, F, I = 256, 400, 6
m = gomel_amd.NewMel()
for k, v in dict(G, GriffinLimIterations=I).items(): setattr(m, k, v)
e = m.engine(); e.set_path(2)
if os.environ.get("GOMEL_FP64_ITERS"):   # -1: the fp64 fused kernel (its stamps use the same slots; in its interleaved passes slot 0 / 3 is the butterfly alone, slot 1 / 4 twiddles + writes + sync)
    e.set_precision_schedule(int(os.environ["GOMEL_FP64_ITERS"]))
base = np.random.default_rng(0).uniform(np.log(1e-5), np.log(50.0), size=(F * m.NumMels, 2))
b = gomel_amd.Batch(e, [F * m.NumMels] * U); b.upload([base] * U, None, list(range(U)))
L = _lib.lib()
L.gomel_debug_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
b.run(); b.sync()
L.gomel_debug_stamps(None, 1)
tm = b.run_timed()
acc = (ctypes.c_ulonglong * 21)()
L.gomel_debug_stamps(acc, 1)
a = np.array(list(acc), dtype=np.float64)
names = ["fwd pass1 (dft+tw)", "fwd write A + sync", "fwd read A + sync", "fwd pass2 (dft+tw)", "fwd write B + sync", "fwd read B + sync", "fwd pass3 (2 dft8)",
         "spectral step", "inv pass3 (2 dft8)", "inv write B + sync", "inv read B + tw2 + sync", "inv dft R2 + write A + sync", "inv read A + tw1 + sync",
         "inv pass1 dft", "OLA + stores", "top: slide window, issue next rows' loads", "top: issue magnitude loads", "top: window reads + multiply",
         "wait for next rows + combine", "-"]
tot = a[:20].sum()
waves_frames = a[20]   # sum over waves of g_max
print(tm)
print("cycles per wave-frame: %.0f" % (tot / waves_frames))
for k in [16, 17, 15, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 18, 14]:
    print("  %-44s %7.0f cycles  %5.1f %%" % (names[k], a[k] / waves_frames, 100 * a[k] / tot))
b.close()
