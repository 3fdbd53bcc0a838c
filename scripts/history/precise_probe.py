"""64-bit mode: fused fp64 kernel vs the generic fp64 kernels (same inputs), timing and agreement."""
import os, sys, math, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from gomel_amd import _lib
if os.environ.get("GOMEL_LIB"): _lib.LIB_PATH = os.path.abspath(os.environ["GOMEL_LIB"])   # A/B another build
import gomel_amd
geom = sys.argv[1] if len(sys.argv) > 1 else "T"
G = {"T": dict(NumMels=192, MelFmax=16000.0, Window=1280, Resolut=4096), "S": dict(NumMels=80, MelFmax=8000.0, Window=256, Resolut=1024),
     "D": dict(NumMels=160, MelFmax=8000.0, Window=256, Resolut=2048)}[geom]
U, F, I = int(sys.argv[2]) if len(sys.argv) > 2 else 256, 400, int(sys.argv[3]) if len(sys.argv) > 3 else 60
m = gomel_amd.NewMel()
for k, v in dict(G, GriffinLimIterations=I).items(): setattr(m, k, v)
e = m.engine(); e.set_precision(64)
K = m.NumMels
mels = [np.random.default_rng(u).uniform(math.log(1e-5), math.log(50.0), size=(F * K, 2)) for u in range(U)]
res = {}
for path, name in ((1, "generic fp64"), (2, "fused fp64")):
    e.set_path(path)
    b = gomel_amd.Batch(e, [F * K] * U); b.upload(mels, None, list(range(U))); b.run(); b.sync()
    t = time.perf_counter(); b.run(); b.sync(); dt = time.perf_counter() - t
    tm = b.run_timed(); res[path] = b.download(); b.close()
    print("%s %-13s: %.2f ms/step  %.3f M frames/s  dominant kernel %.4f ms/launch (fast_path %d)" % (geom, name, dt * 1e3, U * F / dt / 1e6, tm["dom_ms"] / max(1, tm["iter_launches"]), tm["fast_path"]))
d = max(float(np.sqrt(np.mean((a - b) ** 2))) for a, b in zip(res[1], res[2]))
print("max rms difference fused vs generic fp64 over %d utterances: %.2e" % (U, d))
