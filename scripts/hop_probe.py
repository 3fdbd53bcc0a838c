"""Throughput of the generic kernels (hops the fused kernel does not cover) next to the fused kernel at a nearby hop."""
import os, sys, math, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from gomel_amd import _lib
if os.environ.get("GOMEL_LIB"): _lib.LIB_PATH = os.path.abspath(os.environ["GOMEL_LIB"])   # A/B another build
import gomel_amd
def run(N, hop, K, U=256, F=400, I=20):
    PATH = int(os.environ.get("GOMEL_PATH", "0"))
    m = gomel_amd.NewMel()
    for k, v in dict(NumMels=K, MelFmax=8000.0, Window=hop, Resolut=N, GriffinLimIterations=I).items(): setattr(m, k, v)
    e = m.engine()
    if os.environ.get("GOMEL_PRECISION"):
        e.set_precision(int(os.environ["GOMEL_PRECISION"]))        # 64: the fp64 kernels (k_gl_fused_d / k_gl_fused_any_d / generic fp64)
    if PATH:
        try:
            e.set_path(PATH)
        except Exception:
            pass
    mels = [np.random.default_rng(u).uniform(math.log(1e-5), math.log(50.0), size=(F * K, 2)) for u in range(U)]
    b = gomel_amd.Batch(e, [F * K] * U); b.upload(mels, None, list(range(U))); b.run(); b.sync()
    t = time.perf_counter(); b.run(); b.sync(); dt = time.perf_counter() - t
    tm = b.run_timed(); b.close(); e.close()
    print("N %4d hop %4d: %.3f M frames/s (%d iterations)  path %s  dominant kernel %.4f ms/launch" % (N, hop, U * F / dt / 1e6 * I / 60.0 * 60 / I, I, ("generic", "fused", "persistent", "fused (any hop)")[tm["fast_path"]], tm["dom_ms"] / tm["iter_launches"]))
PATH = int(os.environ.get("GOMEL_PATH", "0"))
for N, hop, K in ((2048, 256, 80), (2048, 276, 80), (2048, 300, 80), (2048, 275, 80), (1024, 256, 80), (1024, 300, 80), (1024, 200, 80), (4096, 1280, 192), (4096, 882, 192),
                  (4096, 1000, 192), (4096, 441, 192), (512, 160, 64), (512, 200, 64)):
    run(N, hop, K)
