"""Host-to-host config 3 call (256 x 400 frames, pinned fp64 buffers) a few times: the timeline a rocprofv3 --kernel-trace
--memory-copy-trace run of this script shows is what bench.py's `value` is made of."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gomel_amd
from gomel_amd.mel import host_alloc
U, F, K = 256, 400, 192
m = gomel_amd.NewMel()
m.NumMels, m.MelFmax, m.Window, m.Resolut, m.GriffinLimIterations = K, 16000.0, 1280, 4096, 60
e = m.engine()
rng = np.random.default_rng(0)
mel = host_alloc((U, F * K, 2)); mel[:] = rng.uniform(np.log(1e-5), np.log(50.0), size=(U, F * K, 2))
L = 4096 + (F - 1) * 1280
out = host_alloc((U, L))
mels = [mel[u] for u in range(U)]; outs = [out[u] for u in range(U)]
for _ in range(2): e.from_mel_batch(mels, None, list(range(U)), outs=outs)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
t0 = time.perf_counter()
for _ in range(n): e.from_mel_batch(mels, None, list(range(U)), outs=outs)
dt = (time.perf_counter() - t0) / n
print("host-to-host %.2f ms per call, %.3f M frames/s" % (dt * 1e3, U * F / dt / 1e6))
