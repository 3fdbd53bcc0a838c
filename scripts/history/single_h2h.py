"""One ToWavPng utterance (BASELINE configs[1]) through gomel_from_mel_batch, host buffers in and out, N times: the latency a
caller of mel.FromMel sees.  Under `rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --stats` it shows where the
host-side time goes."""
import sys, time
import numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gomel_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
pinned = len(sys.argv) > 2 and sys.argv[2] == "pinned"
m = gomel_amd.NewMel()
m.NumMels, m.MelFmax, m.Window, m.Resolut, m.GriffinLimIterations = 192, 16000.0, 1280, 4096, 60
e = m.engine()
F = 342
mel = np.random.default_rng(1).uniform(np.log(1e-5), np.log(50.0), size=(F * 192, 2))
out = None
if pinned:
    hm = gomel_amd.mel.host_alloc(mel.shape); hm[:] = mel; mel = hm
    out = [gomel_amd.mel.host_alloc(4096 + (F - 1) * 1280)]
for _ in range(5): e.from_mel_batch([mel], None, [3], outs=out)
t0 = time.perf_counter()
for _ in range(n): e.from_mel_batch([mel], None, [3], outs=out)
dt = (time.perf_counter() - t0) / n
print("host-to-host %.3f ms per utterance (%s buffers), real-time factor %.0f" % (dt * 1e3, "pinned" if pinned else "pageable", 10.0 / dt))
