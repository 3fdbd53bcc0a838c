"""One synchronous gomel_from_mel_batch call over U utterances (default precision mode, pinned buffers): frames/s by call size."""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import gomel_amd
m = gomel_amd.NewMel()
m.NumMels, m.MelFmax, m.Window, m.Resolut, m.GriffinLimIterations = 192, 16000.0, 1280, 4096, 60
e = m.engine()
F, K = 400, 192
Umax = 1024
mels, outs = [], []
base = np.random.default_rng(0).uniform(math.log(1e-5), math.log(50.0), size=(F * K, 2))
for u in range(Umax):
    a = gomel_amd.host_alloc((F * K, 2)); a[:] = base; mels.append(a)
    o = gomel_amd.host_alloc(e.out_len(F * K)); o[:] = 0; outs.append(o)
for U in (128, 256, 512, 1024):
    bound = e.bind_buffers(mels[:U], outs[:U])
    seeds = list(range(U))
    for mode in (0, 1):
        e.set_call_mode(mode)
        e.from_mel_batch_bound(bound, seeds)
        t = time.perf_counter()
        reps = max(1, 1024 // U)
        for _ in range(reps):
            e.from_mel_batch_bound(bound, seeds)
        dt = (time.perf_counter() - t) / reps
        print("U=%4d call_mode %d (%s): %.1f ms per call, %.3f M frames/s, chunks %d" % (U, mode, "whole batch" if mode else "chunked", dt * 1e3, U * F / dt / 1e6, len(e.plan_chunks([F * K] * U)) - 1), flush=True)
