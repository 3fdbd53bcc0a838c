"""A lone synchronous caller in the DEFAULT precision mode: the chunk plan of gomel_from_mel_batch (GOMEL_CHUNKS / GOMEL_CHUNK_SHARES, read per call) against
the built-in one.  256 x 400 frames, towav geometry, pinned buffers.  usage: chunk_sweep_default.py [mode: default|fp32|fp64]"""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import gomel_amd
mode = sys.argv[1] if len(sys.argv) > 1 else "default"
m = gomel_amd.NewMel()
m.NumMels, m.MelFmax, m.Window, m.Resolut, m.GriffinLimIterations = 192, 16000.0, 1280, 4096, 60
e = m.engine()
e.set_precision_schedule({"default": "auto", "fp32": 0, "fp64": -1}[mode])
U, F, K = 256, 400, 192
mels, outs = [], []
for u in range(U):
    a = gomel_amd.host_alloc((F * K, 2)); a[:] = np.random.default_rng(u).uniform(math.log(1e-5), math.log(50.0), size=(F * K, 2)); mels.append(a)
    o = gomel_amd.host_alloc(e.out_len(F * K)); o[:] = 0; outs.append(o)
bound = e.bind_buffers(mels, outs)
seeds = list(range(U))
plans = [None, "3", "4", "5", "6", "1,2,2,1", "1,2,2,2,1", "1,2,2,2,2,1", "1,3,3,3,1", "2,3,3,2"]
for rnd in range(2):
    for p in plans:
        os.environ.pop("GOMEL_CHUNKS", None); os.environ.pop("GOMEL_CHUNK_SHARES", None)
        if p is not None:
            os.environ["GOMEL_CHUNK_SHARES" if "," in p else "GOMEL_CHUNKS"] = p
        e.from_mel_batch_bound(bound, seeds)
        t = time.perf_counter()
        for _ in range(4):
            e.from_mel_batch_bound(bound, seeds)
        dt = (time.perf_counter() - t) / 4
        print("%s plan %-14s %.2f ms per call  %.3f M frames/s" % (mode, p or "built-in", dt * 1e3, U * F / dt / 1e6), flush=True)
