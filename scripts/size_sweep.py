"""Resident time of one call against the number of utterances in it (342-frame utterances of the towav geometry, 60 iterations): where
the kernel paths hand over (persistent -> generic pair -> fused) in each precision mode.  usage: size_sweep.py [GEOM] [frames] [path] [U,U,...]"""
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import gomel_amd

GEOMS = {"T": dict(NumMels=192, MelFmax=16000.0, Window=1280, Resolut=4096), "S": dict(NumMels=80, MelFmax=8000.0, Window=256, Resolut=1024),
         "D": dict(NumMels=160, MelFmax=8000.0, Window=256, Resolut=2048), "Q": dict(NumMels=64, MelFmax=8000.0, Window=128, Resolut=512),
         "E": dict(NumMels=40, MelFmax=4000.0, Window=64, Resolut=256)}
geom = sys.argv[1] if len(sys.argv) > 1 else "T"
F = int(sys.argv[2]) if len(sys.argv) > 2 else 342
PATH = int(sys.argv[3]) if len(sys.argv) > 3 else 0      # gomel_set_path: 0 the library's choice, 1 generic kernels, 2 fused kernels
US = [int(u) for u in sys.argv[4].split(",")] if len(sys.argv) > 4 else [1, 2, 3, 4, 5, 6, 8, 12, 16, 24, 32, 64]
fields = dict(GEOMS[geom], GriffinLimIterations=60)
K = fields["NumMels"]
print("# %s, %d frames per utterance, 60 iterations; ms per call resident (median of 7), frames/s, kernel path" % (geom, F))
for mode, sched in (("default", "auto"), ("fp32", 0), ("fp64", -1)):
    m = gomel_amd.NewMel()
    for k, v in fields.items():
        setattr(m, k, v)
    e = m.engine()
    e.set_precision_schedule(sched)
    if PATH:
        e.set_path(PATH)
    for U in US:
        mels = [np.random.default_rng(u).uniform(math.log(1e-5), math.log(50.0), size=(F * K, 2)) for u in range(U)]
        b = gomel_amd.Batch(e, [F * K] * U)
        b.upload(mels, None, list(range(U)))
        ts = []
        info = None
        for r in range(9):
            info = b.run_timed()
            ts.append(info["total_ms"])
        b.close()
        t = float(np.median(ts[2:]))
        print("%-7s U=%3d frames=%6d: %.3f ms  %.3f M frames/s  %s (fp64 launches %d, other %d)" % (mode, U, U * F, t, U * F / t / 1e3, info["dom_kernel"],
                                                                                                  info["iter64_launches"], info["iter_launches"]), flush=True)
    e.close()
