"""Resident rate of the DEFAULT precision mode (and fp32 / fp64 throughout) at hops on and off the Resolut/16 grid: 256 x 400 frames, 60 iterations.
The off-grid hops run the any-hop fused kernels (k_gl_fused_any_d / k_gl_fused_any)."""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import gomel_amd
U, F, I = 256, 400, 60
for N, hop, K in ((2048, 256, 160), (2048, 275, 80), (2048, 300, 80), (1024, 256, 80), (1024, 300, 80), (4096, 1280, 192), (4096, 882, 192), (4096, 441, 192)):
    m = gomel_amd.NewMel()
    m.NumMels, m.MelFmax, m.Window, m.Resolut, m.GriffinLimIterations = K, 8000.0 if N < 4096 else 16000.0, hop, N, I
    e = m.engine()
    base = np.random.default_rng(0).uniform(math.log(1e-5), math.log(50.0), size=(F * K, 2))
    row = "N %4d hop %4d:" % (N, hop)
    for name, sched in (("default", "auto"), ("fp32", 0), ("fp64", -1)):
        e.set_precision_schedule(sched)
        b = gomel_amd.Batch(e, [F * K] * U)
        b.upload([base] * U, None, list(range(U)))
        b.run(); b.sync()
        t = time.perf_counter()
        for _ in range(3): b.run()
        b.sync()
        dt = (time.perf_counter() - t) / 3
        tm = b.run_timed()
        n64 = tm["iter64_launches"]; n32 = tm["iter_launches"] - n64
        row += "  %s %.2f M frames/s (%s%s)" % (name, U * F / dt / 1e6, "fp64 %.3f ms " % (tm["dom64_ms"] / n64) if n64 else "", "fp32 %.3f ms" % ((tm["dom_ms"] - tm["dom64_ms"]) / n32) if n32 else "")
        b.close()
    print(row, flush=True)
    e.close()
