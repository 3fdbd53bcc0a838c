"""Precision schedule on the GPU: config 3 / 4 full size, fp64 for the first K iterations then fp32, each against the all-fp64 result
(which tests hold to 1e-9 of the oracle).  usage: sched_probe.py [T|S] [K ...]     prints per-K error statistics and resident rates."""
import os, sys, math, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import gomel_amd
geom = sys.argv[1] if len(sys.argv) > 1 else "T"
KS = [int(a) for a in sys.argv[2:]] or [0, 20, 30, 40]
G = {"T": dict(NumMels=192, MelFmax=16000.0, Window=1280, Resolut=4096), "S": dict(NumMels=80, MelFmax=8000.0, Window=256, Resolut=1024)}[geom]
U, F, I = int(os.environ.get("UTTS", 256)), 400, 60
m = gomel_amd.NewMel()
for k, v in dict(G, GriffinLimIterations=I).items(): setattr(m, k, v)
e = m.engine()
K = m.NumMels
first = int(os.environ.get("FIRST", 0))
mels = [np.random.default_rng(first + u).uniform(math.log(1e-5), math.log(50.0), size=(F * K, 2)) for u in range(U)]
seeds = [1000 + first + u for u in range(U)]
def run(sched):
    e.set_precision_schedule(sched)
    b = gomel_amd.Batch(e, [F * K] * U); b.upload(mels, None, seeds); b.run(); b.sync()
    t = time.perf_counter(); b.run(); b.run(); b.sync(); dt = (time.perf_counter() - t) / 2
    tm = b.run_timed(); out = b.download(); b.close()
    return out, dt, tm
ref, dt, tm = run(-1)
print("%s all fp64: %.2f ms/step %.3f M frames/s, fp64 kernel %.4f ms/launch" % (geom, dt * 1e3, U * F / dt / 1e6, tm["dom64_ms"] / max(1, tm["iter64_launches"])), flush=True)
for k in KS:
    out, dt, tm = run(k)
    err = np.array([float(np.sqrt(np.mean((a - r) ** 2))) for a, r in zip(out, ref)])
    n32 = tm["iter_launches"] - tm["iter64_launches"]
    print("K=%2d: %.2f ms/step %.3f M frames/s | fp64 launches %d x %.4f ms, fp32 %d x %.4f ms | err median %.2e p90 %.2e p99 %.2e max %.2e n>1e-4 %d n>3e-5 %d worst %s" % (
        k, dt * 1e3, U * F / dt / 1e6, tm["iter64_launches"], tm["dom64_ms"] / max(1, tm["iter64_launches"]), n32, (tm["dom_ms"] - tm["dom64_ms"]) / max(1, n32),
        np.median(err), np.percentile(err, 90), np.percentile(err, 99), err.max(), (err > 1e-4).sum(), (err > 3e-5).sum(), np.argsort(-err)[:3].tolist()), flush=True)
