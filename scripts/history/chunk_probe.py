"""Host-to-host tuning probe (cfg3): resident rate vs segment length, host-to-host rate vs number of chunks."""
import os, sys, time, math, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np, gomel_amd
    mode, U = sys.argv[2], int(sys.argv[3])
    F, K = 400, 192
    m = gomel_amd.NewMel()
    for k, v in dict(NumMels=192, MelFmax=16000.0, Window=1280, Resolut=4096, GriffinLimIterations=60).items(): setattr(m, k, v)
    e = m.engine()
    mels = [np.random.default_rng(u).uniform(math.log(1e-5), math.log(50.0), size=(F * K, 2)) for u in range(U)]
    if mode == "resident":
        b = gomel_amd.Batch(e, [F * K] * U); b.upload(mels, None, list(range(U))); b.run(); b.sync()
        t = time.perf_counter()
        for _ in range(5): b.run()
        b.sync(); dt = (time.perf_counter() - t) / 5
        tm = b.run_timed()
        print("resident U=%d SEG=%s: %.2f ms/step  %.3f M frames/s  kernel %.4f ms/launch" % (U, os.environ.get("GOMEL_SEG_FRAMES", "auto"), dt * 1e3, U * F / dt / 1e6, tm["dom_ms"] / tm["iter_launches"]))
    else:
        pm = []
        for s in mels:
            a = gomel_amd.host_alloc((F * K, 2)); a[:] = s; pm.append(a)
        outs = [gomel_amd.host_alloc(e.out_len(F * K)) for _ in range(U)]
        for o in outs: o[:] = 0
        e.from_mel_batch(pm, None, list(range(U)), outs)
        t = time.perf_counter()
        for _ in range(5): e.from_mel_batch(pm, None, list(range(U)), outs)
        dt = (time.perf_counter() - t) / 5
        print("host-to-host U=%d CHUNKS=%s SEG=%s: %.2f ms/step  %.3f M frames/s" % (U, os.environ.get("GOMEL_CHUNKS", "auto"), os.environ.get("GOMEL_SEG_FRAMES", "auto"), dt * 1e3, U * F / dt / 1e6))
    sys.exit(0)
def run(mode, U, **env):
    e = dict(os.environ, **{k: str(v) for k, v in env.items()})
    r = subprocess.run([sys.executable, __file__, "child", mode, str(U)], env=e, capture_output=True, text=True)
    print((r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1], flush=True)
for U in (256, 32): run("resident", U)
for ch in (0, 8, 5): run("h2h", 256, **({"GOMEL_CHUNKS": ch} if ch else {}))
for U in (64, 128, 512): run("h2h", U)
run("h2h", 1024)
