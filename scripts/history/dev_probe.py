"""Developer probe (not part of the product or the tests): error margins and a first timing at batch scale."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import gomel_amd, gomel_oracle as go

def rms(a): return float(np.sqrt(np.mean(a * a)))
G = {"T": dict(NumMels=192, MelFmax=16000.0, Window=1280, Resolut=4096), "S": dict(NumMels=80, MelFmax=8000.0, Window=256, Resolut=1024),
     "D": dict(NumMels=160, MelFmax=8000.0, Window=256, Resolut=2048)}
path = int(os.environ.get("GOMEL_PATH", "0"))
for g, F in (("T", 12), ("S", 24), ("D", 16)):
    for I in (2, 60):
        m = gomel_amd.NewMel()
        for k, v in dict(G[g], GriffinLimIterations=I).items(): setattr(m, k, v)
        cfg = go.MelConfig(**dict(G[g], GriffinLimIterations=I))
        mel = go.random_logmel(F, cfg.NumMels, seed=4); L = go.out_len(cfg, len(mel)); x0 = go.x0_splitmix(9, L)
        want = go.from_mel(cfg, mel.copy(), x0)
        e = m.engine(); e.set_path(path) if path else None
        got = e.from_mel(mel, x0=x0)
        print(g, "F", F, "I", I, "rms err %.3e" % rms(got - want), "rms sig %.3f" % rms(want), "max err %.3e" % np.abs(got - want).max(), flush=True)
# batch timing
for g, U, F in (("T", 256, 400), ("S", 256, 400)):
    m = gomel_amd.NewMel()
    for k, v in dict(G[g], GriffinLimIterations=60).items(): setattr(m, k, v)
    e = m.engine(); e.set_path(path) if path else None
    K = m.NumMels
    rng = np.random.default_rng(0)
    base = rng.uniform(np.log(1e-5), np.log(50.0), size=(F * K, 2))
    mels = [base] * U
    b = gomel_amd.Batch(e, [F * K] * U)
    t = time.time(); b.upload(mels, None, list(range(U))); tu = time.time() - t
    for rep in range(3):
        tm = b.run_timed()
        print(g, "batch", U, "x", F, "total %.2f ms iter %.2f ms" % (tm["total_ms"], tm["iter_ms"]), "frames/s %.0f" % (U * F / tm["total_ms"] * 1e3),
              "fast", tm["fast_path"], "upload %.2fs" % tu, flush=True)
    t = time.time(); outs = b.download(); print("download %.2fs" % (time.time() - t), "finite", all(np.isfinite(o).all() for o in outs[:4]), rms(outs[0]))
    b.close()
