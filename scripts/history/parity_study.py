"""Per-utterance parity at BASELINE config 3: fp32 fused path vs the fp64 device path (same seeds), the worst few also vs the C oracle.
Writes gpurun_out/parity_study.npz (errors per utterance, per-hop-block error profile of the worst ones)."""
import os, sys, math, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import gomel_amd, gomel_oracle as go, oracle_c
U, F, K, I = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 400, 192, int(sys.argv[2]) if len(sys.argv) > 2 else 60
fields = dict(NumMels=192, MelFmax=16000.0, Window=1280, Resolut=4096, GriffinLimIterations=I)
mels = [np.random.default_rng(u).uniform(math.log(1e-5), math.log(50.0), size=(F * K, 2)) for u in range(U)]
seeds = [1000 + u for u in range(U)]
m = gomel_amd.NewMel()
for k, v in fields.items(): setattr(m, k, v)
eng = m.engine()
def run(bits, path=0):
    eng.set_precision(bits); eng.set_path(path)
    b = gomel_amd.Batch(eng, [F * K] * U); b.upload(mels, None, seeds); b.run(); o = b.download(); b.close()
    return o
o32 = run(32); o32g = run(32, 1); o64 = run(64)
e = np.array([float(np.sqrt(np.mean((a - w) ** 2))) for a, w in zip(o32, o64)])
eg = np.array([float(np.sqrt(np.mean((a - w) ** 2))) for a, w in zip(o32g, o64)])
e22 = np.array([float(np.sqrt(np.mean((a - w) ** 2))) for a, w in zip(o32, o32g)])
def summ(name, e): print("%s: median %.2e p90 %.2e p99 %.2e max %.2e (#>1e-4: %d, #>3e-5: %d, #>1e-5: %d)" % (name, np.median(e), np.percentile(e, 90), np.percentile(e, 99), e.max(), int((e > 1e-4).sum()), int((e > 3e-5).sum()), int((e > 1e-5).sum())))
summ("fused fp32 vs device fp64", e); summ("generic fp32 vs device fp64", eg); summ("fused vs generic fp32", e22)
worst = np.argsort(-e)[:8]
print("worst utterances (fused):", worst.tolist(), e[worst].tolist()); print("their generic-path errors:", eg[worst].tolist())
cfg = go.MelConfig(**fields); L = go.out_len(cfg, F * K)
co = oracle_c.load()
t = time.time()
want, used = co.from_mel_batch(cfg, [mels[u] for u in worst], [go.x0_splitmix(seeds[u], L) for u in worst], threads=16)
print("oracle: %d threads, %.0f s" % (used, time.time() - t))
for j, u in enumerate(worst):
    print("utt %d: fp64-device vs oracle %.2e, fused vs oracle %.2e" % (u, np.sqrt(np.mean((o64[u] - want[j]) ** 2)), np.sqrt(np.mean((o32[u] - want[j]) ** 2))))
hop = 1280
prof = np.array([[float(np.sqrt(np.mean((o32[u][i * hop:(i + 1) * hop] - o64[u][i * hop:(i + 1) * hop]) ** 2))) for i in range(len(o64[u]) // hop)] for u in worst])
np.savez(os.path.join(ROOT, "gpurun_out", "parity_study.npz"), e=e, eg=eg, e22=e22, worst=worst, prof=prof)
for j, u in enumerate(worst[:4]):
    p = prof[j]; k = np.argsort(-p)[:6]
    print("utt %d hop-block error profile: top blocks %s -> %s ; median block %.2e" % (u, k.tolist(), ["%.1e" % p[i] for i in k], np.median(p)))
