"""How often does a precision schedule leave an utterance beyond the 1e-4 bar?  N_BLOCKS x 256 random utterances (config 3 / config 4
inputs, seeds = global indices), each schedule against 64-bit mode on the GPU.  usage: sched_scale.py [T|S] [blocks] [K ...]"""
import os, sys, math, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import gomel_amd
geom = sys.argv[1] if len(sys.argv) > 1 else "T"
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 8
KS = [int(a) for a in sys.argv[3:]] or [0, 20, 30]
G = {"T": dict(NumMels=192, MelFmax=16000.0, Window=1280, Resolut=4096), "S": dict(NumMels=80, MelFmax=8000.0, Window=256, Resolut=1024)}[geom]
U, F, I = 256, 400, 60
m = gomel_amd.NewMel()
for k, v in dict(G, GriffinLimIterations=I).items(): setattr(m, k, v)
e = m.engine()
K = m.NumMels
errs = {k: [] for k in KS}
t0 = time.time()
for blk in range(blocks):
    first = blk * U
    mels = [np.random.default_rng(first + u).uniform(math.log(1e-5), math.log(50.0), size=(F * K, 2)) for u in range(U)]
    seeds = [1000 + first + u for u in range(U)]
    b = gomel_amd.Batch(e, [F * K] * U)
    b.upload(mels, None, seeds)
    def run(s):
        e.set_precision_schedule(s)
        bb = gomel_amd.Batch(e, [F * K] * U); bb.upload(mels, None, seeds); bb.run(); o = bb.download(); bb.close()
        return o
    b.close()
    ref = run(-1)
    for k in KS:
        out = run(k)
        errs[k] += [float(np.sqrt(np.mean((a - r) ** 2))) for a, r in zip(out, ref)]
    print("block %d done (%.0f s)" % (blk, time.time() - t0), flush=True)
for k in KS:
    v = np.array(errs[k])
    print("%s K=%2d: %d utterances: median %.2e p99 %.2e p99.9 %.2e max %.2e | over 1e-4: %d, over 3e-5: %d, over 1e-5: %d | worst %s" % (
        geom, k, len(v), np.median(v), np.percentile(v, 99), np.percentile(v, 99.9), v.max(), (v > 1e-4).sum(), (v > 3e-5).sum(), (v > 1e-5).sum(), np.argsort(-v)[:4].tolist()))
