"""The default schedule on the PERSISTENT kernel (one utterance per call) against 64-bit mode: n utterances of 342 frames (towav geometry)
or 8 x 100 frames (TTS geometry), 60 iterations.  usage: persist_sched_check.py [T|S] [n]"""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import gomel_amd
geom = sys.argv[1] if len(sys.argv) > 1 else "T"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
G, frames = {"T": (dict(NumMels=192, MelFmax=16000.0, Window=1280, Resolut=4096), [342]), "S": (dict(NumMels=80, MelFmax=8000.0, Window=256, Resolut=1024), [100] * 8)}[geom]
m = gomel_amd.NewMel()
for k, v in dict(G, GriffinLimIterations=60).items(): setattr(m, k, v)
e = m.engine()
K = m.NumMels
errs = []
for u in range(n):
    rng = np.random.default_rng(u)
    mels = [rng.uniform(math.log(1e-5), math.log(50.0), size=(f * K, 2)) for f in frames]
    seeds = [1000 + u * 8 + i for i in range(len(frames))]
    e.set_precision(0); a = e.from_mel_batch(mels, None, seeds)
    e.set_precision(64); b = e.from_mel_batch(mels, None, seeds)
    errs += [float(np.sqrt(np.mean((x - y) ** 2))) for x, y in zip(a, b)]
v = np.array(errs)
bt = gomel_amd.Batch(e, [f * K for f in frames]); e.set_precision(0); bt2 = gomel_amd.Batch(e, [f * K for f in frames])
bt2.upload(mels, None, seeds); tm = bt2.run_timed(); bt.close(); bt2.close()
print("%s persistent path (fast_path %d), default schedule k=%d vs 64-bit mode, %d utterances: median %.2e p99 %.2e max %.2e | over 1e-4: %d over 3e-5: %d over 1e-5: %d" % (
    geom, tm["fast_path"], e.fp64_iterations(), len(v), np.median(v), np.percentile(v, 99), v.max(), (v > 1e-4).sum(), (v > 3e-5).sum(), (v > 1e-5).sum()))
