"""One calibration utterance under every schedule k (fp64 first k iterations, fp32 the rest), on the persistent path (alone) and on the fused
path (with 255 neighbours): RMS distance from fp64 throughout.  usage: chaotic_k.py GEOM SEED"""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import gomel_amd
GEOMS = {"T": dict(NumMels=192, MelFmax=16000.0, Window=1280, Resolut=4096), "S": dict(NumMels=80, MelFmax=8000.0, Window=256, Resolut=1024),
         "D": dict(NumMels=160, MelFmax=8000.0, Window=256, Resolut=2048)}
geom, s = sys.argv[1], int(sys.argv[2])
fields = dict(GEOMS[geom], GriffinLimIterations=60)
K, F = fields["NumMels"], 400
m = gomel_amd.NewMel()
for k, v in fields.items():
    setattr(m, k, v)
e = m.engine()
base = s - (s % 256)
mels = [np.random.default_rng(base + u).uniform(math.log(1e-5), math.log(50.0), size=(F * K, 2)) for u in range(256)]
seeds = [1000 + base + u for u in range(256)]
i = s - base
rms = lambda a: float(np.sqrt(np.mean(a * a)))
def run(sched, batch):
    e.set_precision_schedule(sched)
    if batch:
        e.set_path(2)
        out = e.from_mel_batch([x.copy() for x in mels], None, seeds)[i]
        e.set_path(0)
        return out
    return e.from_mel_batch([mels[i].copy()], None, [seeds[i]])[0]
ref1, ref256 = run(-1, False), run(-1, True)
print("fp64 alone vs fp64 in the batch: %.2e" % rms(ref1 - ref256))
for k in (0, 10, 20, 30, 35, 40, 45, 50, 55, 58, 59):
    print("k=%2d: alone (persistent) %.2e   in the batch (fused) %.2e" % (k, rms(run(k, False) - ref1), rms(run(k, True) - ref256)), flush=True)
