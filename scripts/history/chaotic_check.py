"""Is the tail of the precision schedules made of utterances whose fp64 trajectory itself is not reproducible?  For given calibration seeds
(scripts/calibrate_schedule.py's numbering) the library's fp64-throughout result against the independent C restatement (other FFT
factorisation, other summation order: both are float64 implementations of mel/mel.go:76-139), next to a few ordinary utterances.
usage: chaotic_check.py GEOM SEED [SEED ...]"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import gomel_amd
import gomel_oracle as go
import oracle_c

GEOMS = {"T": dict(NumMels=192, MelFmax=16000.0, Window=1280, Resolut=4096), "S": dict(NumMels=80, MelFmax=8000.0, Window=256, Resolut=1024),
         "D": dict(NumMels=160, MelFmax=8000.0, Window=256, Resolut=2048)}
geom = sys.argv[1]
seeds = [int(a) for a in sys.argv[2:]]
fields = dict(GEOMS[geom], GriffinLimIterations=60)
K, F = fields["NumMels"], 400
m = gomel_amd.NewMel()
for k, v in fields.items():
    setattr(m, k, v)
e = m.engine()
cfg = go.MelConfig(**fields)
co = oracle_c.load()
for s in seeds:
    mel = np.random.default_rng(s).uniform(math.log(1e-5), math.log(50.0), size=(F * K, 2))
    row = "seed %d:" % s
    outs = {}
    for name, sched in (("fp64", -1), ("default", "auto"), ("fp32", 0)):
        e.set_precision_schedule(sched)
        outs[name] = e.from_mel_batch([mel.copy()], None, [1000 + s])[0]
    want, _ = co.from_mel(cfg, mel.copy(), go.x0_splitmix(1000 + s, len(outs["fp64"])))
    rms = lambda a: float(np.sqrt(np.mean(a * a)))
    print("%s library fp64 vs C restatement (fp64) %.2e | default vs library fp64 %.2e | fp32 vs library fp64 %.2e" % (
        row, rms(outs["fp64"] - want), rms(outs["default"] - outs["fp64"]), rms(outs["fp32"] - outs["fp64"])), flush=True)
