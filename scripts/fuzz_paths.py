"""Fuzz of the path choice: random small batches (utterance counts, lengths, geometry, precision mode) through the library's own choice of
kernels against the same call forced onto the generic pair.  Persistent launches (one or two) must give the generic pair's BITS; the fused
kernels its result to rounding (other summation order of the overlap-add): 1e-12 with fp64 iterations only; with fp32 iterations two fp32
summation orders drift apart as any two fp32 runs do on random spectrograms (1e-6 typically, 1e-4 on the odd unstable one: DESIGN.md section 2),
so there only gross differences (> 1e-3: a frame in the wrong place) count.  usage: fuzz_paths.py [cases] [seed]"""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import gomel_amd
GEOMS = {"T": dict(NumMels=192, MelFmax=16000.0, Window=1280, Resolut=4096), "S": dict(NumMels=80, MelFmax=8000.0, Window=256, Resolut=1024),
         "D": dict(NumMels=160, MelFmax=8000.0, Window=256, Resolut=2048), "Q": dict(NumMels=64, MelFmax=8000.0, Window=128, Resolut=512),
         # hops OFF the N/16 grid: the any-hop fused kernels (path 3), even and odd
         "A": dict(NumMels=80, MelFmax=8000.0, Window=400, Resolut=2048), "B": dict(NumMels=80, MelFmax=8000.0, Window=300, Resolut=1024),
         "C": dict(NumMels=96, MelFmax=16000.0, Window=441, Resolut=4096)}
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
rms = lambda a: float(np.sqrt(np.mean(a * a)))
engines = {}
bad = 0
paths = {}
for c in range(cases):
    g = str(rng.choice(list(GEOMS)))
    I = int(rng.choice([2, 3, 6, 7, 12]))
    sched = [("auto", "default"), (0, "fp32"), (-1, "fp64"), (int(rng.integers(1, I)) if I > 1 else 0, "k")][int(rng.integers(0, 4))]
    key = (g, I)
    if key not in engines:
        m = gomel_amd.NewMel()
        for k, v in dict(GEOMS[g], GriffinLimIterations=I).items():
            setattr(m, k, v)
        engines[key] = (m, m.engine())
    e = engines[key][1]
    K = GEOMS[g]["NumMels"]
    cap = {"T": 1200, "S": 5000, "D": 2600, "Q": 9000, "A": 2600, "B": 5000, "C": 1800}[g]
    U = int(rng.integers(1, 9))
    frames = [int(rng.integers(1, max(2, cap // U * 2))) for _ in range(U)]
    mels = [rng.uniform(math.log(1e-5), math.log(50.0), size=(f * K, 2)) for f in frames]
    seeds = [int(s) for s in rng.integers(0, 1 << 30, size=U)]
    e.set_precision_schedule(sched[0])
    outs = {}
    for path in (0, 1):
        e.set_path(path)
        b = gomel_amd.Batch(e, [len(m_) for m_ in mels])
        b.upload(mels, None, seeds)
        tm = b.run_timed()
        outs[path] = (b.download(), tm["fast_path"])
        b.close()
    e.set_path(0)
    fp = outs[0][1]
    paths[fp] = paths.get(fp, 0) + 1
    for a, r in zip(outs[0][0], outs[1][0]):
        if fp in (0, 2):
            ok = np.array_equal(a, r)
        else:
            ok = rms(a - r) < (1e-12 if sched[1] == "fp64" else 1e-3) * max(1.0, rms(r))
        if not ok:
            bad += 1
            print("MISMATCH case %d geom %s I %d mode %s frames %s path %d: rms diff %.3e" % (c, g, I, sched, frames, fp, rms(a - r)), flush=True)
            break
print("fuzz: %d cases, paths taken %s (0 generic, 1 fused, 2 persistent, 3 fused any-hop), mismatches %d" % (cases, paths, bad))
missing = [p for p in (0, 1, 2, 3) if not paths.get(p)]
if missing:
    print("fuzz: paths never taken: %s -- the draw does not cover the kernels, not a pass" % missing)
sys.exit(1 if bad or missing else 0)
