"""Developer probe: fused vs generic kernels for one utterance of F frames, per geometry (where is the crossover?)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import gomel_amd
G = {"T": dict(NumMels=192, MelFmax=16000.0, Window=1280, Resolut=4096), "D": dict(NumMels=160, MelFmax=8000.0, Window=256, Resolut=2048),
     "S": dict(NumMels=80, MelFmax=8000.0, Window=256, Resolut=1024), "Q": dict(NumMels=64, MelFmax=8000.0, Window=160, Resolut=512),
     "E": dict(NumMels=40, MelFmax=4000.0, Window=80, Resolut=256)}
for g, fields in G.items():
    for F in (128, 256, 512, 1024, 2048, 4096, 8192, 16384):
        res = {}
        for path in (1, 2):
            m = gomel_amd.NewMel()
            for k, v in dict(fields, GriffinLimIterations=60).items(): setattr(m, k, v)
            e = m.engine(); e.set_path(path)
            mel = np.random.default_rng(0).uniform(np.log(1e-5), np.log(50.0), size=(F * m.NumMels, 2))
            b = gomel_amd.Batch(e, [F * m.NumMels]); b.upload([mel], None, [1])
            for _ in range(2): b.run()
            b.sync()
            t = time.perf_counter()
            for _ in range(5): b.run()
            b.sync()
            res[path] = (time.perf_counter() - t) / 5 * 1e3
            b.close(); e.close()
        print("%s N %4d F %5d: generic %.3f ms  fused %.3f ms  -> %s" % (g, fields["Resolut"], F, res[1], res[2], "fused" if res[2] < res[1] else "generic"), flush=True)
