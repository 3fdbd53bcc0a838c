"""Do two chunk-sized batches (12 800 frames each) finish sooner when their launches are queued on two streams at once than one
after the other?  (Two handles = two streams; ToWavPng geometry, 60 iterations.)"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gomel_amd
U, F, K = int(sys.argv[1]) if len(sys.argv) > 1 else 32, 400, 192
def mk():
    m = gomel_amd.NewMel()
    m.NumMels, m.MelFmax, m.Window, m.Resolut, m.GriffinLimIterations = K, 16000.0, 1280, 4096, 60
    return m.engine()
e1, e2 = mk(), mk()
base = np.random.default_rng(0).uniform(np.log(1e-5), np.log(50.0), size=(F * K, 2))
def batch(e, n):
    b = gomel_amd.Batch(e, [F * K] * n); b.upload([base] * n, None, list(range(n))); return b
b1, b2, bb = batch(e1, U), batch(e2, U), batch(e1, 2 * U)
for b in (b1, b2, bb): b.run(); b.sync()
def t(fn, n=5):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[n // 2] * 1e3
seq = t(lambda: (b1.run(), b1.sync(), b2.run(), b2.sync()))
def conc():
    b1.run(); b2.run(); b1.sync(); b2.sync()
con = t(conc)
one = t(lambda: (bb.run(), bb.sync()))
print("%d + %d utterances: one after the other %.2f ms, both queued at once %.2f ms, as one batch %.2f ms" % (U, U, seq, con, one))
