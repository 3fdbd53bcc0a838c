"""One ToWavPng utterance (342 frames, 60 iterations) in 64-bit mode: resident time and dominant kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gomel_amd
for bits in (32, 64):
    m = gomel_amd.NewMel()
    m.NumMels, m.MelFmax, m.Window, m.Resolut, m.GriffinLimIterations = 192, 16000.0, 1280, 4096, 60
    e = m.engine(); e.set_precision(bits)
    mel = np.random.default_rng(1).uniform(np.log(1e-5), np.log(50.0), size=(342 * 192, 2))
    b = gomel_amd.Batch(e, [342 * 192]); b.upload([mel], None, [3])
    for _ in range(3): tm = b.run_timed()
    ts = sorted(b.run_timed()["total_ms"] for _ in range(10))
    for _ in range(3): e.from_mel_batch([mel], None, [3])
    t0 = time.perf_counter()
    for _ in range(20): e.from_mel_batch([mel], None, [3])
    print("%d-bit: resident %.3f ms (path %d, %s, %d launches), host to host %.3f ms" % (bits, ts[5], tm["fast_path"], tm["dom_kernel"], tm["iter_launches"], (time.perf_counter() - t0) / 20 * 1e3))
    b.close(); e.close()
