"""Developer probe: ToMel host-to-host time vs length (fp64 STFT kernels), next to FromMel at 2 iterations."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import gomel_amd
for N, hop, K, fmax, sr in ((4096, 1280, 192, 16000.0, 44100), (2048, 256, 160, 8000.0, 22050), (1024, 256, 80, 8000.0, 22050)):
    for seconds in (10, 60, 600):
        m = gomel_amd.NewMel()
        m.NumMels, m.MelFmax, m.Window, m.Resolut, m.GriffinLimIterations = K, fmax, hop, N, 2
        e = m.engine()
        x = np.random.default_rng(0).standard_normal(sr * seconds) * 0.1
        for _ in range(2): mel = e.to_mel(x)
        t = time.perf_counter()
        R = 5
        for _ in range(R): mel = e.to_mel(x)
        tm = (time.perf_counter() - t) / R * 1e3
        for _ in range(2): e.from_mel(mel, seed=1)
        t = time.perf_counter()
        for _ in range(R): e.from_mel(mel, seed=1)
        fm = (time.perf_counter() - t) / R * 1e3
        F = len(mel) // K
        print("N %d hop %d: %4d s audio, %6d frames: ToMel %.3f ms (%.2f M frames/s) | FromMel(2 it) %.3f ms" % (N, hop, seconds, F, tm, F / tm / 1e3, fm), flush=True)
        e.close()
