"""Parity on SIGNAL-DERIVED spectrograms at scale: N utterances of synthetic audio (chirps, AM tones, noise bursts, silence gaps)
through ToMel, then FromMel with 60 iterations in fp32 and in 64-bit mode (which matches the C oracle to 1e-13, tests)."""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import gomel_amd
U = int(sys.argv[1]) if len(sys.argv) > 1 else 128
def sig(seed, sr=44100, seconds=8.0):
    rng = np.random.default_rng(seed)
    n = int(sr * seconds); t = np.arange(n) / sr
    f0, f1 = rng.uniform(80, 800), rng.uniform(200, 6000)
    x = rng.uniform(0.05, 0.4) * np.sin(2 * np.pi * (f0 * t + (f1 - f0) / (2 * seconds) * t * t))
    x += rng.uniform(0.02, 0.3) * np.sin(2 * np.pi * rng.uniform(300, 5000) * t) * np.sin(2 * np.pi * rng.uniform(1, 9) * t)
    x += rng.uniform(0.001, 0.05) * rng.standard_normal(n)
    a, b = sorted(rng.integers(0, n, size=2)); x[a:a + (b - a) // 4] *= rng.uniform(0, 0.01)      # a near-silent stretch
    k = rng.integers(0, n - 2000); x[k:k + 2000] += 0.5 * rng.standard_normal(2000) * np.hanning(2000)   # a noise burst
    return x
m = gomel_amd.NewMel()
m.NumMels, m.MelFmax, m.Window, m.Resolut, m.GriffinLimIterations = 192, 16000.0, 1280, 4096, 60
e = m.engine()
mels = e.to_mel_batch([sig(s) for s in range(U)])
seeds = list(range(U))
e.set_precision(32); o32 = e.from_mel_batch(mels, None, seeds)
e.set_precision(64); o64 = e.from_mel_batch(mels, None, seeds)
err = np.array([float(np.sqrt(np.mean((a - b) ** 2))) for a, b in zip(o32, o64)])
rmsv = np.array([float(np.sqrt(np.mean(b ** 2))) for b in o64])
print("%d signal-derived utterances (%d frames each), 60 iterations: fp32 vs 64-bit mode rms error median %.2e p99 %.2e max %.2e (over 1e-4: %d); signal rms %.3f" % (
    U, len(mels[0]) // 192, np.median(err), np.percentile(err, 99), err.max(), int((err > 1e-4).sum()), rmsv.mean()))
