"""Precision schedule study: K iterations in fp64, the remaining 60-K in fp32 (magnitudes, window, signal rounded to fp32 at the
switch), against the all-fp64 trajectory.  NumPy emulation; prints per-K error statistics over the utterances given."""
import os, sys, math, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np, scipy.fft as sf
import gomel_oracle as go
from multiprocessing import Pool
F, K, I = 400, 192, 60
GEO = os.environ.get("GEO", "T")
if GEO == "T":
    cfg = go.MelConfig(NumMels=192, MelFmax=16000.0, Window=1280, Resolut=4096, GriffinLimIterations=I); N, hop = 4096, 1280
else:
    K = 80
    cfg = go.MelConfig(NumMels=80, MelFmax=8000.0, Window=256, Resolut=1024, GriffinLimIterations=I); N, hop = 1024, 256
w64 = go.hann(N); w32 = w64.astype(np.float32)
idx = np.arange(N)[None, :] + hop * np.arange(F)[:, None]
L = N + (F - 1) * hop
KS = [0, 10, 20, 30, 40, 45, 50, 55]
def ola(y, dt):
    xn = np.zeros(L, dtype=dt)
    for i in range(F): xn[i * hop:i * hop + N] += y[i]
    return xn
def it64(x64, M):
    X = np.fft.rfft(x64[idx] * w64[None, :], axis=1); a = np.abs(X)
    return ola(np.fft.irfft(M * np.where(a > 0, X / np.where(a > 0, a, 1), 1.0), n=N, axis=1) * w64[None, :], np.float64)
def it32(x32, M32):
    X3 = sf.rfft(x32[idx] * w32[None, :], axis=1); a3 = np.abs(X3)
    Y3 = (M32 * np.where(a3 > 0, X3 / np.where(a3 > 0, a3, 1), 1.0)).astype(np.complex64)
    return ola(sf.irfft(Y3, n=N, axis=1).astype(np.float32) * w32[None, :], np.float32)
def one(u):
    mel = np.random.default_rng(u).uniform(math.log(1e-5), math.log(50.0), size=(F * K, 2))
    M = go.target_magnitudes(cfg, np.exp(mel)); M32 = M.astype(np.float32)
    x0 = go.x0_splitmix(1000 + u, L)
    x64 = x0.copy(); snaps = {}
    for it in range(I):
        if it in KS: snaps[it] = x64.copy()
        x64 = it64(x64, M)
    out = []
    for k in KS:
        x32 = snaps[k].astype(np.float32)
        for it in range(k, I): x32 = it32(x32, M32)
        out.append(float(np.sqrt(np.mean((x32.astype(np.float64) - x64) ** 2))))
    return out
if __name__ == "__main__":
    a, b = int(sys.argv[1]), int(sys.argv[2])
    us = list(range(a, b))
    t = time.time()
    with Pool(8) as p: res = np.array(p.map(one, us, chunksize=1))
    np.save("sched_%s_%d_%d.npy" % (GEO, a, b), res)
    print("geo", GEO, "utts", a, b, "%.0f s" % (time.time() - t))
    for j, k in enumerate(KS):
        e = res[:, j]
        print("K=%2d  median %.2e  p90 %.2e  p99 %.2e  max %.2e  n>1e-4 %d  n>5e-5 %d" % (k, np.median(e), np.percentile(e, 90), np.percentile(e, 99), e.max(), (e > 1e-4).sum(), (e > 5e-5).sum()))
