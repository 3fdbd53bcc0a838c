"""Feasibility study (round 4): can the utterances whose fp32 Griffin-Lim trajectory leaves the fp64 one be RECOGNISED from quantities the
fp32 kernel already has in registers?  If so the default mode could run few fp64 iterations and redo only the flagged utterances in fp64.

Emulation in torch (not the library's kernels: the statistics need every bin of every iteration): per block of utterances the fp64
trajectory of I iterations, and for every k in KS the trajectory "fp64 up to k, fp32 from k on"; per utterance the final RMS difference
(what scripts/calibrate_schedule.py calls the error) and, from the fp32 iterations, candidate statistics of the phase-projection's gain
a = m / |X| (the factor k_gl_fused's spectral step multiplies a bin by):
  zmax[i]  max over bins of a * sqrt(P_f)            P_f = mean_b |X_fb|^2, the frame's level (rounding noise in a bin ~ eps * sqrt(P_f))
  zsum[i]  sqrt(mean over bins of a^2 * P_f)
  cons[i]  sqrt(sum (|X| - m)^2 / sum m^2)           the spectral inconsistency
usage: detector_study.py GEOM BLOCKS OUT.npz [I] [k,k,...]     (random class, seeds as calibrate_schedule.py's with CAL_SEED0)
"""
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import gomel_amd

GEOMS = {"T": dict(NumMels=192, MelFmax=16000.0, Window=1280, Resolut=4096),
         "S": dict(NumMels=80, MelFmax=8000.0, Window=256, Resolut=1024),
         "D": dict(NumMels=160, MelFmax=8000.0, Window=256, Resolut=2048)}
U, F = 128, 400


def gl_step(x, mag, win, N, hop, R, stats=None):
    """One Griffin-Lim iteration of mel/mel.go:87-125 on x [U, L] (dtype of x); mag [U, F, H+1]."""
    Uu, L = x.shape
    frames = x.unfold(1, N, hop) * win                         # [U, F, N]
    X = torch.fft.rfft(frames, dim=2)
    ab = X.abs()
    tiny = torch.finfo(x.dtype).tiny
    a = mag / ab.clamp_min(tiny)
    if stats is not None:
        ac = a.clamp_max(1e9)
        P = (ab * ab).mean(dim=2, keepdim=True)
        z = ac * P.sqrt()
        stats["zmax"].append(z.amax(dim=(1, 2)).double())
        stats["zsum"].append((z * z).mean(dim=(1, 2)).sqrt().double())
        stats["cons"].append((((ab - mag) ** 2).sum(dim=(1, 2)) / (mag * mag).sum(dim=(1, 2))).sqrt().double())
        stats["amax"].append(ac.amax(dim=(1, 2)).double())
    y = torch.fft.irfft(X * a, n=N, dim=2) * win
    out = torch.zeros_like(x)
    nf = y.shape[1]
    for j in range(R):                                          # frames j, j+R, ... do not overlap
        cnt = (nf - j + R - 1) // R
        if cnt <= 0:
            continue
        view = out.as_strided((Uu, cnt, N), (L, R * hop, 1), j * hop)
        view += y[:, j::R]
    return out


def main():
    geom, blocks, outp = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    I = int(sys.argv[4]) if len(sys.argv) > 4 else 60
    KS = [int(k) for k in sys.argv[5].split(",")] if len(sys.argv) > 5 else [0, 10, 20]
    fields = GEOMS[geom]
    K, N, hop = fields["NumMels"], fields["Resolut"], fields["Window"]
    R = -(-N // hop)
    dev = torch.device("cuda", 0)
    SEED0 = int(os.environ.get("CAL_SEED0", "0"))
    m = gomel_amd.NewMel()
    for k_, v in fields.items():
        setattr(m, k_, v)
    e = m.engine()
    win = torch.from_numpy(np.array(e.get_window(), dtype=np.float64)).to(dev)
    L = N + (F - 1) * hop
    res = {"err_k%d" % k: [] for k in KS}
    for k in KS:
        for s in ("zmax", "zsum", "cons", "amax"):
            res["%s_k%d" % (s, k)] = []
    t0 = time.time()
    for blk in range(blocks):
        first = SEED0 + blk * U
        mags = np.empty((U, F, N // 2 + 1), dtype=np.float32)
        x0 = np.empty((U, L))
        for u in range(U):
            mel = np.random.default_rng(first + u).uniform(math.log(1e-5), math.log(50.0), size=(F * K, 2))
            mags[u] = e.target_magnitudes(mel)
            x0[u] = np.random.default_rng(1000 + first + u).random(L)
        mags *= N                                                  # (the library folds the inverse FFT's 1/N into them)
        mag64 = torch.from_numpy(mags).to(dev).double()
        mag32 = mag64.float()
        x = torch.from_numpy(x0).to(dev)
        snaps = {}
        for i in range(I):
            if i in KS:
                snaps[i] = x.clone()
            x = gl_step(x, mag64, win, N, hop, R)
        ref = x
        for k in KS:
            st = {"zmax": [], "zsum": [], "cons": [], "amax": []}
            y = snaps[k].float()
            w32 = win.float()
            for i in range(k, I):
                y = gl_step(y, mag32, w32, N, hop, R, st)
            res["err_k%d" % k].append((y.double() - ref).pow(2).mean(dim=1).sqrt().cpu().numpy())
            for s in st:
                res["%s_k%d" % (s, k)].append(torch.stack(st[s], dim=1).cpu().numpy())      # [U, I - k]
        print("block %d / %d after %.0f s" % (blk + 1, blocks, time.time() - t0), flush=True)
    out = {kk: np.concatenate(v, axis=0) for kk, v in res.items()}
    np.savez_compressed(outp, **out)
    for k in KS:
        err = out["err_k%d" % k]
        print("k=%d: n=%d median %.2e max %.2e over 1e-4: %d over 3e-5: %d over 1e-5: %d" % (k, len(err), np.median(err), err.max(), (err > 1e-4).sum(), (err > 3e-5).sum(), (err > 1e-5).sum()))
        bad = err > 3e-5
        for s in ("zmax", "zsum", "cons", "amax"):
            z = out["%s_k%d" % (s, k)]
            for name, red in (("max", z.max(axis=1)), ("sum", z.sum(axis=1)), ("last", z[:, -1])):
                if bad.any():
                    thr = red[bad].min()
                    print("   %s.%s: flags needed to catch every error > 3e-5: %d of %d (%.2f %%)   [threshold %.3g, median %.3g]" % (
                        s, name, (red >= thr).sum(), len(err), 100.0 * (red >= thr).mean(), thr, np.median(red)))


if __name__ == "__main__":
    main()
