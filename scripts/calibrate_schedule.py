"""Calibration of the automatic precision schedule (gomel_auto_fp64_iterations): for a geometry, an input class and a list of
iteration counts I, the error of every schedule k (first k iterations fp64, the remaining I - k fp32) against 64-bit mode, on
BLOCKS x 256 utterances of 400 frames.  Errors are taken on the device (waveforms never leave HBM).

usage: calibrate_schedule.py GEOM CLASS BLOCKS [I:k,k,k ...]
  GEOM  T (N 4096 / hop 1280 / 192 mels, towav), S (N 1024 / hop 256 / 80 mels, TTS), D (NewMel defaults: N 2048 / hop 256 / 160 mels)
  CLASS random (iid U[ln 1e-5, ln 50] log-mel, bench.py's inputs) | random0 (the same, lowest band and bin N/2 silent) | signal (synthetic audio through ToMel)
One line per (I, k): count, median, p99, p99.9, max, utterances over 1e-4 / 3e-5 / 1e-5.
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

GEOMS = {"T": (dict(NumMels=192, MelFmax=16000.0, Window=1280, Resolut=4096), 44100),
         "S": (dict(NumMels=80, MelFmax=8000.0, Window=256, Resolut=1024), 22050),
         "D": (dict(NumMels=160, MelFmax=8000.0, Window=256, Resolut=2048), 22050),
         "X": (dict(NumMels=80, MelFmax=8000.0, Window=128, Resolut=1024), 22050),        # eight-fold overlap at another frame length
         "W": (dict(NumMels=80, MelFmax=8000.0, Window=64, Resolut=1024), 22050),         # sixteen-fold
         "Y": (dict(NumMels=160, MelFmax=8000.0, Window=512, Resolut=2048), 22050),       # four-fold at N 2048
         "Z": (dict(NumMels=160, MelFmax=8000.0, Window=400, Resolut=2048), 22050)}       # 5.1-fold: just under the second table's rule
DEFAULT_GRID = {2: [0, 1], 5: [0, 2, 3], 10: [0, 2, 4, 6], 20: [0, 4, 8, 12, 16], 30: [0, 5, 10, 15, 20], 40: [0, 10, 15, 20, 25, 30],
                60: [0, 10, 20, 24, 26, 28, 30, 35, 40, 45], 100: [0, 20, 40, 50, 60, 70, 80]}
U, F = 256, 400


def signals(first, n, sr, device):
    """n synthetic utterances (a chirp, an AM tone, noise, a near-silent stretch, a noise burst): [U, n] float64 on the device."""
    t = torch.arange(n, dtype=torch.float64, device=device) / sr
    out = torch.empty((U, n), dtype=torch.float64, device=device)
    seconds = n / sr
    nyq = sr / 2.0
    for u in range(U):
        rng = np.random.default_rng(first + u)
        f0, f1 = rng.uniform(80, 800), rng.uniform(200, 0.27 * nyq)
        x = rng.uniform(0.05, 0.4) * torch.sin(2 * math.pi * (f0 * t + (f1 - f0) / (2 * seconds) * t * t))
        x += rng.uniform(0.02, 0.3) * torch.sin(2 * math.pi * rng.uniform(300, 0.23 * nyq) * t) * torch.sin(2 * math.pi * rng.uniform(1, 9) * t)
        g = torch.Generator(device=device); g.manual_seed(first + u)
        x += rng.uniform(0.001, 0.05) * torch.randn(n, dtype=torch.float64, device=device, generator=g)
        a, b = sorted(rng.integers(0, n, size=2)); x[a:a + (b - a) // 4] *= rng.uniform(0, 0.01)
        blen = min(2000, n // 8)
        k = int(rng.integers(0, n - blen))
        x[k:k + blen] += 0.5 * torch.randn(blen, dtype=torch.float64, device=device, generator=g) * torch.hann_window(blen, periodic=False, dtype=torch.float64, device=device)
        out[u] = x
    return out


def main():
    geom, cls, blocks = sys.argv[1], sys.argv[2], int(sys.argv[3])
    grid = dict(DEFAULT_GRID)
    if len(sys.argv) > 4:
        grid = {}
        for a in sys.argv[4:]:
            i, ks = a.split(":")
            grid[int(i)] = [int(k) for k in ks.split(",")]
    fields, sr = GEOMS[geom]
    dev = torch.device("cuda", 0)
    K, N, hop = fields["NumMels"], fields["Resolut"], fields["Window"]
    t0 = time.time()
    # inputs of every block, kept on the host
    mels = []
    m0 = gomel_amd.NewMel()
    for k_, v in fields.items():
        setattr(m0, k_, v)
    SEED0 = int(os.environ.get("CAL_SEED0", "0"))     # a held-out set: other seeds than the calibration's 0 ... 4095

    def make_block(blk):
        first = SEED0 + blk * U
        if cls in ("random", "random0"):
            ms = [np.random.default_rng(first + u).uniform(math.log(1e-5), math.log(50.0), size=(F * K, 2)) for u in range(U)]
            if cls == "random0":      # the same with a silent lowest band (bin 0's magnitude) and a silent bin N/2: the two REAL bins, whose clamp is a sign decision
                for a in ms:
                    v = a.reshape(F, K, 2)
                    v[:, 0, :] = math.log(1e-5)
                    v[:, K - 1, 1] = math.log(1e-5)
            return ms
        x = signals(first, N + (F - 1) * hop, sr, dev).cpu().numpy()
        return m0.engine().to_mel_batch([x[u] for u in range(U)])
    stream = len(grid) == 1                             # one iteration count: blocks are made as they are used (any number of them)
    for blk in range(1 if stream else blocks):
        mels.append(make_block(blk))
    frames = len(mels[0][0]) // K
    L = N + (frames - 1) * hop
    print("# %s %s: %d blocks x %d utterances x %d frames, inputs ready after %.0f s" % (geom, cls, blocks, U, frames, time.time() - t0), flush=True)
    ref = torch.empty((U, L), dtype=torch.float64, device=dev)
    out = torch.empty((U, L), dtype=torch.float64, device=dev)
    for I in sorted(grid):
        m = gomel_amd.NewMel()
        for k_, v in dict(fields, GriffinLimIterations=I).items():
            setattr(m, k_, v)
        e = m.engine()
        e.set_path(2)                                                  # the fused kernels, as a large batch takes them
        errs = {k: [] for k in grid[I]}
        rms = []
        for blk in range(blocks):
            seeds = [1000 + SEED0 + blk * U + u for u in range(U)]
            if stream and blk:
                mels[0] = make_block(blk)
            cur = mels[0] if stream else mels[blk]

            def run(s, dst):
                e.set_precision_schedule(s)
                b = gomel_amd.Batch(e, [frames * K] * U)
                b.upload(cur, None, seeds)
                b.run()
                b.download([dst[u] for u in range(U)])
                b.close()
            run(-1, ref)
            rms.append(float(ref.pow(2).mean().sqrt()))
            for k in grid[I]:
                if k >= I:
                    continue
                run(k, out)
                errs[k] += (out - ref).pow(2).mean(dim=1).sqrt().cpu().tolist()
        for k in grid[I]:
            if not errs[k]:
                continue
            v = np.array(errs[k])
            print("%s %-6s I=%3d k=%3d: n=%d median %.2e p99 %.2e p99.9 %.2e max %.2e | over 1e-4: %d  over 3e-5: %d  over 1e-5: %d | signal rms %.3f | worst %s" % (
                geom, cls, I, k, len(v), np.median(v), np.percentile(v, 99), np.percentile(v, 99.9), v.max(), (v > 1e-4).sum(), (v > 3e-5).sum(), (v > 1e-5).sum(),
                float(np.mean(rms)), np.argsort(-v)[:3].tolist()), flush=True)
        e.close()
        print("# I=%d done after %.0f s" % (I, time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
