#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/ from the CPU oracle (oracle/gomel_oracle.py).

The reference holds no golden vectors for the mel path and cannot be run here, so these fixtures freeze the
oracle's outputs ("parity unpinned", see oracle/gomel_oracle.py).  Inputs are regenerated from seeds by the tests; only
expected outputs (and the tables) are stored.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import gomel_oracle as go  # noqa: E402

CASES = {
    # name: (config fields, frames, mel seed, x0 seed, iteration counts)
    "toy_n64": (dict(NumMels=8, MelFmax=8000.0, Window=20, Resolut=64), 6, 1, 2, (0, 1, 2, 60)),
    "tts_n1024": (dict(NumMels=80, MelFmax=8000.0, Window=256, Resolut=1024), 8, 3, 4, (1, 2, 60)),
    "default_n2048": (dict(NumMels=160, MelFmax=8000.0, Window=256, Resolut=2048), 6, 5, 6, (2,)),
    "towav_n4096": (dict(NumMels=192, MelFmax=16000.0, Window=1280, Resolut=4096), 4, 7, 8, (2, 60)),
}
TABLES = [(192, 0.0, 16000.0, 2048), (80, 0.0, 8000.0, 512), (160, 0.0, 8000.0, 1024), (80, 50.0, 7600.0, 512)]


def main():
    out = {}
    for name, (fields, F, ms, xs, iters) in CASES.items():
        for I in iters:
            cfg = go.MelConfig(GriffinLimIterations=I, **fields)
            mel = go.random_logmel(F, cfg.NumMels, seed=ms)
            x0 = go.x0_splitmix(xs, go.out_len(cfg, len(mel)))
            out["%s_i%d_wave" % (name, I)] = go.from_mel(cfg, mel, x0)
    for K, fmin, fmax, H in TABLES:
        rows = np.array(go.undomel_rows(H, K, fmin, fmax))
        out["undomel_%d_%g_%g_%d" % (K, fmin, fmax, H)] = rows
        out["domel_%d_%g_%g_%d" % (K, fmin, fmax, H)] = np.array(go.domel_rows(H, K, fmin, fmax))
    # ToMel on a short chirp (config 2's signal, 0.5 s) for two geometries
    sig = go.config2_signal(seconds=0.5)
    for name in ("tts_n1024", "towav_n4096"):
        cfg = go.MelConfig(**CASES[name][0])
        out["%s_tomel" % name] = go.to_mel(cfg, sig)
    out["x0_seed1_first8"] = go.x0_splitmix(1, 8)
    np.savez_compressed(os.path.join(HERE, "golden_v1.npz"), **out)
    # PNG fixture: the 192-row image the towav configuration needs (the reference's banner PNG is 183x80 legacy RGB and
    # cannot run with NumMels=192): 1.2 s of config 2's signal -> oracle ToMel -> dumpimage.
    cfg = go.towav_config()
    sig = go.config2_signal(seconds=1.2)
    mel = go.to_mel(cfg, sig)
    img = go.dumpimage_pixels(mel, cfg.NumMels, True, float(len(sig) * cfg.NumMels) / float(len(mel)), 44100.0)
    go.write_png_rgba(os.path.join(HERE, "chirp_192.png"), img)
    print("wrote", sorted(out)[:4], "...", len(out), "arrays; chirp_192.png", img.shape)


if __name__ == "__main__":
    main()
