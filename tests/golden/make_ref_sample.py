"""Writes tests/golden/ref_schema_sample.json: a file with the SCHEMA of the Go pinning kit's output
(bindings/go/mel/golden_gen_test.go) whose numbers come from this repository's own oracle.  It pins nothing -- it exists so that
tests/test_reference_vectors.py (the consumer of real reference vectors in tests/golden/ref/*.json) is exercised end to end even
when no Go-generated file is present.  Re-run after changing the schema:  python tests/golden/make_ref_sample.py"""
import json
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import gomel_oracle as go  # noqa: E402


def c2(z):
    return [[float(v.real), float(v.imag)] for v in np.asarray(z).ravel()]


def cfgd(c):
    return {"NumMels": c.NumMels, "Window": c.Window, "Resolut": c.Resolut, "GriffinLimIterations": c.GriffinLimIterations,
            "MelFmin": c.MelFmin, "MelFmax": c.MelFmax, "TuneMul": c.TuneMul, "TuneAdd": c.TuneAdd}


def main():
    g = {"Meta": {"go": "none", "note": "SELF-GENERATED from oracle/gomel_oracle.py: exercises the loader, pins nothing"},
         "Windows": {}, "NumFrames": [], "FFT": [], "RandSeed1": [], "RandSeed42": [], "HzToMel": [], "MelToHz": [], "Pad": [], "IsPadded": [],
         "Tables": [], "FromMel": [], "ToMel": [], "ISTFT": []}
    for hop, n in ((20, 64), (3, 8), (256, 1024)):
        g["Windows"]["%dx%d" % (hop, n)] = go.hann(n).tolist()
        for ln in (n, n + hop - 1, n + hop, n + 7 * hop + 1):
            g["NumFrames"].append([ln, hop, n, go.num_frames(ln, n, hop)])
    for n in (8, 12, 64):
        imp = np.zeros(n); imp[1] = 1
        ramp = np.arange(n) - 0.25 * n
        for x in (imp, ramp):
            f = np.fft.fft(x)
            g["FFT"].append({"N": n, "Input": x.tolist(), "FFTReal": c2(f), "IFFT": c2(np.fft.ifft(f))})
    for v in (0, 1, 7.8125, 700, 8000, 16000):
        g["HzToMel"].append([v, go.hz_to_mel(v)]); g["MelToHz"].append([v, go.mel_to_hz(v)])
    for w in (256, 275):
        for n in (1, 15 * w - 1, 15 * w, 15 * w + 1, 16 * w + 1):
            p = len(go.pad(np.zeros(n), w))
            g["Pad"].append([n, w, p])
            for q in (p - 1, p, p + 1):
                g["IsPadded"].append([n, q, w, int(go.is_padded(n, q, w))])
    cfgs = [go.MelConfig(NumMels=8, Window=20, Resolut=64, GriffinLimIterations=3, MelFmax=8000.0),
            go.MelConfig(NumMels=80, Window=256, Resolut=1024, GriffinLimIterations=2, MelFmax=8000.0)]
    for c in cfgs:
        h = c.Resolut // 2
        tin = np.array([[k + 1.0, 1000.0 + k] for k in range(c.NumMels)])
        din = np.array([[k + 1.0, 5000.0 + k] for k in range(h)])
        t = {"Cfg": cfgd(c), "In": tin.tolist(), "Undomel": go.undomel(h, c.NumMels, tin, c.MelFmin, c.MelFmax).tolist(), "DomelIn": din.tolist(), "DomelPanic": ""}
        try:
            t["Domel"] = go.domel(h, c.NumMels, din, c.MelFmin, c.MelFmax).tolist()
        except IndexError:
            t["Domel"], t["DomelPanic"] = None, "panic"
        g["Tables"].append(t)
        frames = 5
        for it in (0, 2):
            c.GriffinLimIterations = it
            mel = go.random_logmel(frames, c.NumMels, seed=3)
            x0 = go.x0_splitmix(5, go.out_len(c, len(mel)))
            after = mel.copy()
            wave = go.from_mel(c, after, x0)
            g["FromMel"].append({"Cfg": cfgd(c), "Mel": mel.tolist(), "X0": x0.tolist(), "Seed": 5, "Wave": wave.tolist(), "MelAfter": after.tolist()})
        sig = go.config2_signal(sr=22050, seconds=(15 * c.Window + 3 * c.Window + 17) / 22050.0, seed=7)
        g["ToMel"].append({"Cfg": cfgd(c), "Wave": sig.tolist(), "Pairs": go.to_mel(c, sig).tolist()})
    rng = np.random.default_rng(77)
    for hop, n, frames, it in ((3, 8, 4, 2), (20, 64, 5, 3), (256, 1024, 3, 2)):
        spec = rng.uniform(0.1, 2.0, size=(frames, n)) * np.exp(2j * np.pi * (rng.uniform(size=(frames, n)) - 0.5))
        w = go.hann(n)
        x0 = go.x0_splitmix(9, n + (frames - 1) * hop)
        out = spec.copy()
        wave = go.istft_griffin_lim(w, hop, out, it, x0)
        g["ISTFT"].append({"FrameShift": hop, "FrameLen": n, "Iterations": it, "Window": w.tolist(), "SpecIn": [c2(r) for r in spec],
                           "SpecOut": [c2(r) for r in out], "X0": x0.tolist(), "Seed": 9, "Wave": wave.tolist()})
    with open(os.path.join(HERE, "ref_schema_sample.json"), "w") as f:
        json.dump(g, f)
    print("wrote", os.path.join(HERE, "ref_schema_sample.json"), os.path.getsize(os.path.join(HERE, "ref_schema_sample.json")), "bytes")


if __name__ == "__main__":
    main()
