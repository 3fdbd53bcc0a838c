"""Host logic without a GPU: the table blob gomel_tables_build makes from a configuration (gossp's Hann, undomel's and domel's
index math, mel/impl.go:311-322,348-359) against the oracle's literal undomel / domel -- the library's rows are applied with
numpy here exactly as the kernels apply them (copy / two-tap / sum divided by div)."""
import ctypes
import struct

import numpy as np
import pytest

ROW = np.dtype([("lo", "<i4"), ("hi", "<i4"), ("mode", "<i4"), ("pad", "<i4"), ("modlo", "<f8"), ("div", "<f8")])   # struct BinRow
HEADER = struct.Struct("<IIiiiiiidd")                                                                                   # struct TableHeader


def config(**kw):
    import gomel_amd
    m = gomel_amd.NewMel()
    for k, v in kw.items():
        setattr(m, k, v)
    return m._config()


def parse(blob, N, K):
    magic, version, n, hop, k, periodic, dom_status, _, fmin, fmax = HEADER.unpack_from(blob, 0)
    assert magic == 0x474D454C and n == N and k == K
    off = HEADER.size
    window = np.frombuffer(blob, dtype="<f8", count=N, offset=off)
    off += N * 8
    undo = np.frombuffer(blob, dtype=ROW, count=N // 2, offset=off)
    off += (N // 2) * ROW.itemsize
    dom = np.frombuffer(blob, dtype=ROW, count=K, offset=off)
    assert off + K * ROW.itemsize == len(blob)
    return dict(hop=hop, periodic=periodic, dom_status=dom_status, fmin=fmin, fmax=fmax, window=window, undo=undo, dom=dom)


def apply_rows(rows, e):
    """e: [F, n] -> [F, len(rows)], the three modes of apply_row() in gl_kernels.hpp."""
    out = np.empty((e.shape[0], len(rows)))
    for i, r in enumerate(rows):
        lo, hi = int(r["lo"]), int(r["hi"])
        if r["mode"] == 0:
            out[:, i] = e[:, lo]
        elif r["mode"] == 1:
            out[:, i] = e[:, lo] * (1 - r["modlo"]) + e[:, hi] * r["modlo"]
        else:
            tot = np.zeros(e.shape[0])
            for k in range(lo, hi):
                tot = tot + e[:, k]
            with np.errstate(divide="ignore", invalid="ignore"):
                out[:, i] = tot / r["div"]
    return out


CASES = [dict(NumMels=192, MelFmax=16000.0, Window=1280, Resolut=4096), dict(NumMels=80, MelFmax=8000.0, Window=256, Resolut=1024),
         dict(NumMels=160, MelFmax=8000.0, Window=256, Resolut=2048), dict(NumMels=80, MelFmin=50.0, MelFmax=7600.0, Window=256, Resolut=1024),
         dict(NumMels=64, MelFmax=8000.0, Window=160, Resolut=512), dict(NumMels=40, MelFmax=4000.0, Window=80, Resolut=256),
         dict(NumMels=300, MelFmax=22050.0, Window=441, Resolut=4096), dict(NumMels=13, MelFmin=120.0, MelFmax=3000.0, Window=99, Resolut=512)]


@pytest.mark.parametrize("fields", CASES, ids=lambda f: "K%d_N%d" % (f["NumMels"], f["Resolut"]))
def test_tables_from_config_match_the_oracle(oracle, fields):
    import gomel_amd
    cfg = config(**fields)
    N, K = fields["Resolut"], fields["NumMels"]
    fmin, fmax = fields.get("MelFmin", 0.0), fields["MelFmax"]
    t = parse(gomel_amd.build_tables(cfg), N, K)
    assert t["hop"] == fields["Window"] and t["fmin"] == fmin and t["fmax"] == fmax and t["periodic"] == 0
    assert np.max(np.abs(t["window"] - oracle.hann(N))) < 1e-15                 # symmetric Hann, gossp window.CreateHanning
    rng = np.random.default_rng(K)
    bands = np.exp(rng.uniform(np.log(1e-5), np.log(50.0), size=(3, K, 2)))
    want = oracle.undomel(N // 2, K, bands.reshape(3 * K, 2), fmin, fmax).reshape(3, N // 2, 2)
    for c in range(2):
        got = apply_rows(t["undo"], bands[:, :, c])
        bad = ~np.isfinite(want[:, :, c])
        assert np.array_equal(bad, ~np.isfinite(got))
        assert np.max(np.abs(got[~bad] - want[:, :, c][~bad]) / np.abs(want[:, :, c][~bad])) < 1e-13
    bins = rng.uniform(0.0, 3.0, size=(3, N // 2 + 1))
    pairs = np.stack([bins[:, :N // 2], bins[:, 1:]], axis=2)                     # channel 1 = the same spectrum one bin up
    try:
        wantd = oracle.domel(N // 2, K, pairs.reshape(3 * (N // 2), 2), fmin, fmax).reshape(3, K, 2)
    except IndexError:
        assert t["dom_status"] == 5                                             # the reference panics in ToMel; status-coded
        return
    assert t["dom_status"] == 0
    for c in range(2):
        gotd = apply_rows(t["dom"], bins[:, c:])
        assert np.max(np.abs(gotd - wantd[:, :, c])) < 1e-12 * max(1.0, float(np.abs(wantd).max()))


def test_tables_fuzz_and_refusals(oracle):
    import gomel_amd
    rng = np.random.default_rng(99)
    ok = refused = 0
    for trial in range(120):
        N = int(rng.choice([256, 512, 1024, 2048, 4096]))
        K = int(rng.integers(4, 400))
        fmin = float(rng.choice([0.0, rng.uniform(0, 500)]))
        fmax = float(rng.uniform(800, 26000))
        cfg = config(NumMels=K, MelFmin=fmin, MelFmax=fmax, Resolut=N, Window=N // 4)
        bands = np.exp(rng.uniform(-3, 3, size=(1, K, 2)))
        try:
            want = oracle.undomel(N // 2, K, bands.reshape(K, 2), fmin, fmax)[:, 0]
        except (IndexError, ValueError):
            want = None
        if want is None:
            with pytest.raises(gomel_amd.GomelError) as ei:
                gomel_amd.build_tables(cfg)
            assert ei.value.status == 5, (N, K, fmin, fmax)
            refused += 1
            continue
        t = parse(gomel_amd.build_tables(cfg), N, K)
        got = apply_rows(t["undo"], bands[:, :, 0])[0]
        bad = ~np.isfinite(want)
        assert np.array_equal(bad, ~np.isfinite(got)), (N, K, fmin, fmax)
        assert np.max(np.abs(got[~bad] - want[~bad]) / np.abs(want[~bad])) < 1e-13, (N, K, fmin, fmax)
        ok += 1
    assert ok >= 60 and ok + refused == 120, (ok, refused)


def test_tables_build_argument_checks():
    from gomel_amd import _lib
    L = _lib.lib()
    cfg = config(Resolut=70000)
    assert L.gomel_tables_bytes_for(ctypes.byref(cfg)) == 0
    assert L.gomel_tables_bytes_for(ctypes.byref(config(Resolut=3000))) > 0       # any length up to 65536 (radix-2 / Bluestein, above 8192 through HBM)
    assert L.gomel_tables_bytes_for(ctypes.byref(config(Resolut=6000))) > 0
    assert L.gomel_tables_bytes_for(ctypes.byref(config(Resolut=9000))) > 0 and L.gomel_tables_bytes_for(ctypes.byref(config(Resolut=65536))) > 0
    buf = ctypes.create_string_buffer(16)
    assert L.gomel_tables_build(ctypes.byref(cfg), buf, 16) == 2
    cfg = config()
    assert L.gomel_tables_build(ctypes.byref(cfg), buf, 16) == 6
    assert L.gomel_tables_build(None, buf, 16) == 7
    n = L.gomel_tables_bytes_for(ctypes.byref(cfg))
    assert n == HEADER.size + 2048 * 8 + (1024 + 160) * ROW.itemsize
