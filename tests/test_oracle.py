"""CPU tests of the oracle itself: the NumPy restatement, the C restatement, the golden fixtures, and the invariants the
reference's algorithm has.  No GPU."""
import math
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "golden_v1.npz"))

CASES = {
    "toy_n64": (dict(NumMels=8, MelFmax=8000.0, Window=20, Resolut=64), 6, 1, 2, (0, 1, 2, 60)),
    "tts_n1024": (dict(NumMels=80, MelFmax=8000.0, Window=256, Resolut=1024), 8, 3, 4, (1, 2, 60)),
    "default_n2048": (dict(NumMels=160, MelFmax=8000.0, Window=256, Resolut=2048), 6, 5, 6, (2,)),
    "towav_n4096": (dict(NumMels=192, MelFmax=16000.0, Window=1280, Resolut=4096), 4, 7, 8, (2, 60)),
}


def rms(a):
    return float(np.sqrt(np.mean(np.square(a))))


def case_inputs(oracle, name, iters):
    fields, F, ms, xs, _ = CASES[name]
    cfg = oracle.MelConfig(GriffinLimIterations=iters, **fields)
    mel = oracle.random_logmel(F, cfg.NumMels, seed=ms)
    return cfg, mel, oracle.x0_splitmix(xs, oracle.out_len(cfg, len(mel)))


@pytest.mark.parametrize("n", [2, 8, 64, 1024, 4096])
def test_c_fft_matches_numpy_dft_definition(coracle, n):
    rng = np.random.default_rng(n)
    z = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    assert np.max(np.abs(coracle.fft(z) - np.fft.fft(z))) < 1e-10 * n           # X[k] = sum x[n] e^{-2 pi i nk/N}, unnormalised
    assert np.max(np.abs(coracle.fft(z, inverse=True) - np.fft.ifft(z))) < 1e-12  # inverse scaled 1/N
    if n <= 64:  # the definition itself
        k = np.arange(n)
        dft = np.exp(-2j * np.pi * np.outer(k, k) / n) @ z
        assert np.max(np.abs(coracle.fft(z) - dft)) < 1e-10


def test_hann_variants(oracle, coracle):
    for n in (64, 1024, 4096):
        w = oracle.hann(n)
        assert np.allclose(w, np.hanning(n), atol=1e-15)             # symmetric: what phase.py:122 restates for gossp
        assert w[0] == 0 and abs(w[-1]) < 1e-15
        assert np.array_equal(w, coracle.hann(n))
        wp = oracle.hann(n, symmetric=False)
        assert np.allclose(wp, 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n) / n), atol=1e-15)
        assert np.array_equal(wp, coracle.hann(n, symmetric=False))
    w = oracle.hann(4096)                                             # sum of w^2 over the hops
    s = sum(np.pad(w * w, (k * 1280, 4 * 1280 - k * 1280))[4096:4096 + 1280] for k in range(4))
    assert 1.190 < s.min() and s.max() < 1.211


@pytest.mark.parametrize("K,fmax,H,counts", [(192, 16000.0, 2048, (1856, 191, 1)), (80, 8000.0, 512, (432, 79, 1)), (160, 8000.0, 1024, (864, 159, 1))])
def test_undomel_table_branch_counts(oracle, K, fmax, H, counts):
    rows = oracle.undomel_rows(H, K, 0.0, fmax)
    copy = sum(1 for lo, m, hi in rows if int(lo) == int(hi))
    interp = sum(1 for lo, m, hi in rows if int(lo) + 1 == int(hi) and int(hi) < K)
    assert (copy, interp, H - copy - interp) == counts
    lo, m, hi = rows[-1]
    assert (int(lo), int(hi)) == (K - 1, K)                           # top bin: the "else" branch, sum of 1 band / 2
    mel = np.zeros((K, 2)); mel[K - 1] = 3.0
    assert oracle.undomel(H, K, mel, 0.0, fmax)[H - 1, 0] == 1.5


def test_undomel_negative_zero_quirk(oracle):
    """With MelFmin > 0 the first bin's mel coordinate falls in (-1, 0): Go's Modf gives -0.0, which is NOT < 0, so the
    reference falls to the "else" branch with an empty sum divided by inhi - inlo + 1 = 0 -> NaN (mel/impl.go:355-375)."""
    rows = oracle.undomel_rows(512, 80, 50.0, 7600.0)
    assert rows[0] == (0.0, 0.0, 0.0) and rows[1] == (0.0, 0.0, 0.0)  # coordinate <= -1: zeroed -> copies band 0
    lo, m, hi = rows[2]
    assert lo == 0 and math.copysign(1.0, lo) == -1.0 and m < 0 and hi == -1.0
    out = oracle.undomel(512, 80, np.arange(1.0, 161.0).reshape(80, 2), 50.0, 7600.0)
    assert out[0, 0] == 1.0 and out[1, 1] == 2.0 and np.isnan(out[2, 0]) and np.all(np.isfinite(out[3:]))


def test_tables_match_golden(oracle):
    for K, fmin, fmax, H in [(192, 0.0, 16000.0, 2048), (80, 0.0, 8000.0, 512), (160, 0.0, 8000.0, 1024), (80, 50.0, 7600.0, 512)]:
        assert np.array_equal(np.array(oracle.undomel_rows(H, K, fmin, fmax)), GOLD["undomel_%d_%g_%g_%d" % (K, fmin, fmax, H)], equal_nan=True)
        assert np.array_equal(np.array(oracle.domel_rows(H, K, fmin, fmax)), GOLD["domel_%d_%g_%g_%d" % (K, fmin, fmax, H)], equal_nan=True)


@pytest.mark.parametrize("name", sorted(CASES))
def test_from_mel_py_c_and_golden_agree(oracle, coracle, name):
    for iters in CASES[name][4]:
        cfg, mel, x0 = case_inputs(oracle, name, iters)
        inp = mel.copy()
        a = oracle.from_mel(cfg, inp, x0)
        assert np.allclose(inp, np.exp(mel), rtol=1e-15)             # in-place exp side effect (mel/mel.go:143)
        b, mel_c = coracle.from_mel(cfg, mel, x0)
        assert np.allclose(mel_c, np.exp(mel), rtol=1e-14)
        gold = GOLD["%s_i%d_wave" % (name, iters)]
        assert a.shape == gold.shape == (oracle.out_len(cfg, len(mel)),)
        assert rms(a - b) < 1e-10 * max(1.0, rms(a))                 # two independent restatements
        assert rms(a - gold) < 1e-12 * max(1.0, rms(a))              # frozen outputs
        if iters == 0:
            assert np.array_equal(a, x0) and np.array_equal(b, x0)    # I = 0 returns the init (mel.go:85)


def test_literal_form_equals_real_fft_form(oracle):
    """The reference's N-point complex loop (abs/atan2/rect on all bins, conjugate fill) and the half-spectrum form the
    HIP kernels implement (Y = M X/|X|, irfft) agree to rounding; only channel 0 and channel 1's last bin matter."""
    cfg, mel, x0 = case_inputs(oracle, "tts_n1024", 5)
    a = oracle.from_mel(cfg, mel.copy(), x0)
    M = oracle.target_magnitudes(cfg, np.exp(mel))
    b = oracle.griffin_lim_rfft(oracle.hann(cfg.Resolut), cfg.Window, M, 5, x0)
    assert rms(a - b) < 1e-12
    mel2 = mel.copy()
    mel2[:, 1] = np.random.default_rng(0).uniform(-5, 3, len(mel))      # scramble channel 1 ...
    last = np.arange(cfg.NumMels - 1, len(mel), cfg.NumMels)
    mel2[last, 1] = mel[last, 1]                                        # ... except the band feeding bin N/2
    assert rms(oracle.from_mel(cfg, mel2, x0) - a) < 1e-12


def test_tune_and_negative_magnitudes(oracle, coracle):
    cfg, mel, x0 = case_inputs(oracle, "toy_n64", 3)
    cfg.TuneMul, cfg.TuneAdd = -2.0, 0.5                              # magnitudes go through cmplx.Abs (mel.go:99)
    a = oracle.from_mel(cfg, mel.copy(), x0)
    b, _ = coracle.from_mel(cfg, mel, x0)
    assert rms(a - b) < 1e-12 and np.all(np.isfinite(a))
    M = oracle.target_magnitudes(cfg, np.exp(mel))
    assert np.all(M >= 0)


def test_hann_variant_is_the_top_parity_risk(oracle):
    cfg, mel, x0 = case_inputs(oracle, "towav_n4096", 2)
    a = oracle.from_mel(cfg, mel.copy(), x0)
    b = oracle.from_mel(cfg, mel.copy(), x0, window=oracle.hann(4096, symmetric=False))
    assert 1e-5 < rms(a - b) < 1e-2                                   # ~2.4e-4 at 2 iterations


def test_error_cases_mirror_reference_panics(oracle, coracle):
    cfg = oracle.towav_config()
    with pytest.raises(IndexError):                                   # banner PNG geometry: 183*80 pairs, K = 192
        oracle.from_mel(cfg, np.zeros((183 * 80, 2)), np.zeros(1))
    with pytest.raises(IndexError):
        oracle.from_mel(cfg, np.zeros((0, 2)), np.zeros(0))
    with pytest.raises(IndexError):
        coracle.from_mel(cfg, np.zeros((183 * 80, 2)), np.zeros(1))


@pytest.mark.parametrize("n,hop,want", [(441000, 1280, 441599), (19200, 1280, 19200), (19201, 1280, 20479), (100, 256, 3839), (3840, 256, 3840), (0, 256, 3839)])
def test_pad_and_is_padded(oracle, coracle, n, hop, want):
    assert len(oracle.pad(np.zeros(n), hop)) == want                  # mel/impl.go:429-455
    assert coracle.lib.oracle_pad_len(n, hop) == want
    assert oracle.is_padded(n, want, hop) and not oracle.is_padded(n, want + 1, hop)


@pytest.mark.parametrize("name", ["tts_n1024", "towav_n4096"])
def test_to_mel_py_c_golden(oracle, coracle, name):
    cfg = oracle.MelConfig(**CASES[name][0])
    sig = oracle.config2_signal(seconds=0.5)
    a = oracle.to_mel(cfg, sig)
    b = coracle.to_mel(cfg, sig)
    F = oracle.num_frames(len(oracle.pad(sig, cfg.Window)), cfg.Resolut, cfg.Window)
    assert a.shape == b.shape == (F * cfg.NumMels, 2)
    assert np.max(np.abs(a - b)) < 1e-9
    assert np.max(np.abs(a - GOLD[name + "_tomel"])) < 1e-10
    assert a.min() >= math.log(1e-5) - 1e-12                          # ln floor (impl.go:413-416)


def test_splitmix_generator_fixture(oracle):
    assert np.array_equal(oracle.x0_splitmix(1, 8), GOLD["x0_seed1_first8"])
    x = oracle.x0_splitmix(123, 100000)
    assert x.min() >= 0 and x.max() < 1 and abs(x.mean() - 0.5) < 0.01
    assert np.array_equal(oracle.x0_splitmix(123, 10, offset=5), x[5:15])
