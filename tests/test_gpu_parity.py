"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle on the same seeded inputs.

Tolerances: the Griffin-Lim waveform must match the fp64 oracle within 1e-4 RMS (BASELINE.json north_star); the FFT
hooks within 2e-6 relative to the frame's largest bin (fp32 arithmetic)."""
import math

import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("fp32_start_mode")]   # these tests target the fp32 kernels unless they select a mode (conftest.py)

RMS_TOL = 1e-4  # north_star: "matching the reference FromMel output sample-for-sample within 1e-4 RMS"

GEOMS = {
    "T": dict(NumMels=192, MelFmax=16000.0, Window=1280, Resolut=4096),   # cmd/towav/main.go:30-36
    "S": dict(NumMels=80, MelFmax=8000.0, Window=256, Resolut=1024),      # BASELINE config 4
    "D": dict(NumMels=160, MelFmax=8000.0, Window=256, Resolut=2048),     # NewMel defaults mel/mel.go:30-41
    "Q": dict(NumMels=64, MelFmax=8000.0, Window=160, Resolut=512),       # 16 kHz speech front end: 32 ms frames, 10 ms hop
    "E": dict(NumMels=40, MelFmax=4000.0, Window=80, Resolut=256),        # 8 kHz telephony: 32 ms frames, 10 ms hop
}


def rms(a):
    return float(np.sqrt(np.mean(np.square(a))))


def make_mel(geom, iters, path=0, **kw):
    """path: 0 = library's choice, 1 = force the generic kernels, 2 = require the fused kernel (hop = SH*N/16)."""
    import gomel_amd
    m = gomel_amd.NewMel()
    for k, v in dict(GEOMS[geom], GriffinLimIterations=iters, **kw).items():
        setattr(m, k, v)
    if path:
        m.engine().set_path(path)
    return m


def ocfg(oracle, geom, iters, **kw):
    return oracle.MelConfig(**dict(GEOMS[geom], GriffinLimIterations=iters, **kw))


@pytest.mark.parametrize("geom", ["T", "S", "D", "Q", "E"])
def test_rfft_irfft_hooks(geom):
    m = make_mel(geom, 1)
    e = m.engine()
    N = m.Resolut
    rng = np.random.default_rng(1)
    frames = rng.standard_normal((7, N)).astype(np.float32)
    frames[3] = 0.0
    frames[3, 5] = 1.0          # a shifted impulse: flat magnitude, linear phase
    frames[4] = 1.0             # DC only
    frames[5] = np.cos(np.pi * np.arange(N))  # Nyquist only
    got = e.rfft_frames(frames)
    ref = np.fft.rfft(frames.astype(np.float64), axis=1)
    scale = np.abs(ref).max(axis=1, keepdims=True)
    assert np.max(np.abs(got - ref) / scale) < 2e-6
    back = e.irfft_frames(ref)
    assert np.max(np.abs(back - frames)) < 5e-6 * max(1.0, float(np.abs(frames).max()))


@pytest.mark.parametrize("geom", ["T", "S", "D", "Q", "E"])
def test_target_magnitudes(oracle, geom):
    m = make_mel(geom, 1, TuneMul=1.25, TuneAdd=0.01)
    cfg = ocfg(oracle, geom, 1, TuneMul=1.25, TuneAdd=0.01)
    mel = oracle.random_logmel(5, cfg.NumMels, seed=11)
    want = oracle.target_magnitudes(cfg, np.exp(mel)) / cfg.Resolut
    got = m.engine().target_magnitudes(mel)
    assert got.shape == want.shape
    assert np.max(np.abs(got - want) / (np.abs(want) + 1e-30)) < 2e-7


@pytest.mark.parametrize("path", [2, 1])
@pytest.mark.parametrize("geom,frames", [("T", 4), ("T", 9), ("S", 8), ("S", 3), ("D", 6), ("T", 1), ("T", 40), ("S", 70), ("D", 33), ("Q", 5), ("Q", 90), ("Q", 1), ("E", 7), ("E", 130), ("E", 1)])
@pytest.mark.parametrize("iters", [0, 1, 2, 60])
def test_from_mel_matches_oracle(oracle, geom, frames, iters, path):
    if iters == 60 and frames > 40:
        iters = 20   # keep the oracle side of the long cases to seconds
    m = make_mel(geom, iters, path)
    cfg = ocfg(oracle, geom, iters)
    mel = oracle.random_logmel(frames, cfg.NumMels, seed=100 + frames)
    L = oracle.out_len(cfg, len(mel))
    x0 = oracle.x0_splitmix(7, L)
    want = oracle.from_mel(cfg, mel.copy(), x0)
    inp = mel.copy()
    got = m.FromMel(inp, x0=x0)
    assert got.shape == (L,)
    assert np.all(np.isfinite(got))
    if iters == 0:
        assert np.array_equal(got, x0)          # mel.go:85 never runs: the init comes back untouched
    else:
        assert rms(got - want) < RMS_TOL, (rms(got - want), rms(want))
    assert np.allclose(inp, np.exp(mel), rtol=1e-15)  # FromMel leaves the caller's slice exp'd (mel.go:143)


def test_seed_generator_matches_documented_splitmix(oracle):
    m = make_mel("S", 0)
    mel = oracle.random_logmel(4, 80, seed=1)
    got = m.FromMel(mel.copy(), seed=12345)
    assert np.array_equal(got, oracle.x0_splitmix(12345, len(got)))
    m2 = make_mel("S", 2)
    a = m2.FromMel(mel.copy(), seed=99)
    b = m2.FromMel(mel.copy(), x0=oracle.x0_splitmix(99, len(a)))
    assert rms(a - b) < 1e-6


@pytest.mark.parametrize("path", [2, 1])
def test_ragged_batch_equals_single_calls(oracle, path):
    m = make_mel("S", 3, path)
    cfg = ocfg(oracle, "S", 3)
    fr = [3, 17, 1, 8, 40]
    mels = [oracle.random_logmel(f, 80, seed=i) for i, f in enumerate(fr)]
    x0s = [oracle.x0_splitmix(50 + i, oracle.out_len(cfg, len(mm))) for i, mm in enumerate(mels)]
    x0s[2] = None
    outs = m.FromMelBatch([mm.copy() for mm in mels], x0s, seeds=[0, 0, 77, 0, 0])
    for i, (mm, o) in enumerate(zip(mels, outs)):
        x0 = x0s[i] if x0s[i] is not None else oracle.x0_splitmix(77, oracle.out_len(cfg, len(mm)))
        want = oracle.from_mel(cfg, mm.copy(), x0)
        assert rms(o - want) < RMS_TOL


def test_injected_window_and_periodic_hann(oracle):
    m = make_mel("S", 2)
    m.HannPeriodic = True
    cfg = ocfg(oracle, "S", 2)
    mel = oracle.random_logmel(6, 80, seed=5)
    x0 = oracle.x0_splitmix(3, oracle.out_len(cfg, len(mel)))
    want = oracle.from_mel(cfg, mel.copy(), x0, window=oracle.hann(1024, symmetric=False))
    got = m.FromMel(mel.copy(), x0=x0)
    assert rms(got - want) < RMS_TOL
    e = make_mel("S", 2).engine()
    w = np.sqrt(oracle.hann(1024))
    e.set_window(w)
    want = oracle.from_mel(cfg, mel.copy(), x0, window=w)
    assert rms(e.from_mel(mel, x0=x0) - want) < RMS_TOL


def test_error_statuses():
    import gomel_amd
    m = make_mel("T", 2)
    with pytest.raises(gomel_amd.GomelError) as ei:   # 183*80 pairs with K=192: the reference panics
        m.FromMel(np.zeros((183 * 80, 2)))
    assert ei.value.status == 3
    with pytest.raises(gomel_amd.GomelError) as ei:
        m.FromMel(np.zeros((0, 2)))
    assert ei.value.status == 4
    for bad in (65537, 70000, 131072):   # above 65536 (everything up to there the generic kernels take: gl_any.hpp to 8192, gl_big.hpp beyond)
        m.Resolut = bad
        with pytest.raises(gomel_amd.GomelError) as ei:
            m.FromMel(np.zeros((192, 2)))
        assert ei.value.status == 2


def test_fused_segments_are_invisible(oracle, monkeypatch):
    """The fused kernel splits utterances into segments of consecutive frames; any split must give the generic path's
    answer (segment tails are exchanged through the tail buffers).  GOMEL_SEG_FRAMES is the library's tuning knob."""
    cfg = ocfg(oracle, "T", 4)
    fr = [37, 5, 64, 3, 11]
    mels = [oracle.random_logmel(f, 192, seed=20 + i) for i, f in enumerate(fr)]
    x0s = [oracle.x0_splitmix(i, oracle.out_len(cfg, len(mm))) for i, mm in enumerate(mels)]
    ref = make_mel("T", 4, path=1).FromMelBatch([mm.copy() for mm in mels], x0s)
    for seg in ("3", "4", "7", "64"):
        monkeypatch.setenv("GOMEL_SEG_FRAMES", seg)
        got = make_mel("T", 4, path=2).FromMelBatch([mm.copy() for mm in mels], x0s)
        for a, b in zip(got, ref):
            assert rms(a - b) < 2e-6, seg
    want = oracle.from_mel(cfg, mels[0].copy(), x0s[0])
    assert rms(ref[0] - want) < RMS_TOL


@pytest.mark.parametrize("hop", [256, 512, 1024, 2048, 1280, 768, 100, 275, 441, 1, 4095])
def test_hops_fused_and_generic(oracle, hop):
    """hop = SH*N/16 with SH in {1,2,4,5,8} takes the fused kernel, every other hop below N its any-hop form (768 = 3 rows; the odd
    hops 275 = 12.5 ms at 22.05 kHz and 441 = 10 ms at 44.1 kHz; N-1), the rest -- hop 1 and hop 100, whose segments would need
    more than 31 frames of lead -- the generic kernels."""
    kw = dict(NumMels=192, MelFmax=16000.0, Window=hop, Resolut=4096, GriffinLimIterations=3)
    import gomel_amd
    m = gomel_amd.NewMel()
    for k, v in kw.items():
        setattr(m, k, v)
    cfg = oracle.MelConfig(**kw)
    if hop < 4096 and 4095 // hop <= 31:
        m.engine().set_path(2)                  # must be accepted: a fused kernel takes these hops (and runs the 21 frames below)
    else:
        with pytest.raises(gomel_amd.GomelError):
            m.engine().set_path(2)              # served by the generic kernels only
    mel = oracle.random_logmel(21, 192, seed=hop)
    x0 = oracle.x0_splitmix(hop, oracle.out_len(cfg, len(mel)))
    want = oracle.from_mel(cfg, mel.copy(), x0)
    got = m.FromMel(mel.copy(), x0=x0)
    assert rms(got - want) < RMS_TOL * max(1.0, rms(want))
    m.Precision = 64                            # the fp64 kernels gather the same (possibly odd-aligned) frames
    assert rms(m.FromMel(mel.copy(), x0=x0) - want) < 1e-9 * max(1.0, rms(want))


@pytest.mark.parametrize("path", [2, 1])
@pytest.mark.parametrize("geom", ["T", "S"])
def test_zero_signal_takes_the_phase_zero_rule(oracle, geom, path):
    """x0 = 0 makes every re-STFT bin exactly zero: cmplx.Phase(0) = 0, so the reference imposes Y = M (mel.go:100-101).
    In the kernels this is the rare wave-uniform branch of the spectral step."""
    # (one iteration: after it the signal is a train of symmetric pulses with almost purely real spectra, whose phases sit on
    # the 0/pi knife-edge -- fp32 and fp64 then part ways by ~1e-3 relative, which says nothing about the rule under test)
    m = make_mel(geom, 1, path)
    cfg = ocfg(oracle, geom, 1)
    mel = oracle.random_logmel(7, cfg.NumMels, seed=2)
    L = oracle.out_len(cfg, len(mel))
    x0 = np.zeros(L)
    want = oracle.from_mel(cfg, mel.copy(), x0)
    got = m.FromMel(mel.copy(), x0=x0)
    assert rms(want) > 1e-3 and rms(got - want) < 1e-6 * rms(want) + 1e-7
    # one silent stretch inside an otherwise random signal
    m = make_mel(geom, 2, path)
    cfg = ocfg(oracle, geom, 2)
    mel = oracle.random_logmel(9, cfg.NumMels, seed=4)
    L = oracle.out_len(cfg, len(mel))
    x0 = oracle.x0_splitmix(5, L)
    x0[2 * cfg.Window: 2 * cfg.Window + cfg.Resolut] = 0.0          # frame 2 sees only zeros
    assert rms(m.FromMel(mel.copy(), x0=x0) - oracle.from_mel(cfg, mel.copy(), x0)) < RMS_TOL


def test_extreme_magnitudes_and_negative_tune(oracle):
    """Huge and tiny target magnitudes (moduli outside v_rsq's comfortable range) and TuneMul < 0 (targets go through
    cmplx.Abs, mel.go:99)."""
    for scale, tol in ((30.0, 1e-4), (-30.0, 1e-4)):
        m = make_mel("S", 2)
        cfg = ocfg(oracle, "S", 2)
        mel = oracle.random_logmel(6, 80, seed=8) + scale           # magnitudes * e^+-30
        x0 = oracle.x0_splitmix(3, oracle.out_len(cfg, len(mel))) * (1e-12 if scale < 0 else 1e12)
        want = oracle.from_mel(cfg, mel.copy(), x0)
        got = m.FromMel(mel.copy(), x0=x0)
        assert np.all(np.isfinite(got)) and rms(got - want) < tol * rms(want)
    m = make_mel("T", 3, TuneMul=-2.0, TuneAdd=0.5)
    cfg = ocfg(oracle, "T", 3, TuneMul=-2.0, TuneAdd=0.5)
    mel = oracle.random_logmel(5, 192, seed=9)
    x0 = oracle.x0_splitmix(4, oracle.out_len(cfg, len(mel)))
    assert rms(m.FromMel(mel.copy(), x0=x0) - oracle.from_mel(cfg, mel.copy(), x0)) < RMS_TOL


@pytest.mark.parametrize("N", [256, 512, 1024, 2048, 4096])
@pytest.mark.parametrize("SH", [1, 2, 4, 5, 8])
def test_every_fused_instantiation(oracle, N, SH):
    """k_gl_fused<N, SH> for all 25 instantiated (N, hop = SH*N/16) pairs, on a ragged batch so that streams sharing a
    workgroup walk segments of different lengths, against the oracle and against the generic kernels."""
    import gomel_amd
    hop = SH * N // 16
    K = {256: 40, 512: 64, 1024: 80, 2048: 160, 4096: 192}[N]
    kw = dict(NumMels=K, MelFmax=8000.0, Window=hop, Resolut=N, GriffinLimIterations=3)
    cfg = oracle.MelConfig(**kw)
    fr = [1, 2, 3, 5, 9, 17, 33, 4, 7, 21]
    mels = [oracle.random_logmel(f, K, seed=N + SH + i) for i, f in enumerate(fr)]
    x0s = [oracle.x0_splitmix(i, oracle.out_len(cfg, len(mm))) for i, mm in enumerate(mels)]
    outs = {}
    for path in (0, 1):
        m = gomel_amd.NewMel()
        for k, v in kw.items():
            setattr(m, k, v)
        m.engine().set_path(1 if path else 2)
        outs[path] = m.FromMelBatch([mm.copy() for mm in mels], x0s)
    for i, mm in enumerate(mels):
        want = oracle.from_mel(cfg, mm.copy(), x0s[i])
        assert rms(outs[0][i] - want) < RMS_TOL, (N, SH, i)
        assert rms(outs[1][i] - want) < RMS_TOL, (N, SH, i)


def test_automatic_path_choice(oracle):
    """Library's choice: what fits the chip at once (or twice, as whole utterances) runs the persistent kernel, a long utterance below the
    measured crossover (1536 frames at N 4096 / hop 1280) the generic kernels (every frame in parallel), batches past it the fused
    kernel; all agree with the oracle."""
    import gomel_amd
    m = make_mel("T", 2)
    e = m.engine()
    for n_utt, frames, want_fast in ((1, 300, 2), (4, 300, 2), (1, 1400, 0), (8, 300, 1)):   # 2 = the persistent kernel (tests/test_gpu_persist.py; 4 x 300: two launches)
        b = gomel_amd.Batch(e, [frames * 192] * n_utt)
        mels = [oracle.random_logmel(frames, 192, seed=u) for u in range(n_utt)]
        b.upload(mels, None, list(range(n_utt)))
        tm = b.run_timed()
        assert tm["fast_path"] == want_fast and tm["frames"] == n_utt * frames
        out = b.download()[0]
        b.close()
        cfg = ocfg(oracle, "T", 2)
        assert rms(out - oracle.from_mel(cfg, mels[0].copy(), oracle.x0_splitmix(0, len(out)))) < RMS_TOL


@pytest.mark.parametrize("geom,frames,iters", [("T", 9, 60), ("S", 24, 60), ("D", 12, 20), ("T", 3, 2), ("Q", 30, 60), ("E", 50, 60)])
def test_precise_mode_is_the_fp64_algorithm(oracle, geom, frames, iters):
    """gomel_set_precision(64): the generic kernels in fp64 end to end.  Against the fp64 oracle this is agreement between two
    implementations of the same arithmetic (different FFT factorisation, different summation order): ~1e-13."""
    m = make_mel(geom, iters)
    m.engine().set_precision(64)
    cfg = ocfg(oracle, geom, iters)
    mel = oracle.random_logmel(frames, cfg.NumMels, seed=31 + frames)
    x0 = oracle.x0_splitmix(11, oracle.out_len(cfg, len(mel)))
    want = oracle.from_mel(cfg, mel.copy(), x0)
    got = m.FromMel(mel.copy(), x0=x0)
    assert rms(got - want) < 1e-10 * max(1.0, rms(want)), rms(got - want)
    m.engine().set_precision(32)
    assert 1e-10 < rms(m.FromMel(mel.copy(), x0=x0) - want) < RMS_TOL      # back to the fp32 kernels


def test_mel_tables_fuzz_against_oracle(oracle):
    """Random configurations (bands, fmin/fmax, tune, all three FFT sizes): the host-built undomel table + k_mel_to_mag against
    the oracle's literal undomel -- including configurations whose table runs past the band slice, where the reference panics:
    there the library must return GOMEL_ERR_TABLE_RANGE and the oracle must raise."""
    import gomel_amd
    rng = np.random.default_rng(2024)
    ran = refused = 0
    for trial in range(60):
        N = int(rng.choice([256, 512, 1024, 2048, 4096]))
        K = int(rng.integers(6, 400))
        fmin = float(rng.choice([0.0, 0.0, rng.uniform(0, 400)]))
        fmax = float(rng.uniform(1500, 26000))
        add, mul = (0.0, 1.0) if trial % 3 == 0 else (float(rng.uniform(-0.05, 0.05)), float(rng.choice([-2.0, 0.5, 1.0, 3.0])))
        fields = dict(NumMels=K, MelFmin=fmin, MelFmax=fmax, Window=N // 4, Resolut=N, TuneAdd=add, TuneMul=mul, GriffinLimIterations=1)
        cfg = oracle.MelConfig(**fields)
        mel = oracle.random_logmel(3, K, seed=trial)
        try:
            want = oracle.target_magnitudes(cfg, np.exp(mel)) / N
        except (IndexError, ValueError):
            want = None
        m = gomel_amd.NewMel()
        for k, v in fields.items():
            setattr(m, k, v)
        if want is None:
            with pytest.raises(gomel_amd.GomelError) as ei:
                m.engine().target_magnitudes(mel)
            assert ei.value.status == 5, fields
            refused += 1
            continue
        got = m.engine().target_magnitudes(mel)
        bad = ~np.isfinite(want)
        assert np.array_equal(bad, ~np.isfinite(got)), fields                       # the -0.0 / empty-sum NaN quirk, where it occurs
        err = np.abs(got[~bad] - want[~bad]) / (np.abs(want[~bad]) + 1e-30)
        assert err.max() < 2e-7, (fields, float(err.max()))
        m.engine().close()
        ran += 1
    assert ran >= 30, (ran, refused)


def test_from_mel_fuzz_configs(oracle):
    """Random geometry through the whole FromMel path: any hop up to N, odd ones included (fused where hop = SH*N/16, generic elsewhere), random
    band layout, ragged frame counts, 0-3 iterations, both x0 routes."""
    import gomel_amd
    rng = np.random.default_rng(5150)
    for trial in range(40):
        N = int(rng.choice([256, 512, 1024, 2048, 4096]))
        hop = int(rng.choice([N // 16, N // 4, 5 * N // 16, N // 2, N, int(rng.integers(8, N)), 2 * int(rng.integers(8, N // 2)) + 1]))
        K = int(rng.integers(8, 256))
        I = int(rng.integers(0, 4))
        F = int(rng.integers(1, 14))
        fields = dict(NumMels=K, MelFmax=float(rng.uniform(4000, 22000)), Window=hop, Resolut=N, GriffinLimIterations=I)
        cfg = oracle.MelConfig(**fields)
        mel = oracle.random_logmel(F, K, seed=100 + trial)
        L = oracle.out_len(cfg, len(mel))
        x0 = oracle.x0_splitmix(trial, L)
        try:
            want = oracle.from_mel(cfg, mel.copy(), x0)
        except (IndexError, ValueError):
            continue                                      # table range: covered by test_mel_tables_fuzz_against_oracle
        m = gomel_amd.NewMel()
        for k, v in fields.items():
            setattr(m, k, v)
        got = m.FromMel(mel.copy(), x0=x0) if trial % 2 else m.FromMel(mel.copy(), seed=trial)
        assert got.shape == want.shape, fields
        assert np.all(np.isfinite(got)) == np.all(np.isfinite(want)), fields
        if np.all(np.isfinite(want)):
            assert rms(got - want) < RMS_TOL * max(1.0, rms(want)), (fields, F, rms(got - want))
        m.engine().close()


@pytest.mark.parametrize("N,hop,F,iters", [(4096, 1280, 12, 3), (1024, 256, 40, 60), (512, 160, 9, 2), (2048, 300, 5, 1), (256, 64, 700, 2),
                                           (1024, 256, 3, 0)])
def test_istft_entry_matches_the_literal_loop(oracle, N, hop, F, iters):
    """gomel_istft / gomel_amd.ISTFT against the oracle's literal restatement of mel/mel.go:76-139 fed a COMPLEX spectrogram
    with arbitrary phases and an arbitrary (inconsistent) upper half -- the loop must ignore both."""
    import gomel_amd
    from types import SimpleNamespace
    rng = np.random.default_rng(N + F)
    mag = rng.uniform(0.0, 2.0, size=(F, N))
    spec = mag * np.exp(1j * rng.uniform(-np.pi, np.pi, size=(F, N)))
    window = oracle.hann(N) if F % 2 else rng.uniform(0.1, 1.0, size=N)          # injected s.Window, not only Hann
    L = N + (F - 1) * hop
    x0 = oracle.x0_splitmix(3, L)
    ref_spec = spec.copy()
    want = oracle.istft_griffin_lim(window, hop, ref_spec, iters, x0)            # leaves the last phase estimate in ref_spec (mel.go:98-108)
    s = SimpleNamespace(FrameShift=hop, Window=window)
    keep = spec.copy()
    got = gomel_amd.ISTFT(s, spec, iters, x0=x0)
    if iters == 0:
        assert np.array_equal(spec, keep)                                        # the loop never ran: nothing written
    else:                                                                        # the same side effect as the reference (fp32 phases)
        assert np.max(np.abs(spec - ref_spec)) < 5e-4 * np.max(np.abs(ref_spec))
    assert got.shape == want.shape
    assert rms(got - want) < RMS_TOL * max(1.0, rms(want)), rms(got - want)
    assert rms(gomel_amd.ISTFT(s, spec, iters, seed=3) - want) < RMS_TOL * max(1.0, rms(want))   # same stream through the seed route
    if iters == 0:
        assert np.array_equal(got, x0)                                           # no iteration: the random init comes back
    with pytest.raises(gomel_amd.GomelError):
        gomel_amd.ISTFT(s, spec[:0], iters)
