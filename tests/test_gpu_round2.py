"""GPU tests added in round 2: BASELINE's configurations at FULL size against the C oracle (every utterance), the new C-ABI
entry points (spectrogram write-back of ISTFT, batched ToMel, fp32 host I/O, multi-handle batch, pinned host memory), the
host-side fixes (device restored, prefix views pinned once, table blob range check, large NumMels), and a two-process run on
the one GPU in which rank 1 builds its handle from the broadcast table blob."""
import ctypes
import math
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("fp32_start_mode")]   # these tests target the fp32 kernels unless they select a mode (conftest.py)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RMS_TOL = 1e-4          # north_star: sample-for-sample within 1e-4 RMS
AUTO_60 = 35            # gomel_auto_fp64_iterations(cfg, 60): fp64 iterations of the DEFAULT call at 60 iterations (profiles/r04/calibration/TABLE.txt)

T = dict(NumMels=192, MelFmax=16000.0, Window=1280, Resolut=4096)     # towav geometry (cmd/towav/main.go:30-37)
S = dict(NumMels=80, MelFmax=8000.0, Window=256, Resolut=1024)        # BASELINE configs[3], the TTS geometry


def rms(a):
    return float(np.sqrt(np.mean(np.square(a))))


def mel_obj(fields, **kw):
    import gomel_amd
    m = gomel_amd.NewMel()
    for k, v in dict(fields, **kw).items():
        setattr(m, k, v)
    return m


def cfg3_inputs(U, F, K):
    mels = [np.random.default_rng(u).uniform(math.log(1e-5), math.log(50.0), size=(F * K, 2)) for u in range(U)]
    return mels, [1000 + u for u in range(U)]


# ------------------------------------------------------------------------------------------------ full-size configurations
# (The full-size oracle comparisons of BASELINE configs[0..4] in the DEFAULT mode live in tests/test_gpu_configs.py since round 6 and are
# collected first; what stays here is configs[1] in the opt-in fp32 fast mode.)
def test_config2_full_size_single_utterance_fp32_fast_mode(oracle, coracle):
    """BASELINE configs[1]: one 10 s utterance @44.1 kHz (342 frames), 60 iterations, the path the library picks by itself for a batch of
    this size, in the fp32 fast mode."""
    fields = dict(T, GriffinLimIterations=60)
    m = mel_obj(fields)
    m.Precision = 32
    sig = oracle.config2_signal()
    mel = m.ToMel(sig)
    assert len(mel) == 342 * 192
    cfg = oracle.MelConfig(**fields)
    x0 = oracle.x0_splitmix(1, oracle.out_len(cfg, len(mel)))
    got = m.FromMel(mel.copy(), x0=x0)
    want, _ = coracle.from_mel(cfg, mel, x0)
    print("config 2 (342 frames x 60 iterations), fp32 fast mode vs oracle: %.2e (signal rms %.3f)" % (rms(got - want), rms(want)))
    assert rms(got - want) < RMS_TOL


# ------------------------------------------------------------------------------------------------ new entry points
@pytest.mark.parametrize("bits", [32, 64])
def test_istft_leaves_the_last_phase_estimate_in_the_spectrogram(oracle, bits):
    """mel/mel.go:98-108: ISTFT overwrites its spectrogram argument; gomel_istft_spec hands the lower half back."""
    import gomel_amd

    class STFT:
        pass
    N, hop, F, I = 1024, 256, 9, 3
    rng = np.random.default_rng(4)
    spec = (rng.uniform(0.1, 2.0, size=(F, N)) * np.exp(2j * np.pi * rng.uniform(size=(F, N)))).astype(np.complex128)
    s = STFT()
    s.FrameShift, s.Window = hop, oracle.hann(N)
    x0 = oracle.x0_splitmix(3, N + (F - 1) * hop)
    ref_spec = spec.copy()
    want = oracle.istft_griffin_lim(s.Window, hop, ref_spec, I, x0)          # mutates ref_spec like the reference
    mine = spec.copy()
    cfg = gomel_amd.GomelConfig()
    gomel_amd._lib.lib().gomel_default_config(ctypes.byref(cfg))
    cfg.resolut, cfg.window = N, hop
    e = gomel_amd.Engine(cfg)
    e.set_window(s.Window)
    e.set_precision(bits)
    wave, sp = e.istft(np.abs(spec[:, :N // 2 + 1]), I, x0=x0, spec=True)
    tol = 1e-9 if bits == 64 else 2e-4
    assert rms(wave - want) < (1e-9 if bits == 64 else RMS_TOL)
    assert np.max(np.abs(sp - ref_spec[:, :N // 2 + 1])) < tol * np.max(np.abs(ref_spec))
    # zero iterations: the reference's loop never runs, nothing is written
    w0, sp0 = e.istft(np.abs(spec[:, :N // 2 + 1]), 0, x0=x0, spec=True)
    assert sp0 is None and np.array_equal(w0, x0)
    e.close()
    if bits == 32:   # the Python mirror of the exported function mutates a complex128 array as the Go function does
        got = gomel_amd.ISTFT(s, mine, I, x0=x0)
        assert rms(got - want) < RMS_TOL
        assert np.max(np.abs(mine - ref_spec)) < 2e-4 * np.max(np.abs(ref_spec))


def test_to_mel_batch_equals_single_calls(oracle):
    m = mel_obj(S)
    sigs = [oracle.config2_signal(seconds=0.3 + 0.17 * i, seed=i) for i in range(5)] + [oracle.config2_signal(seconds=0.5)[:1000]]
    many = m.ToMelBatch(sigs)
    cfg = oracle.MelConfig(**S)
    for sig, got in zip(sigs, many):
        assert np.array_equal(got, m.ToMel(sig))
        assert np.max(np.abs(got - oracle.to_mel(cfg, sig))) < 1e-9


def test_fp32_host_io_entry(oracle):
    fields = dict(S, GriffinLimIterations=5)
    m = mel_obj(fields)
    e = m.engine()
    mels = [oracle.random_logmel(f, 80, seed=f).astype(np.float32) for f in (7, 300, 40)]
    seeds = [5, 6, 7]
    got = e.from_mel_batch_f32(mels, None, seeds)
    ref = e.from_mel_batch([mm.astype(np.float64) for mm in mels], None, seeds)       # same values, float64 interface
    for a, r in zip(got, ref):
        assert a.dtype == np.float32 and a.shape == r.shape
        assert rms(a.astype(np.float64) - r) < 1e-6
    x0 = [oracle.x0_splitmix(9, len(r)).astype(np.float32) for r in ref]
    got2 = e.from_mel_batch_f32(mels, x0, None)
    ref2 = e.from_mel_batch([mm.astype(np.float64) for mm in mels], [x.astype(np.float64) for x in x0], None)
    for a, r in zip(got2, ref2):
        assert rms(a.astype(np.float64) - r) < 1e-6
    # every precision mode takes float host buffers (round 4): float at the boundary, the mode's arithmetic inside -- the waveform is
    # the float64 interface's, rounded to float once; on the persistent / generic kernels (these three utterances) and on the fused
    # kernels (path 2: 64-bit mode, an explicit schedule, the automatic one)
    for path in (0, 2):
        e.set_path(path)
        for mode in (64, "schedule", "auto"):
            if mode == 64:
                e.set_precision(64)
            else:
                e.set_precision_schedule(2 if mode == "schedule" else "auto")
            gotm = e.from_mel_batch_f32(mels, x0, None)
            refm = e.from_mel_batch([mm.astype(np.float64) for mm in mels], [x.astype(np.float64) for x in x0], None)
            for a, r in zip(gotm, refm):
                r32 = r.astype(np.float32)
                assert a.dtype == np.float32
                if mode == 64:
                    assert np.array_equal(a, r32), (path, mode)                                   # one rounding of the same fp64 signal
                else:                                                                              # (fused schedule: x + tail summed in fp32 resp. fp64 first)
                    assert np.max(np.abs(a - r32)) <= 2 * float(np.spacing(np.abs(r32).max())), (path, mode)
    e.set_path(0)
    e.set_precision(32)
    # the generic frame lengths take float host buffers too (their kernels are fp64 inside: the values are widened / narrowed once)
    for N, hop, K in ((1000, 250, 40), (6000, 1500, 96)):
        mg = mel_obj(dict(NumMels=K, MelFmax=8000.0, Window=hop, Resolut=N, GriffinLimIterations=3))
        eg = mg.engine()
        melsg = [oracle.random_logmel(f, K, seed=f).astype(np.float32) for f in (5, 9)]
        x0g = [oracle.x0_splitmix(3 + i, N + (f - 1) * hop).astype(np.float32) for i, f in enumerate((5, 9))]
        gotg = eg.from_mel_batch_f32(melsg, x0g, None)
        refg = eg.from_mel_batch([mm.astype(np.float64) for mm in melsg], [x.astype(np.float64) for x in x0g], None)
        for a, r in zip(gotg, refg):
            assert a.dtype == np.float32 and rms(a.astype(np.float64) - r) < 1e-6 * max(1.0, rms(r))
        eg.close()


def test_multi_handle_batch_equals_one_handle(oracle):
    """gomel_from_mel_batch_multi over two handles (here: both on the one GPU) = the single-handle batch, bit for bit."""
    import gomel_amd
    fields = dict(S, GriffinLimIterations=4)
    m = mel_obj(fields)
    e0 = m.engine()
    e1 = gomel_amd.Engine(m._config(), tables=e0.export_tables())
    e2 = gomel_amd.Engine(m._config(), tables=e0.export_tables())
    mels = [oracle.random_logmel(f, 80, seed=f) for f in (12, 5, 33, 8, 21, 4, 17)]
    one = e0.from_mel_batch(mels)
    many = gomel_amd.from_mel_batch_multi([e0, e1, e2], mels)
    assert all(np.array_equal(a, b) for a, b in zip(one, many))
    more = gomel_amd.from_mel_batch_multi([e0, e1, e2], mels[:2])           # fewer utterances than handles
    assert all(np.array_equal(a, b) for a, b in zip(one[:2], more))
    e1.close()
    e2.close()


def test_pinned_host_buffers_and_prefix_views(oracle):
    """Buffers from gomel_host_alloc go through the batch call untouched by the per-call pinning; two utterances that are prefix
    views of ONE large array (same base pointer, different lengths) are registered once with the longer extent."""
    import gomel_amd
    fields = dict(S, GriffinLimIterations=2)
    m = mel_obj(fields)
    e = m.engine()
    F = 1000                                                                   # 80 k pairs = 1.25 MB: above the pinning threshold
    big = gomel_amd.host_alloc((F * 80, 2))
    big[:] = oracle.random_logmel(F, 80, seed=1)
    outs = [gomel_amd.host_alloc(e.out_len(F * 80)), gomel_amd.host_alloc(e.out_len(500 * 80))]
    pinned = e.from_mel_batch([big, big[:500 * 80]], None, [1, 2], outs)
    plain_big = np.array(big)                                                  # pageable copy
    plain = e.from_mel_batch([plain_big[:500 * 80], plain_big], None, [2, 1])  # shorter view FIRST, same base
    assert np.array_equal(pinned[0], plain[1]) and np.array_equal(pinned[1], plain[0])


def test_device_is_restored_after_every_entry_point(oracle, hip):
    """entry points used to leave the handle's device current on the calling thread.  (hipGetDevice / hipSetDevice through
    ctypes on the runtime the library runs on: the process that runs the tests does not import torch.)"""
    import gomel_amd

    def current():
        d = ctypes.c_int(-1)
        assert hip.hipGetDevice(ctypes.byref(d)) == 0
        return d.value

    n = gomel_amd._lib.lib().gomel_device_count()
    other = 1 if n > 1 else 0
    assert hip.hipSetDevice(0) == 0
    m = mel_obj(dict(S, GriffinLimIterations=2))
    m.Device = other
    e = m.engine()
    mel = oracle.random_logmel(6, 80, seed=0)
    e.from_mel(mel, seed=1)
    assert current() == 0
    e.to_mel(oracle.config2_signal(seconds=0.3))
    assert current() == 0
    b = gomel_amd.Batch(e, [len(mel)])
    b.upload([mel], None, [1])
    b.run()
    b.download()
    b.close()
    assert current() == 0
    assert gomel_amd._lib.lib().gomel_device_synchronize(other) == 0 and current() == 0      # round 6's entry point keeps the rule too
    p = ctypes.c_void_p()
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    assert hip.hipMalloc(ctypes.byref(p), 64) == 0 and hip.hipFree(p) == 0                   # the thread's own allocations still land on ITS device
    assert current() == 0


def test_many_mel_bands_and_corrupted_table_blob(oracle):
    """K = 1024 bands at N 4096: the exp() rows of k_mel_to_mag no longer fit eight to a block; and a table blob
    whose rows point outside the band range is refused instead of becoming device-side indices."""
    import gomel_amd
    fields = dict(NumMels=1024, MelFmax=16000.0, Window=1280, Resolut=4096, GriffinLimIterations=1)
    m = mel_obj(fields)
    e = m.engine()
    mel = oracle.random_logmel(3, 1024, seed=3)
    cfg = oracle.MelConfig(**fields)
    want = oracle.target_magnitudes(cfg, np.exp(mel)) / 4096.0
    got = e.target_magnitudes(mel)
    assert np.max(np.abs(got - want) / (np.abs(want) + 1e-30)) < 1e-6
    x0 = oracle.x0_splitmix(2, oracle.out_len(cfg, len(mel)))
    assert rms(e.from_mel(mel, x0=x0) - oracle.from_mel(cfg, mel.copy(), x0)) < RMS_TOL
    blob = bytearray(e.export_tables())
    # first undomel row (bin 0 sums bands [0, 4) here): lo, hi sit right behind the header (48 bytes) and the window (N doubles)
    off = 48 + 4096 * 8
    blob[off + 4:off + 8] = (5000).to_bytes(4, "little")                       # hi = 5000 > NumMels: the sum would run off the bands
    with pytest.raises(gomel_amd.GomelError) as ei:
        gomel_amd.Engine(m._config(), tables=bytes(blob))
    assert ei.value.status == 10


@pytest.mark.parametrize("N,K,fmax", [(4096, 192, 16000.0), (2048, 160, 8000.0), (1024, 80, 8000.0), (512, 40, 8000.0), (256, 24, 8000.0)])
@pytest.mark.parametrize("SH", [1, 2, 4, 5, 8])
def test_fused_fp64_kernel_matches_the_oracle(oracle, N, K, fmax, SH):
    """64-bit mode through the fused fp64 kernel (forced: gomel_set_path 2) on every instantiation: ragged batch with segments of
    every kind (single-segment utterances, first / middle / last segments, a one-frame utterance), 5 iterations, < 1e-9 from
    the fp64 oracle; and bit-identical run to run."""
    import gomel_amd
    hop = SH * N // 16
    fields = dict(NumMels=K, MelFmax=fmax, Window=hop, Resolut=N, GriffinLimIterations=5)
    cfg = oracle.MelConfig(**fields)
    m = mel_obj(fields)
    e = m.engine()
    e.set_precision(64)
    e.set_path(2)
    frames = [1, 2, 3, 17, 40, 66, 9] if N >= 2048 else [1, 2, 3, 17, 40, 150, 9]
    mels = [oracle.random_logmel(f, K, seed=10 * SH + i) for i, f in enumerate(frames)]
    seeds = list(range(len(mels)))
    b = gomel_amd.Batch(e, [len(mm) for mm in mels])
    b.upload(mels, None, seeds)
    tm = b.run_timed()
    got = b.download()
    b.run()
    again = b.download()
    b.close()
    assert tm["fast_path"] == 1
    assert all(np.array_equal(a, c) for a, c in zip(got, again))
    for mm, sd, g in zip(mels, seeds, got):
        want = oracle.from_mel(cfg, mm.copy(), oracle.x0_splitmix(sd, len(g)))
        assert rms(g - want) < 1e-9 * max(1.0, rms(want)), (N, SH, len(mm) // K, rms(g - want))


def test_uneven_chunk_plans_recycle_buffers_correctly(oracle, monkeypatch):
    """The pipeline's chunks are not equal (half-size first and last by default; any plan through GOMEL_CHUNK_SHARES): buffers of
    a finished chunk serve a later, differently sized one while two other chunks are in flight.  Ten uneven chunks, twice, against
    the default plan and against one resident batch."""
    import gomel_amd
    fields = dict(S, GriffinLimIterations=3)
    m = mel_obj(fields)
    e = m.engine()
    rng = np.random.default_rng(5)
    frames = rng.integers(20, 400, size=120).tolist()
    mels = [oracle.random_logmel(f, 80, seed=900 + i) for i, f in enumerate(frames)]
    seeds = list(range(120))
    base = e.from_mel_batch(mels, None, seeds)
    for plan in ("1,3,1,2,1,1,2,1,1,1", "5,1,1,1,1,1,5", "1,1"):
        monkeypatch.setenv("GOMEL_CHUNK_SHARES", plan)
        got = e.from_mel_batch(mels, None, seeds)
        again = e.from_mel_batch(mels, None, seeds)
        monkeypatch.delenv("GOMEL_CHUNK_SHARES")
        assert all(np.array_equal(a, b) for a, b in zip(got, again)), plan
        assert all(a.shape == b.shape and rms(a - b) < 1e-5 for a, b in zip(got, base)), plan      # other segmentation: rounding only
    b = gomel_amd.Batch(e, [len(mm) for mm in mels])
    b.upload(mels, None, seeds)
    b.run()
    for a, r in zip(base, b.download()):
        assert rms(a - r) < 1e-5
    b.close()


def test_chunked_pipeline_on_a_ragged_batch(oracle):
    """gomel_from_mel_batch cuts big calls into chunks that cycle through three sets of device buffers (upload c+1 / compute c /
    download c-1): a ragged 300-utterance batch (3 ... 700 frames, 8+ chunks), explicit x0 for some utterances, against the same
    utterances computed one batch-resident call at a time."""
    import gomel_amd
    fields = dict(S, GriffinLimIterations=4)
    m = mel_obj(fields)
    e = m.engine()
    rng = np.random.default_rng(11)
    frames = rng.integers(3, 700, size=300).tolist()
    mels = [oracle.random_logmel(f, 80, seed=i) for i, f in enumerate(frames)]
    x0s = [oracle.x0_splitmix(77 + i, e.out_len(len(mm))) if i % 3 == 0 else None for i, mm in enumerate(mels)]
    seeds = [500 + i for i in range(len(mels))]
    got = e.from_mel_batch(mels, x0s, seeds)
    assert [len(g) for g in got] == [e.out_len(len(mm)) for mm in mels] and all(np.isfinite(g).all() for g in got)
    again = e.from_mel_batch(mels, x0s, seeds)
    assert all(np.array_equal(a, b) for a, b in zip(got, again))                # deterministic, buffers recycled correctly
    for lo in range(0, 300, 60):                                                # reference: small resident batches, no chunking
        b = gomel_amd.Batch(e, [len(mm) for mm in mels[lo:lo + 60]])
        b.upload(mels[lo:lo + 60], x0s[lo:lo + 60], seeds[lo:lo + 60])
        b.run()
        ref = b.download()
        b.close()
        for a, r in zip(got[lo:lo + 60], ref):
            assert rms(a - r) < 1e-5                                            # same kernels, other segmentation: rounding only
    cfg = oracle.MelConfig(**fields)
    for i in (0, 1, 150, 299):
        x0 = x0s[i] if x0s[i] is not None else oracle.x0_splitmix(seeds[i], len(got[i]))
        assert rms(got[i] - oracle.from_mel(cfg, mels[i].copy(), x0)) < RMS_TOL


@pytest.mark.stress
def test_concurrent_batch_calls_on_one_handle(oracle):
    """Four host threads push chunked batch calls through ONE handle at the same time (the handle's mutex is taken per stage,
    batches are created and destroyed between stages): every call must come back complete and equal to the serial result."""
    import threading
    fields = dict(S, GriffinLimIterations=3)
    m = mel_obj(fields)
    e = m.engine()
    sets = [[oracle.random_logmel(150 + 10 * t + j, 80, seed=100 * t + j) for j in range(120)] for t in range(4)]
    serial = [e.from_mel_batch(ms, None, list(range(len(ms)))) for ms in sets]
    out, errs = [None] * 4, []

    def work(t):
        try:
            out[t] = e.from_mel_batch(sets[t], None, list(range(len(sets[t]))))
        except Exception as ex:  # noqa: BLE001
            errs.append(repr(ex))

    ts = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    for t in range(4):
        assert all(np.array_equal(a, b) for a, b in zip(out[t], serial[t]))


# ------------------------------------------------------------------------------------------------ any Resolut (the reference: go-dsp radix-2 / Bluestein)
@pytest.mark.parametrize("N,hop,K,fmax", [(8192, 2048, 96, 16000.0), (3000, 750, 64, 8000.0), (1000, 250, 40, 8000.0), (8, 3, 2, 8000.0), (64, 20, 8, 8000.0),
                                           (1001, 250, 40, 8000.0), (4095, 1024, 96, 16000.0), (63, 20, 8, 8000.0),
                                           (6000, 1500, 96, 16000.0), (4097, 1024, 96, 16000.0), (8191, 2000, 96, 16000.0), (5000, 1250, 80, 8000.0),   # Bluestein in two halves
                                           (16384, 4096, 128, 16000.0), (10000, 2500, 96, 16000.0), (12001, 3000, 96, 16000.0), (8193, 2048, 96, 16000.0),   # above 8192: gl_big.hpp, through HBM
                                           (40000, 9000, 160, 20000.0), (65536, 16384, 192, 22000.0)])
def test_any_frame_length_from_mel_to_mel_istft(oracle, N, hop, K, fmax):
    """The reference takes any frame length (fft.FFTReal / fft.IFFT, mel/mel.go:95,116); outside 256 ... 4096 powers of two the
    generic kernels (gl_any.hpp: radix-2 in LDS, Bluestein for the rest, fp64; above 8192 gl_big.hpp: Bluestein through HBM, up to
    65536) carry FromMel, ToMel and ISTFT."""
    import gomel_amd
    fields = dict(NumMels=K, MelFmax=fmax, Window=hop, Resolut=N)
    cfg = oracle.MelConfig(GriffinLimIterations=3, **fields)
    m = mel_obj(fields, GriffinLimIterations=3)
    F = 7
    mel = oracle.random_logmel(F, K, seed=N)
    x0 = oracle.x0_splitmix(N, oracle.out_len(cfg, len(mel)))
    try:
        want = oracle.from_mel(cfg, mel.copy(), x0)
    except IndexError:
        want = None                                                            # this geometry's undomel table runs off the bands: the reference panics
    if want is None:
        with pytest.raises(gomel_amd.GomelError):
            m.FromMel(mel.copy(), x0=x0)
    else:
        got = m.FromMel(mel.copy(), x0=x0)
        assert got.shape == want.shape and rms(got - want) < 1e-9 * max(1.0, rms(want)), rms(got - want)
        outs = m.engine().from_mel_batch([mel, mel[:3 * K]], [x0, x0[:N + 2 * hop]], None)    # the batch entry point takes the same path
        assert rms(outs[0] - want) < 1e-9 * max(1.0, rms(want))
    sig = oracle.config2_signal(seconds=(15 * hop + 4 * hop + N) / 44100.0)
    try:
        wantm = oracle.to_mel(cfg, sig)
    except IndexError:
        wantm = None
    if wantm is not None:
        gotm = m.ToMel(sig)
        assert gotm.shape == wantm.shape and np.max(np.abs(gotm - wantm)) < 1e-8

    class STFT:
        pass
    s = STFT()
    s.FrameShift, s.Window = hop, oracle.hann(N)
    rng = np.random.default_rng(N)
    spec = rng.uniform(0.1, 2.0, size=(5, N)) * np.exp(2j * np.pi * rng.uniform(size=(5, N)))
    xi = oracle.x0_splitmix(2, N + 4 * hop)
    want_spec = spec.copy()
    wanti = oracle.istft_griffin_lim(s.Window, hop, want_spec, 2, xi)          # (mutates want_spec, as the reference does, mel/mel.go:98-108)
    goti = gomel_amd.ISTFT(s, [list(r) for r in spec], 2, x0=xi)               # a list cannot be written to: plain gomel_istft
    assert rms(goti - wanti) < 1e-9 * max(1.0, rms(wanti))
    arr = spec.copy()                                                          # a complex128 ndarray: the same data, same waveform, PLUS the write-back
    gota = gomel_amd.ISTFT(s, arr, 2, x0=xi)
    assert np.array_equal(gota, goti)
    assert np.max(np.abs(arr - want_spec)) < 1e-9 * np.max(np.abs(want_spec))


# ------------------------------------------------------------------------------------------------ two processes, one GPU, real engines
def _rank_main(rep, rank, world, port):
    """One rank in a child process (tests/_children.py has imported torch)."""
    import torch.distributed as dist
    import gomel_oracle as go
    import gomel_amd
    from gomel_amd import dist as gd
    rep.stage("init_process_group")
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        m = gomel_amd.NewMel()
        for k, v in dict(S, GriffinLimIterations=6).items():
            setattr(m, k, v)
        m.Device = 0
        made = {}

        def factory(tables=None):
            made["from_blob"] = tables is not None
            return gomel_amd.Engine(m._config(), tables=tables)

        rep.stage("table broadcast")
        eng = gd.create_engine_broadcast(factory, rank, world)
        frames = [30, 7, 64, 12, 45, 9, 21, 50]
        mels = [go.random_logmel(f, 80, seed=i) for i, f in enumerate(frames)]
        rep.stage("sharded FromMel")
        outs = gd.from_mel_sharded(lambda ms, xs, ss: eng.from_mel_batch(ms, xs, ss), mels, None, None, rank, world)
        rep.result((made["from_blob"], [o.tobytes() for o in outs], eng.export_tables()))
        eng.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.torch_child
@pytest.mark.guard(700)
def test_two_ranks_on_one_gpu_with_real_engines(oracle):
    """world_size 2, gloo for the collectives, both ranks on cuda:0 with REAL handles: rank 1 builds its handle from the table blob
    that crossed the process boundary (gomel_create_from_tables), each rank computes its contiguous shard, and the gathered
    waveforms equal the single-process batch bit for bit."""
    import _children
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    got = _children.run_children(_rank_main, [(r, 2, port) for r in range(2)])
    assert got[0][0] is False and got[1][0] is True                            # rank 1's handle came from the broadcast
    assert got[0][2] == got[1][2]                                               # bit-identical tables on both ranks
    assert got[0][1] == got[1][1]                                               # every rank holds the full gathered result
    m = mel_obj(dict(S, GriffinLimIterations=6))
    frames = [30, 7, 64, 12, 45, 9, 21, 50]
    mels = [oracle.random_logmel(f, 80, seed=i) for i, f in enumerate(frames)]
    single = m.engine().from_mel_batch(mels, None, list(range(len(mels))))
    assert [o.tobytes() for o in single] == got[0][1]


ANY_HOPS = [(4096, 882, 96), (4096, 1000, 96), (4096, 2050, 96), (4096, 3000, 96), (2048, 276, 80), (2048, 300, 80), (2048, 130, 80), (1024, 300, 80),
            (1024, 200, 80), (1024, 600, 80), (1024, 34, 40), (512, 100, 40), (512, 510, 40), (256, 50, 40), (256, 18, 40),
            # odd hops: a frame starts in the middle of a pair
            (4096, 441, 96), (4096, 4095, 96), (4096, 2049, 96), (2048, 275, 80), (2048, 1025, 80), (1024, 275, 80), (1024, 33, 40), (512, 161, 40),
            (256, 51, 40), (256, 255, 40), (256, 9, 40)]


@pytest.mark.parametrize("N,hop,K", ANY_HOPS)
def test_any_hop_fused_kernel_matches_generic_and_oracle(oracle, N, hop, K):
    """k_gl_fused_any: the sliding-window kernel for hops that are no multiple of N/16 (the window and the overlap-add
    accumulator move through LDS by hop samples per frame; odd hops shift by floats instead of pairs).  Against the generic kernels on a ragged batch (same arithmetic per
    frame, other summation order of the overlap-add: rounding level) and against the fp64 oracle."""
    import gomel_amd
    fields = dict(NumMels=K, MelFmax=8000.0, Window=hop, Resolut=N, GriffinLimIterations=3)
    gmin = (N - 1) // hop
    frames = [gmin * 4 + 3, 1, 2, gmin + 1, 150, 64, 65]
    mels = [oracle.random_logmel(f, K, seed=300 + i) for i, f in enumerate(frames)]
    seeds = list(range(len(frames)))
    outs = {}
    for path in (2, 1):
        m = mel_obj(fields)
        e = m.engine()
        e.set_path(path)
        b = gomel_amd.Batch(e, [len(mm) for mm in mels])
        b.upload(mels, None, seeds)
        tm = b.run_timed()
        assert tm["fast_path"] == (3 if path == 2 else 0), tm
        outs[path] = b.download()
        b.close(); e.close()
    for a, r in zip(outs[2], outs[1]):
        assert a.shape == r.shape and rms(a - r) < 2e-6 * max(1.0, rms(r)), rms(a - r)
    cfg = oracle.MelConfig(**fields)
    for i in (0, 1, 4):
        want = oracle.from_mel(cfg, mels[i].copy(), oracle.x0_splitmix(seeds[i], len(outs[2][i])))
        assert rms(outs[2][i] - want) < RMS_TOL
