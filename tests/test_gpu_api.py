"""GPU tests of the rest of the mel.Mel surface through the C ABI: ToMel, ToWavPng, the golden fixtures, table-blob
handles, and size-independent properties at BASELINE's full config-3 size."""
import math
import os

import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("fp32_start_mode")]   # these tests target the fp32 kernels unless they select a mode (conftest.py)
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "golden_v1.npz"))
RMS_TOL = 1e-4

FIELDS = {
    "tts_n1024": dict(NumMels=80, MelFmax=8000.0, Window=256, Resolut=1024),
    "default_n2048": dict(NumMels=160, MelFmax=8000.0, Window=256, Resolut=2048),
    "towav_n4096": dict(NumMels=192, MelFmax=16000.0, Window=1280, Resolut=4096),
}
GOLD_CASES = {"tts_n1024": (8, 3, 4, (1, 2, 60)), "default_n2048": (6, 5, 6, (2,)), "towav_n4096": (4, 7, 8, (2, 60))}


def rms(a):
    return float(np.sqrt(np.mean(np.square(a))))


def mel_obj(fields, **kw):
    import gomel_amd
    m = gomel_amd.NewMel()
    for k, v in dict(fields, **kw).items():
        setattr(m, k, v)
    return m


@pytest.mark.parametrize("name", sorted(GOLD_CASES))
def test_from_mel_against_committed_golden(oracle, name):
    F, ms, xs, iters = GOLD_CASES[name]
    for I in iters:
        m = mel_obj(FIELDS[name], GriffinLimIterations=I)
        mel = oracle.random_logmel(F, m.NumMels, seed=ms)
        x0 = oracle.x0_splitmix(xs, F and (m.Resolut + (F - 1) * m.Window))
        got = m.FromMel(mel, x0=x0)
        assert rms(got - GOLD["%s_i%d_wave" % (name, I)]) < RMS_TOL


@pytest.mark.parametrize("name", ["tts_n1024", "towav_n4096", "default_n2048"])
def test_to_mel_matches_oracle(oracle, name):
    m = mel_obj(FIELDS[name])
    cfg = oracle.MelConfig(**FIELDS[name])
    sig = oracle.config2_signal(seconds=0.5)
    got = m.ToMel(sig)
    want = oracle.to_mel(cfg, sig)
    assert got.shape == want.shape
    # log domain, fp64 on both sides: bins at the 1e-5 floor included
    assert np.max(np.abs(got - want)) < 1e-9
    if name + "_tomel" in GOLD.files:
        assert np.max(np.abs(got - GOLD[name + "_tomel"])) < 1e-9
    # edge: shorter than 15 hops is padded to 15*hop - 1 (mel/impl.go:444-447)
    short = sig[:1000]
    assert np.max(np.abs(m.ToMel(short) - oracle.to_mel(cfg, short))) < 1e-9


def test_to_mel_too_short_is_an_error():
    import gomel_amd
    m = mel_obj(dict(NumMels=80, MelFmax=8000.0, Window=2, Resolut=1024))   # 15 hops = 30 samples < one frame
    with pytest.raises(gomel_amd.GomelError) as ei:
        m.ToMel(np.zeros(10))
    assert ei.value.status == 11


def test_round_trip_to_mel_from_mel_towavpng(oracle, tmp_path):
    """ToWavPng on the generated 192-row PNG (the towav configuration, cmd/towav/main.go:30-40) equals the oracle's
    loadpng -> FromMel -> trim -> 16-bit PCM, sample for sample up to one LSB where the fp32 path rounds across a step."""
    from gomel_amd import codec
    png = os.path.join(HERE, "golden", "chirp_192.png")
    m = mel_obj(FIELDS["towav_n4096"], MelFmin=0.0, YReverse=True, GriffinLimIterations=2, VolumeBoost=0.25, SampleRate=0)
    img = codec.read_png(png)
    buf, samples, sr = oracle.loadpng_pixels(img, True)
    cfg = oracle.towav_config(VolumeBoost=0.25)
    buf = buf + 0.25
    x0 = oracle.x0_splitmix(42, oracle.out_len(cfg, len(buf)))
    want = oracle.from_mel(cfg, buf, x0)
    # mel.go:228-230: trim only if isPadded(samples, len) and len > samples.  The STFT drops the partial last frame, so
    # L = N + (F-1) hop is shorter than the padded length and the trim does not fire here (it practically never does).
    if int(samples) > 0 and oracle.is_padded(int(samples), len(want), 1280) and len(want) > int(samples):
        want = want[:int(samples)]
    assert len(want) == 4096 + 38 * 1280
    out = str(tmp_path / "o.wav")
    assert m.ToWavPng(png, out, x0=x0) is None
    assert m.SampleRate == 44096                                            # float16(44100), adopted since SampleRate was 0
    raw = open(out, "rb").read()
    pcm = np.frombuffer(raw[44:], dtype="<i2")
    wpcm = oracle.wav_pcm16(want)
    assert len(pcm) == len(wpcm)
    assert np.max(np.abs(pcm.astype(int) - wpcm.astype(int))) <= 1
    import gomel_amd
    with pytest.raises(gomel_amd.ErrFileNotLoaded):
        m.ToWavPng(str(tmp_path / "missing.png"), out)


def test_tables_blob_handle_is_bit_identical(oracle):
    import gomel_amd
    m = mel_obj(FIELDS["tts_n1024"], GriffinLimIterations=4)
    e0 = m.engine()
    blob = e0.export_tables()
    assert blob == gomel_amd.build_tables(m._config())      # the host-only builder makes the same bytes (tests/test_tables.py checks them)
    e1 = gomel_amd.Engine(m._config(), tables=blob)
    mel = oracle.random_logmel(9, 80, seed=2)
    a = e0.from_mel(mel, seed=5)
    b = e1.from_mel(mel, seed=5)
    assert np.array_equal(a, b)
    bad = bytearray(blob)
    bad[0] ^= 0xFF
    with pytest.raises(gomel_amd.GomelError) as ei:
        gomel_amd.Engine(m._config(), tables=bytes(bad))
    assert ei.value.status == 10
    m2 = mel_obj(FIELDS["tts_n1024"], NumMels=64)
    with pytest.raises(gomel_amd.GomelError):
        gomel_amd.Engine(m2._config(), tables=blob)


def test_sharded_helper_on_gpu_world1(oracle):
    from gomel_amd import dist as gd
    m = mel_obj(FIELDS["tts_n1024"], GriffinLimIterations=2)
    e = gd.create_engine_broadcast(lambda tables=None: m.engine(), 0, 1)
    mels = [oracle.random_logmel(f, 80, seed=f) for f in (4, 9, 6)]
    outs = gd.from_mel_sharded(lambda ms, xs, ss: e.from_mel_batch(ms, xs, ss), mels, None, [7, 8, 9], 0, 1)
    cfg = oracle.MelConfig(GriffinLimIterations=2, **FIELDS["tts_n1024"])
    for mm, s, o in zip(mels, (7, 8, 9), outs):
        assert rms(o - oracle.from_mel(cfg, mm.copy(), oracle.x0_splitmix(s, len(o)))) < RMS_TOL


def test_full_size_config3_properties(oracle):
    """BASELINE configs[2]: 256 x (192 mels x 400 frames), N 4096 / hop 1280, 60 iterations -- through size-independent
    properties: run-to-run bitwise determinism; exact homogeneity (TuneMul = 1/2 doubles every target magnitude exactly,
    so the waveform doubles bit for bit: the iteration only ever uses the phase of x); independence of batch neighbours;
    and two utterances checked against the fp64 oracle at full length."""
    import gomel_amd
    fields = dict(FIELDS["towav_n4096"], GriffinLimIterations=60)
    U, F, K = 256, 400, 192
    mels = [np.random.default_rng(u).uniform(math.log(1e-5), math.log(50.0), size=(F * K, 2)) for u in range(U)]
    seeds = [1000 + u for u in range(U)]
    m = mel_obj(fields)
    b = gomel_amd.Batch(m.engine(), [F * K] * U)
    b.upload(mels, None, seeds)
    b.run()
    out1 = b.download()
    b.run()
    out2 = b.download()
    b.close()
    L = 4096 + (F - 1) * 1280
    assert all(o.shape == (L,) and np.isfinite(o).all() for o in out1)
    assert all(np.array_equal(a, c) for a, c in zip(out1, out2))                        # deterministic, no atomics
    # homogeneity on a sub-batch that keeps utterances 3 and 200 but different neighbours
    m2 = mel_obj(fields, TuneMul=0.5)
    sub = [3, 200, 17]
    o_half = m2.engine().from_mel_batch([mels[u] for u in sub], None, [seeds[u] for u in sub])
    o_same = m.engine().from_mel_batch([mels[u] for u in sub], None, [seeds[u] for u in sub])
    for u, a, s in zip(sub, o_half, o_same):
        assert np.array_equal(a, 2.0 * s)                                               # exact: scaling by 2 commutes with rounding
        # other batch shape -> other segmentation -> other fp32 summation order.  Griffin-Lim keeps only the PHASE of each
        # bin, which is ill-conditioned where |X[k]| sinks to the FFT's rounding floor; on these iid-random magnitudes a few
        # such bins per utterance flip differently between two roundings (isolated spikes, typical RMS 1e-6, tail ~5e-5).
        assert rms(s - out1[u]) < RMS_TOL
    cfg = oracle.MelConfig(**fields)
    for u in (3, 200):
        want = oracle.from_mel(cfg, mels[u].copy(), oracle.x0_splitmix(seeds[u], L))
        assert rms(out1[u] - want) < RMS_TOL, (u, rms(out1[u] - want))


@pytest.mark.stress
def test_concurrent_callers_share_and_separate_handles(oracle):
    """The reference is plain Go and callable from many goroutines; the handle serialises calls on itself with a mutex and sets
    the device on every entry.  Eight host threads hammer one shared handle and one private handle each."""
    import threading
    m_shared = mel_obj(FIELDS["tts_n1024"], GriffinLimIterations=3)
    shared = m_shared.engine()
    cfg = oracle.MelConfig(GriffinLimIterations=3, **FIELDS["tts_n1024"])
    mels = [oracle.random_logmel(5 + i, 80, seed=i) for i in range(8)]
    wants = [oracle.from_mel(cfg, mm.copy(), oracle.x0_splitmix(i, oracle.out_len(cfg, len(mm)))) for i, mm in enumerate(mels)]
    errors = []

    def work(i):
        try:
            own = mel_obj(FIELDS["tts_n1024"], GriffinLimIterations=3).engine()
            for _ in range(6):
                for e in (shared, own):
                    got = e.from_mel(mels[i], seed=i)
                    if rms(got - wants[i]) > RMS_TOL:
                        errors.append((i, rms(got - wants[i])))
            own.close()
        except Exception as ex:  # noqa: BLE001
            errors.append((i, repr(ex)))

    ts = [threading.Thread(target=work, args=(i,)) for i in range(8)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors[:3]


def test_precision_field_on_mel(oracle):
    m = mel_obj(FIELDS["tts_n1024"], GriffinLimIterations=10)
    cfg = oracle.MelConfig(GriffinLimIterations=10, **FIELDS["tts_n1024"])
    mel = oracle.random_logmel(12, 80, seed=3)
    x0 = oracle.x0_splitmix(2, oracle.out_len(cfg, len(mel)))
    want = oracle.from_mel(cfg, mel.copy(), x0)
    m.Precision = 64
    assert rms(m.FromMel(mel.copy(), x0=x0) - want) < 1e-10
    m.Precision = 32
    assert 1e-10 < rms(m.FromMel(mel.copy(), x0=x0) - want) < RMS_TOL


def test_aliased_and_repeated_host_buffers(oracle):
    """The same host array passed for several utterances, and big (> 1 MiB, i.e. pinned for the call) buffers that alias each
    other: legal inputs; the library pins caller buffers for the duration of a call and must cope."""
    import gomel_amd
    m = mel_obj(FIELDS["towav_n4096"], GriffinLimIterations=1)
    e = m.engine()
    mel = oracle.random_logmel(400, 192, seed=1)             # 1.2 MB: above the pinning threshold
    for call in range(2):
        outs = e.from_mel_batch([mel] * 6, None, [5] * 6)
        assert all(np.array_equal(outs[0], o) for o in outs[1:])
        b = gomel_amd.Batch(e, [len(mel)] * 4)
        b.upload([mel] * 4, None, [5] * 4)
        b.run()
        big = np.zeros(4 * len(outs[0]))
        views = [big[i * len(outs[0]):(i + 1) * len(outs[0])] for i in range(4)]   # four slices of ONE allocation
        got = b.download(views)
        b.close()
        assert all(np.array_equal(g, outs[0]) for g in got)
    cfg = oracle.MelConfig(GriffinLimIterations=1, **FIELDS["towav_n4096"])
    want = oracle.from_mel(cfg, mel.copy(), oracle.x0_splitmix(5, len(outs[0])))
    assert rms(outs[0] - want) < RMS_TOL


@pytest.mark.stress
def test_threads_sharing_one_input_buffer(oracle):
    """Several threads, each with its own handle, pass the SAME large input array at the same time (pinned per call through a
    reference-counted registry inside the library)."""
    import threading
    mel = oracle.random_logmel(400, 192, seed=2)              # 1.2 MB, above the pinning threshold
    cfg = oracle.MelConfig(GriffinLimIterations=1, **FIELDS["towav_n4096"])
    want = oracle.from_mel(cfg, mel.copy(), oracle.x0_splitmix(9, oracle.out_len(cfg, len(mel))))
    errors = []

    def work(i):
        try:
            e = mel_obj(FIELDS["towav_n4096"], GriffinLimIterations=1).engine()
            for _ in range(5):
                outs = e.from_mel_batch([mel, mel], None, [9, 9])
                if rms(outs[0] - want) > RMS_TOL or not np.array_equal(outs[0], outs[1]):
                    errors.append(i)
            e.close()
        except Exception as ex:  # noqa: BLE001
            errors.append((i, repr(ex)))

    ts = [threading.Thread(target=work, args=(i,)) for i in range(6)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors[:3]


def _device_resident_case(oracle, to_device, empty_device, to_host, wrong_dtype=None):
    """Spectrograms, x0 and the output may live in HBM (device pointers through the C ABI): same bits as the host-buffer route, mixed
    host / device batches included.  `to_device(ndarray)`, `empty_device(n)` and `to_host(obj)` supply the device memory."""
    import gomel_amd
    m = mel_obj(FIELDS["towav_n4096"], GriffinLimIterations=2)
    e = m.engine()
    mels = [oracle.random_logmel(f, 192, seed=20 + i) for i, f in enumerate((40, 400, 3))]   # 400 frames: above the pin threshold
    lens = [e.out_len(len(x)) for x in mels]
    x0s = [oracle.x0_splitmix(7, lens[0]), None, oracle.x0_splitmix(8, lens[2])]
    b = gomel_amd.Batch(e, [len(x) for x in mels])
    b.upload(mels, x0s, [1, 2, 3])
    b.run()
    want = b.download()
    d_mels = [to_device(x) for x in mels]
    d_x0s = [to_device(x0s[0]), None, x0s[2]]                                               # device, absent, host
    b.upload([d_mels[0], mels[1], d_mels[2]], d_x0s, [1, 2, 3])                             # mixed host / device
    b.run()
    d_outs = [empty_device(L) for L in lens]
    got = b.download([d_outs[0], np.empty(lens[1]), d_outs[2]])
    for g, w in zip(got, want):
        g = g if isinstance(g, np.ndarray) else to_host(g)
        assert np.array_equal(g, w)
    if wrong_dtype is not None:
        with pytest.raises(ValueError):
            b.upload([wrong_dtype(d_mels[0]), mels[1], d_mels[2]], None, [1, 2, 3])         # wrong dtype on the device
    b.close()
    e.close()


def test_device_resident_inputs_and_outputs(oracle, hip):
    """Device pointers from the runtime's own allocator (hipMalloc through ctypes: what a Go / C++ host with its own HIP binding hands
    over), wrapped in the minimal object gomel_amd.mel takes for device memory (data_ptr / is_cuda / dtype / numel)."""
    import ctypes
    import gomel_amd
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    held = []

    def empty_device(n):
        p = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(p), 8 * int(n)) == 0
        held.append(p)
        return gomel_amd.DevicePointer(p.value, int(n))

    def to_device(a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        d = empty_device(a.size)
        assert hip.hipMemcpy(d.data_ptr(), a.ctypes.data, a.nbytes, 1) == 0                  # hipMemcpyHostToDevice
        return d

    def to_host(d):
        out = np.empty(d.numel())
        assert hip.hipMemcpy(out.ctypes.data, d.data_ptr(), out.nbytes, 2) == 0              # hipMemcpyDeviceToHost
        return out

    try:
        _device_resident_case(oracle, to_device, empty_device, to_host)
    finally:
        for p in held:
            hip.hipFree(p)


def _torch_device_resident_child(rep):
    """Child process (tests/_children.py): torch is imported, then the library; torch CUDA tensors go through the C ABI as device
    pointers (two allocators on one runtime; torch copies on its legacy null stream, the library on its own non-blocking streams)."""
    import torch
    import gomel_oracle
    rep.stage("torch.cuda init")
    dev = torch.device("cuda", torch.cuda.current_device())
    torch.zeros(1, device=dev)
    rep.stage("library calls")
    _device_resident_case(gomel_oracle, lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev),
                          lambda n: torch.empty(n, dtype=torch.float64, device=dev), lambda t: t.cpu().numpy(), wrong_dtype=lambda t: t.float())
    import gomel_amd
    rep.result({"runtime": gomel_amd.hip_runtime_path()})


@pytest.mark.torch_child
@pytest.mark.guard(700)
def test_device_resident_torch_tensors_in_a_child_process():
    """The same case with torch CUDA tensors, in a child process that imports torch first (the one order in which a Python host shares
    torch's HIP runtime with the library without asking, gomel_amd/_lib.py)."""
    import _children
    got = _children.run_child(_torch_device_resident_child)
    assert got["runtime"] and "torch" in got["runtime"] and " + " not in got["runtime"], got   # ONE runtime in that process: torch's copy


def test_working_buffers_reused_across_calls_leave_no_trace(oracle):
    """A handle keeps the device buffers of finished calls and hands them out again (hipFree is the slowest part of a short
    call).  Results must not depend on what a recycled buffer held: every sequence below is compared, bit for bit, with the
    same call on a fresh handle."""
    import gomel_amd
    fields = FIELDS["towav_n4096"]

    def run(e, frames, seed, path):
        e.set_path(path)
        mels = [oracle.random_logmel(f, 192, seed=seed + i) for i, f in enumerate(frames)]
        return e.from_mel_batch(mels, None, [seed + 10 * i for i in range(len(frames))])

    plan = [((300, 280, 41), 1, 0), ((200, 390), 5, 0), ((37,), 9, 0), ((210, 200, 199), 3, 1), ((16, 900), 7, 0), ((300, 280, 41), 1, 0)]
    shared = mel_obj(fields, GriffinLimIterations=3).engine()
    for frames, seed, path in plan:
        got = run(shared, frames, seed, path)
        fresh = mel_obj(fields, GriffinLimIterations=3).engine()
        want = run(fresh, frames, seed, path)
        fresh.close()
        assert all(np.array_equal(g, w) for g, w in zip(got, want)), (frames, seed, path)
    wave = oracle.config2_signal(44100, 0.4)
    a = shared.to_mel(wave)                                   # ToMel's temporaries come from the same cache
    fresh = mel_obj(fields, GriffinLimIterations=3).engine()
    assert np.array_equal(a, fresh.to_mel(wave))
    fresh.close()
    shared.close()


def test_to_mel_tables_fuzz_against_oracle(oracle):
    """Random band counts / band edges / FFT sizes / hops through ToMel (host-built domel table + fp64 STFT kernels) against the
    oracle; configurations whose domel table runs past the bins (the reference panics) must come back as GOMEL_ERR_TABLE_RANGE."""
    import gomel_amd
    rng = np.random.default_rng(77)
    sig = oracle.config2_signal(seconds=0.25)
    ran = refused = 0
    for trial in range(40):
        N = int(rng.choice([256, 512, 1024, 2048, 4096]))
        hop = int(rng.choice([N // 16, N // 8, N // 4, int(rng.integers(40, N // 2)), 2 * int(rng.integers(20, N // 4)) + 1]))
        K = int(rng.integers(6, 300))
        fields = dict(NumMels=K, MelFmin=float(rng.choice([0.0, rng.uniform(0, 300)])), MelFmax=float(rng.uniform(1500, 26000)), Window=hop, Resolut=N)
        cfg = oracle.MelConfig(**fields)
        x = sig[:int(rng.integers(N, len(sig)))]
        try:
            want = oracle.to_mel(cfg, x)
        except (IndexError, ValueError):
            want = None
        m = mel_obj(fields)
        if want is None:
            with pytest.raises(gomel_amd.GomelError) as ei:
                m.ToMel(x)
            assert ei.value.status == 5, fields
            refused += 1
        else:
            got = m.ToMel(x)
            assert got.shape == want.shape, fields
            bad = ~np.isfinite(want)
            assert np.array_equal(bad, ~np.isfinite(got)), fields
            assert np.max(np.abs(got[~bad] - want[~bad])) < 1e-9, fields
            ran += 1
        m.engine().close()
    assert ran >= 20, (ran, refused)


@pytest.mark.parametrize("path", [0, 1])
def test_more_utterances_than_grid_y(oracle, path):
    """70 000 one- and two-frame utterances in one batch: more than the 65 535 blocks grid.y can hold (the per-utterance kernels
    stride), on the fused and on the generic kernels."""
    import gomel_amd
    fields = dict(NumMels=8, MelFmax=4000.0, Window=64, Resolut=256, GriffinLimIterations=2)
    m = mel_obj(fields)
    e = m.engine()
    e.set_path(path)
    n = 70000
    base = [oracle.random_logmel(f, 8, seed=f) for f in (1, 2)]
    mels = [base[i & 1] for i in range(n)]
    outs = e.from_mel_batch(mels, None, list(range(n)))
    cfg = oracle.MelConfig(**fields)
    for u in (0, 1, 65534, 65535, 65536, n - 1):
        want = oracle.from_mel(cfg, mels[u].copy(), oracle.x0_splitmix(u, len(outs[u])))
        assert rms(outs[u] - want) < RMS_TOL, u
    e.close()
