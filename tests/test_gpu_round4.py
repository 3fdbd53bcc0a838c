"""GPU tests added in round 4: what a caller gets WITHOUT a knob (the calibrated fp64-then-fp32 schedule, gomel_auto_fp64_iterations),
calls in flight (gomel_from_mel_batch_submit / gomel_job_wait, whole-batch call mode).  (That a schedule's fp32 magnitudes -- narrowed
from the fp64 rows since this round -- are the bits k_mel_to_mag<float> produces is held by test_gpu_round3.py::test_precision_schedule_is_fp64_then_fp32.)
No fp32_start_mode fixture here: handles start as the library creates them."""
import ctypes
import math
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

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


def inputs(U, F, K, first=0):
    mels = [np.random.default_rng(first + u).uniform(math.log(1e-5), math.log(50.0), size=(F * K, 2)) for u in range(U)]
    return mels, [1000 + first + u for u in range(U)]


def run_batch(eng, mels, seeds):
    import gomel_amd
    b = gomel_amd.Batch(eng, [len(m) for m in mels])
    b.upload(mels, None, seeds)
    b.run()
    out = b.download()
    tm = b.run_timed()
    b.close()
    return out, tm


@pytest.mark.parametrize("fields,iters", [(T, 60), (S, 60), (T, 10), (S, 2)])
def test_the_default_call_is_the_calibrated_schedule(oracle, fields, iters, monkeypatch):
    """A handle nobody touched runs gomel_auto_fp64_iterations(cfg, I) iterations in fp64 and the rest in fp32: the same BITS as the
    explicit schedule of that k, launches split k : I - k between k_gl_fused_d and k_gl_fused; GOMEL_PRECISION picks another start
    mode; gomel_set_precision(h, 0) / GOMEL_SCHEDULE_AUTO come back to it."""
    import gomel_amd
    from gomel_amd import _lib
    monkeypatch.delenv("GOMEL_PRECISION", raising=False)
    m = mel_obj(fields, GriffinLimIterations=iters)
    eng = m.engine()
    k = _lib.lib().gomel_auto_fp64_iterations(ctypes.byref(eng.cfg), iters)
    assert 0 < k <= iters and eng.fp64_iterations() == k and eng.fp64_iterations(iters) == k
    K = fields["NumMels"]
    mels, seeds = inputs(24, 200, K)
    eng.set_path(2)
    odef, tm = run_batch(eng, mels, seeds)
    assert tm["iter_launches"] == iters and tm["iter64_launches"] == min(k, iters)
    eng.set_precision_schedule(k)
    osch, _ = run_batch(eng, mels, seeds)
    assert all(np.array_equal(a, b) for a, b in zip(odef, osch))
    eng.set_precision(32)
    assert eng.fp64_iterations() == 0
    o32, tm32 = run_batch(eng, mels, seeds)
    assert tm32["iter64_launches"] == 0
    eng.set_precision(64)
    assert eng.fp64_iterations() == iters
    o64, _ = run_batch(eng, mels, seeds)
    for mode in (0, "auto"):
        if mode == 0:
            eng.set_precision(0)
        else:
            eng.set_precision(32); eng.set_precision_schedule("auto")
        assert eng.fp64_iterations() == k
        again, _ = run_batch(eng, mels, seeds)
        assert all(np.array_equal(a, b) for a, b in zip(odef, again))
    # the default sits between the two pure modes: (much) nearer to fp64 throughout than fp32 throughout is
    edef = max(rms(a - b) for a, b in zip(odef, o64))
    e32 = max(rms(a - b) for a, b in zip(o32, o64))
    print("%s I=%d k=%d: default vs fp64 %.2e, fp32 vs fp64 %.2e" % (fields["Resolut"], iters, k, edef, e32))
    assert edef < 3e-5 and edef <= e32
    eng.close()
    for env, want in (("32", 0), ("64", iters), ("auto", k)):
        monkeypatch.setenv("GOMEL_PRECISION", env)
        e2 = gomel_amd.Engine(m._config())
        assert e2.fp64_iterations() == want
        e2.close()


def test_small_batches_run_the_default_schedule_too(oracle, coracle, monkeypatch):
    """One utterance (the persistent kernel's size) and a handful (the generic kernels') run the default schedule as the fused kernels
    do (round 4): gomel_auto_fp64_iterations(cfg, I) iterations in the fp64 launch(es), the rest in the fp32 one(s).  The persistent
    path gives the generic pair's bits; both are far inside the tolerance; a forced timeout of the persistent launch recomputes to
    the same bits."""
    import gomel_amd
    monkeypatch.delenv("GOMEL_PRECISION", raising=False)
    fields = dict(T, GriffinLimIterations=8)
    cfg = oracle.MelConfig(**fields)
    for frames in ([342], [342, 300], [700, 650, 40]):   # the middle one: two fp64 persistent launches (each a chip-full at most), one fp32 launch
        mels = [oracle.random_logmel(f, 192, seed=f) for f in frames]
        seeds = list(range(len(frames)))
        m = mel_obj(fields)
        e = m.engine()
        k = e.fp64_iterations()
        assert 0 < k < 8
        got, tm = run_batch(e, mels, seeds)
        assert tm["fast_path"] == (2 if max(frames) <= 512 else 0) and (tm["fast_path"] == 2 or tm["iter64_launches"] == k)
        e.set_path(1)
        gen, tmg = run_batch(e, mels, seeds)
        assert tmg["fast_path"] == 0 and tmg["iter64_launches"] == k
        assert all(np.array_equal(a, b) for a, b in zip(got, gen))
        for g, mel, s in zip(got, mels, seeds):
            want, _ = coracle.from_mel(cfg, mel, oracle.x0_splitmix(s, len(g)))
            assert rms(g - want) < 1e-5
        e.close()
        if frames == [342]:
            monkeypatch.setenv("GOMEL_PERSIST_TICKS", "0")      # every wait of the persistent launches times out at once
            e2 = mel_obj(fields).engine()
            monkeypatch.delenv("GOMEL_PERSIST_TICKS")
            redo = e2.from_mel_batch(mels, None, seeds)
            assert all(np.array_equal(a, b) for a, b in zip(redo, gen))
            e2.close()


@pytest.mark.parametrize("mode", ["default", 32, 64])
def test_submitted_jobs_equal_the_whole_batch_bit_for_bit(oracle, mode, monkeypatch):
    """gomel_from_mel_batch_submit / gomel_job_wait: two calls in flight on one handle (different inputs, different output buffers),
    waited for out of order; each result equals the resident batch API's over the same utterances -- the same whole-batch launches --
    bit for bit, and so does the synchronous call on a handle in whole-batch call mode (gomel_set_call_mode).  A call too small for
    the fused kernels comes back as a finished job with the chunked call's result."""
    import gomel_amd
    monkeypatch.delenv("GOMEL_PRECISION", raising=False)
    fields = dict(T, GriffinLimIterations=12)
    K, U, F = 192, 40, 200
    m = mel_obj(fields)
    eng = m.engine()
    if mode != "default":
        eng.set_precision(mode)
    sets = [inputs(U, F, K, first=100 * i) for i in range(3)]
    refs = [run_batch(eng, mels, seeds)[0] for mels, seeds in sets]
    L = eng.out_len(F * K)
    bound = []
    for mels, _ in sets:
        hm = [gomel_amd.host_alloc((F * K, 2)) for _ in range(U)]
        for a, src in zip(hm, mels):
            a[:] = src
        ho = [gomel_amd.host_alloc(L) for _ in range(U)]
        for o in ho:
            o[:] = -1.0
        bound.append(eng.bind_buffers(hm, ho))
    jobs = [eng.submit_bound(bound[0], sets[0][1]), eng.submit_bound(bound[1], sets[1][1])]
    got1 = [np.array(v) for v in jobs[1].wait()]                      # out of order
    jobs.append(eng.submit_bound(bound[2], sets[2][1]))               # a third while the first is still unclaimed
    got0 = [np.array(v) for v in jobs[0].wait()]
    got2 = [np.array(v) for v in jobs[2].wait()]
    assert jobs[0].done() and jobs[1].done()
    for got, ref in zip((got0, got1, got2), refs):
        assert all(np.array_equal(a, b) for a, b in zip(got, ref))
    # pageable caller buffers (pinned by the library for the life of the job)
    pm = [np.array(a) for a in sets[0][0]]
    po = [np.zeros(L) for _ in range(U)]
    j = eng.submit_bound(eng.bind_buffers(pm, po), sets[0][1])
    assert all(np.array_equal(a, b) for a, b in zip(j.wait(), refs[0]))
    # the synchronous call in whole-batch call mode: the same launches, the same bits; chunked mode: other segments, rounding level
    eng.set_call_mode(1)
    sync = eng.from_mel_batch(sets[1][0], None, sets[1][1])
    assert all(np.array_equal(a, b) for a, b in zip(sync, refs[1]))
    eng.set_call_mode(0)
    # too small for the fused kernels: finished inside submit, the chunked call's result
    small_m, small_s = inputs(2, 30, K, first=7)
    sm = [np.array(a) for a in small_m]
    so = [np.zeros(eng.out_len(30 * K)) for _ in range(2)]
    js = eng.submit_bound(eng.bind_buffers(sm, so), small_s)
    assert js.done()
    want = eng.from_mel_batch(small_m, None, small_s)
    assert all(np.array_equal(a, b) for a, b in zip(js.wait(), want))
    # a handle with a job outstanding refuses to be destroyed (GOMEL_ERR_BUSY = 13) and stays usable
    jd = eng.submit_bound(bound[0], sets[0][1])
    assert gomel_amd._lib.lib().gomel_destroy(eng._h) == 13 and b"busy" in gomel_amd._lib.lib().gomel_strerror(13)
    assert all(np.array_equal(a, b) for a, b in zip(jd.wait(), refs[0]))
    # Engine.close() on a busy handle raises and KEEPS the handle (round 6: dropping it leaked the handle and its device buffers)
    jc = eng.submit_bound(bound[1], sets[1][1])
    with pytest.raises(gomel_amd.GomelError) as busy:
        eng.close()
    assert busy.value.status == 13 and eng._h
    # the bounded wait (gomel_job_wait_timeout): a deadline of one poll on a job that has just been submitted comes back as BUSY with the
    # job still valid; a generous one returns the result; a finished job is freed by it exactly as by gomel_job_wait
    with pytest.raises(TimeoutError):
        jc.wait(timeout_ms=0)
    assert not jc.done() or True                                       # (it may have finished since; either way it is still ours)
    assert all(np.array_equal(a, b) for a, b in zip(jc.wait(timeout_ms=60000), refs[1]))
    assert jc.done() and jc.wait() is not None                        # freed: further waits are no-ops on the Python side
    # errors surface at submit, nothing is left in flight
    with pytest.raises(gomel_amd.GomelError):
        eng.submit_bound(eng.bind_buffers([np.zeros((191, 2))], [np.zeros(8192)]), [0])
    eng.close()


@pytest.mark.stress
def test_two_threads_calling_in_whole_batch_mode_get_the_sequential_bits(oracle, monkeypatch):
    """SURVEY 8(b) "handle usable from concurrent goroutines": two host threads call gomel_from_mel_batch in a loop on ONE handle in
    whole-batch call mode -- their calls pipeline (copies of one beside the kernels of the other) -- and every result equals the
    same call made alone, bit for bit."""
    monkeypatch.delenv("GOMEL_PRECISION", raising=False)
    fields = dict(S, GriffinLimIterations=20)
    K, U, F = 80, 96, 300
    m = mel_obj(fields)
    eng = m.engine()
    eng.set_call_mode(1)
    work = [inputs(U, F, K, first=1000 * i) for i in range(2)]
    alone = [eng.from_mel_batch(mels, None, seeds) for mels, seeds in work]
    errors = []

    def loop(i):
        try:
            for _ in range(6):
                got = eng.from_mel_batch(work[i][0], None, work[i][1])
                if not all(np.array_equal(a, b) for a, b in zip(got, alone[i])):
                    errors.append((i, "mismatch"))
        except Exception as ex:                                       # noqa: BLE001 -- reported below
            errors.append((i, repr(ex)))
    ts = [threading.Thread(target=loop, args=(i,)) for i in range(2)]
    for t in ts: t.start()
    for t in ts: t.join()
    assert not errors, errors[:3]
    eng.close()


def test_tomel_command_line(oracle, tmp_path, capsys):
    """`python -m gomel_amd.tomel <name | name.wav | name.flac>` = cmd/tomel/main.go:10-61: fixed towav-geometry parameters, the PNG
    beside the input, `.wav` / base-name / `.flac` dispatch, the usage line and the error line with exit code 1; and the PNG it writes
    goes back through `python -m gomel_amd.towav` (the reference's two CLIs are each other's inverse up to Griffin-Lim)."""
    import os
    import sys
    import gomel_amd
    from gomel_amd import codec, tomel, towav
    sig = oracle.config2_signal(seconds=1.5)
    wav = str(tmp_path / "a.wav")
    gomel_amd.SaveWav(wav, sig, 44100)
    assert tomel.main(["tomel", wav]) == 0 and os.path.exists(wav + ".png")
    assert tomel.main(["tomel", str(tmp_path / "a")]) == 0 and os.path.exists(str(tmp_path / "a.png"))       # base name: <name>.wav -> <name>.png
    assert open(wav + ".png", "rb").read() == open(str(tmp_path / "a.png"), "rb").read()
    buf, samples, sr = codec.loadpng(wav + ".png", True)
    m = gomel_amd.NewMel()
    m.MelFmax, m.Window, m.NumMels, m.Resolut = 16000, 1280, 192, 4096
    want = m.ToMel(codec.loadwav(wav)[0])
    assert len(buf) == len(want) and float(sr) == float(np.float16(44100.0)) and abs(int(samples) - len(sig)) <= len(sig) // 100            # float16 metadata: < 1 % off
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import flac_writer as fw
    pcm = np.round(sig * 20000).astype(np.int64).tolist()
    frames = [fw.frame([pcm[i:i + 4096]], 16, k, "independent", [dict(kind="fixed", order=2, porder=3 if len(pcm[i:i + 4096]) % 8 == 0 else 0)])
              for k, i in enumerate(range(0, len(pcm), 4096))]
    fl = str(tmp_path / "b.flac")
    open(fl, "wb").write(fw.stream(frames, 44100, 1, 16, len(pcm)))
    assert tomel.main(["tomel", fl]) == 0 and os.path.exists(fl + ".png")
    assert len(codec.loadpng(fl + ".png", True)[0]) == len(want)
    assert tomel.main(["tomel"]) == 1 and "Usage" in capsys.readouterr().out
    assert tomel.main(["tomel", str(tmp_path / "missing")]) == 1
    assert "Error generating mel spectrogram: wavNotLoaded" in capsys.readouterr().out
    assert towav.main(["towav", wav + ".png"]) == 0 and os.path.exists(wav + ".png.wav")


GEOMS = {
    "T": dict(NumMels=192, MelFmax=16000.0, Window=1280, Resolut=4096),   # cmd/towav/main.go:30-36
    "S": dict(NumMels=80, MelFmax=8000.0, Window=256, Resolut=1024),      # BASELINE config 4
    "D": dict(NumMels=160, MelFmax=8000.0, Window=256, Resolut=2048),     # NewMel defaults mel/mel.go:30-41
    "Q": dict(NumMels=64, MelFmax=8000.0, Window=160, Resolut=512),       # 16 kHz speech front end
    "E": dict(NumMels=40, MelFmax=4000.0, Window=80, Resolut=256),        # 8 kHz telephony
    "A": dict(NumMels=80, MelFmax=8000.0, Window=300, Resolut=1024),      # a hop off the N/16 grid: the any-hop fused kernels
    "O": dict(NumMels=80, MelFmax=8000.0, Window=275, Resolut=2048),      # ... an odd one (12.5 ms at 22.05 kHz)
}


@pytest.mark.parametrize("iters", [1, 2, 5, 13, 60])
@pytest.mark.parametrize("geom,frames", [("T", [4]), ("T", [1]), ("T", [40, 9, 1]), ("S", [70, 3, 8]), ("D", [33, 6]), ("Q", [90, 1, 5]), ("E", [130, 7]), ("A", [60, 33, 47]), ("O", [50, 70])])
def test_default_mode_on_every_path_and_shape(oracle, geom, frames, iters, monkeypatch):
    """What a caller gets without a knob, through EVERY kernel path (the library's choice, the generic pair, the fused kernels, the
    persistent kernel where the geometry admits it), on ragged batches of every geometry incl. hops off the N/16 grid and an odd hop:
    the oracle within 1e-5 (the schedule's fp32 tail sits at 1e-7; one iteration, or k = I, is fp64 throughout: 1e-10), the persistent
    kernel the generic pair's bits, explicit x0 and seeded starts, the in-place exp of FromMel."""
    import gomel_amd
    monkeypatch.delenv("GOMEL_PRECISION", raising=False)
    if iters == 60 and sum(frames) > 60:
        iters = 24                                                     # keep the oracle side to seconds
    fields = dict(GEOMS[geom], GriffinLimIterations=iters)
    cfg = oracle.MelConfig(**fields)
    K = fields["NumMels"]
    mels = [oracle.random_logmel(f, K, seed=300 + 7 * i + f) for i, f in enumerate(frames)]
    x0s = [oracle.x0_splitmix(11 + i, oracle.out_len(cfg, len(mm))) if i % 2 == 0 else None for i, mm in enumerate(mels)]
    seeds = [900 + i for i in range(len(frames))]
    wants = [oracle.from_mel(cfg, mm.copy(), x0 if x0 is not None else oracle.x0_splitmix(sd, oracle.out_len(cfg, len(mm)))) for mm, x0, sd in zip(mels, x0s, seeds)]
    m = mel_obj(fields)
    eng = m.engine()
    k = eng.fp64_iterations()
    tol = 1e-10 if k >= iters else 1e-5
    outs = {}
    for path in (0, 1, 2, 3):
        try:
            eng.set_path(path)
            outs[path] = eng.from_mel_batch(mels, x0s, seeds)
        except gomel_amd.GomelError as ex:
            assert ex.status == 1 and path in (2, 3), (path, ex)      # a path this geometry / batch does not admit: bad configuration, nothing else
            continue
        for o, w in zip(outs[path], wants):
            assert o.shape == w.shape and np.all(np.isfinite(o))
            assert rms(o - w) < tol * max(1.0, rms(w)), (path, rms(o - w), k, iters)
    assert 0 in outs and 1 in outs
    if 3 in outs:
        assert all(np.array_equal(a, b) for a, b in zip(outs[3], outs[1]))
    eng.set_path(0)
    one = mels[0].copy()
    got = m.FromMel(one, x0=x0s[0])                                    # the single-call entry point, side effect included (mel/mel.go:143)
    assert rms(got - wants[0]) < tol * max(1.0, rms(wants[0])) and np.allclose(one, np.exp(mels[0]), rtol=1e-15)


def test_big_frame_lengths_in_chunks_of_frames(oracle):
    """gl_big.hpp keeps its two scratch buffers at <= 1 GiB: a batch of more frames than fit goes through in chunks of frames (2048 at
    Resolut 9000, whose convolution has 32768 points).  2100 frames in two utterances, one iteration, against the oracle; ToMel of a
    signal that long likewise."""
    fields = dict(NumMels=64, MelFmax=16000.0, Window=9000, Resolut=9000, GriffinLimIterations=1)
    cfg = oracle.MelConfig(**fields)
    m = mel_obj(fields)
    mels = [oracle.random_logmel(f, 64, seed=f) for f in (2070, 30)]
    x0s = [oracle.x0_splitmix(5 + i, oracle.out_len(cfg, len(mm))) for i, mm in enumerate(mels)]
    got = m.engine().from_mel_batch(mels, x0s, None)
    for g, mm, x0 in zip(got, mels, x0s):
        want = oracle.from_mel(cfg, mm.copy(), x0)
        assert g.shape == want.shape and rms(g - want) < 1e-9 * max(1.0, rms(want))
    sig = np.random.default_rng(3).standard_normal(9000 * 2100) * 0.1
    gotm = m.ToMel(sig)
    wantm = oracle.to_mel(cfg, sig)
    assert gotm.shape == wantm.shape and np.max(np.abs(gotm - wantm)) < 1e-8


def test_float_host_buffers_as_jobs(oracle, monkeypatch):
    """gomel_from_mel_batch_submit_f32: two float32 calls in flight equal the synchronous float32 call in whole-batch call mode, bit for bit."""
    monkeypatch.delenv("GOMEL_PRECISION", raising=False)
    fields = dict(S, GriffinLimIterations=9)
    K, U, F = 80, 48, 200
    eng = mel_obj(fields).engine()
    sets = [[a.astype(np.float32) for a in inputs(U, F, K, first=50 * i)[0]] for i in range(2)]
    seeds = [list(range(10 * i, 10 * i + U)) for i in range(2)]
    jobs = [eng.submit_f32(sets[i], None, seeds[i]) for i in range(2)]
    got = [[np.array(v) for v in j.wait()] for j in jobs]
    eng.set_call_mode(1)
    for i in range(2):
        want = eng.from_mel_batch_f32(sets[i], None, seeds[i])
        assert all(g.dtype == np.float32 and np.array_equal(g, w) for g, w in zip(got[i], want))
    eng.close()


def test_jobs_of_a_size_the_persistent_kernels_would_take(oracle, monkeypatch):
    """Two launches of the fp32 persistent kernel reach up to 8192 frames at N 1024 -- above the size from which a call is submitted as whole-batch
    launches (2 x 3072 frames): a submitted job never takes the persistent path (it wants a verifying wait), a synchronous call of that size does."""
    monkeypatch.delenv("GOMEL_PRECISION", raising=False)
    fields = dict(S, GriffinLimIterations=6)
    K, U, F = 80, 35, 200
    eng = mel_obj(fields).engine()
    eng.set_precision(32)
    mels, seeds = inputs(U, F, K, first=7)
    _, tm = run_batch(eng, mels, seeds)
    assert tm["fast_path"] == 2 and tm["frames"] == U * F          # the resident batch: persistent, two launches
    job = eng.submit_f32([a.astype(np.float32) for a in mels], None, seeds)
    got = [np.array(v) for v in job.wait()]
    eng.set_call_mode(1)
    want = eng.from_mel_batch_f32([a.astype(np.float32) for a in mels], None, seeds)
    assert all(np.array_equal(g, w) for g, w in zip(got, want))
    ref = eng.from_mel_batch(mels, None, seeds)
    assert max(rms(g - r) for g, r in zip(got, ref)) < 1e-4
    eng.close()


@pytest.mark.parametrize("N,hop,F,I", [(1024, 256, 9, 3), (4096, 1280, 40, 6), (2048, 300, 25, 4)])
def test_exported_istft_in_the_default_mode(oracle, N, hop, F, I, monkeypatch):
    """ISTFT(s, spectrogram, numIterations) (mel/mel.go:76-139) on a handle nobody touched: from a list (no write-back: the default
    schedule on whatever path the batch takes) and from a complex128 array (the write-back runs fp64 throughout), both against the
    oracle, the latter with the spectrogram side effect."""
    import gomel_amd
    monkeypatch.delenv("GOMEL_PRECISION", raising=False)

    class STFT:
        pass
    s = STFT()
    s.FrameShift, s.Window = hop, oracle.hann(N)
    rng = np.random.default_rng(N + hop)
    spec = rng.uniform(0.1, 2.0, size=(F, N)) * np.exp(2j * np.pi * rng.uniform(size=(F, N)))
    x0 = oracle.x0_splitmix(3, N + (F - 1) * hop)
    ref_spec = spec.copy()
    want = oracle.istft_griffin_lim(s.Window, hop, ref_spec, I, x0)
    got = gomel_amd.ISTFT(s, [list(r) for r in spec], I, x0=x0)
    assert rms(got - want) < 1e-5 * max(1.0, rms(want))
    arr = spec.copy()
    got2 = gomel_amd.ISTFT(s, arr, I, x0=x0)
    assert rms(got2 - want) < 1e-9 * max(1.0, rms(want))
    assert np.max(np.abs(arr - ref_spec)) < 1e-9 * np.max(np.abs(ref_spec))


def test_fused_threshold_follows_the_precision_mode(monkeypatch):
    """The batch size from which the fused kernel beats the generic pair depends on whether the call has fp64 iterations (N 4096: 2048
    frames against 1536, scripts/size_sweep.py): 5 x 342 frames run the generic kernels in the default and the 64-bit mode, the
    fused kernel in the fast mode; 6 x 342 run the fused kernels in every mode.  (N 2048 / hop 256: 4200 against 5973, N 1024: 4500 against 3072.)"""
    import gomel_amd
    monkeypatch.delenv("GOMEL_PRECISION", raising=False)
    for sched, small, large in (("auto", 0, 1), (-1, 0, 1), (0, 1, 1)):
        m = mel_obj(T, GriffinLimIterations=4)
        eng = m.engine()
        eng.set_precision_schedule(sched)
        for U, want in ((5, small), (6, large)):
            mels, seeds = inputs(U, 342, 192)
            _, tm = run_batch(eng, mels, seeds)
            assert tm["fast_path"] == want and tm["frames"] == U * 342, (sched, U, tm)
        eng.close()
    for fields, frames, want_default, want_fast in ((S, 4000, 2, 2), (S, 4800, 1, 2),   # (2: the persistent kernels -- 4000 frames of N 1024: fp32 one launch, default two fp64 launches + one fp32; 4800: fp32 two launches, default three would not pay: fused)
                                                    (dict(NumMels=160, MelFmax=8000.0, Window=256, Resolut=2048), 4800, 1, 0)):
        for sched, want in (("auto", want_default), (0, want_fast)):
            m = mel_obj(fields, GriffinLimIterations=4)
            eng = m.engine()
            eng.set_precision_schedule(sched)
            mels, seeds = inputs(frames // 400, 400, fields["NumMels"])
            _, tm = run_batch(eng, mels, seeds)
            assert tm["fast_path"] == want, (fields["Resolut"], frames, sched, tm)
            eng.close()
