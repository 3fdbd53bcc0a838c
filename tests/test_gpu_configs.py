"""One oracle comparison per BASELINE.json config, in the DEFAULT arithmetic mode (a handle nobody has turned a knob on: the calibrated
fp64-then-fp32 schedule), at the config's full size.  Collected FIRST (tests/conftest.py): whatever else a session hits later, the
parity of the hot path (mel/mel.go:76-152) on the five named workloads has been decided by then.

How a full-size batch is checked in seconds: the C restatement of the reference (oracle/gomel_oracle.c) runs a SAMPLE of the batch's
utterances; the GPU's fp64-throughout mode must match it to 1e-9 on that sample (same float64 arithmetic, two implementations), and
then serves as the reference for EVERY utterance of the batch: the default call must stay within the north-star's 1e-4 RMS of it on all
of them (the sample is also compared with the oracle directly).  Run-to-run bitwise determinism is asserted on the whole batch.
The opt-in fp32 fast mode gets its statistical check on the same inputs (what that arithmetic can promise: DESIGN.md section 2).
"""
import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu                      # no fp32_start_mode here: handles start as the library creates them
HERE = os.path.dirname(os.path.abspath(__file__))
RMS_TOL = 1e-4
AUTO_60 = 35                                      # gomel_auto_fp64_iterations at 60 iterations, overlap <= 4-fold
T = dict(NumMels=192, MelFmax=16000.0, Window=1280, Resolut=4096)     # BASELINE configs[0..2,4]: the towav geometry (cmd/towav/main.go:30-37)
S = dict(NumMels=80, MelFmax=8000.0, Window=256, Resolut=1024)        # BASELINE configs[3]: the TTS geometry


def rms(a):
    return float(np.sqrt(np.mean(np.square(a))))


def mel_obj(fields, **kw):
    import gomel_amd
    m = gomel_amd.NewMel()
    for k, v in dict(fields, **kw).items():
        setattr(m, k, v)
    return m


def cfg3_inputs(U, F, K, first=0):
    """iid U[ln 1e-5, ln 50] log-mel, seed = GLOBAL utterance index; x0 seed 1000 + index."""
    mels = [np.random.default_rng(first + u).uniform(math.log(1e-5), math.log(50.0), size=(F * K, 2)) for u in range(U)]
    return mels, [1000 + first + u for u in range(U)]


@pytest.fixture(autouse=True)
def _no_knob(monkeypatch):
    monkeypatch.delenv("GOMEL_PRECISION", raising=False)     # the handle as a caller gets it: no knob, no environment


def _batch_twice(gomel_amd, eng, mels, seeds):
    b = gomel_amd.Batch(eng, [len(m) for m in mels])
    b.upload(mels, None, seeds)
    b.run()
    o1 = b.download()
    b.run()
    o2 = b.download()
    b.close()
    assert all(np.array_equal(a, c) for a, c in zip(o1, o2))                # run-to-run bitwise (no atomics)
    return o1


def _full_size_batch(oracle, coracle, fields, K, name, sample):
    """256 utterances x 400 frames, 60 iterations: default call, fp64 throughout, fp32 fast mode (module docstring)."""
    import gomel_amd
    fields = dict(fields, GriffinLimIterations=60)
    U, F = 256, 400
    mels, seeds = cfg3_inputs(U, F, K)
    cfg = oracle.MelConfig(**fields)
    L = oracle.out_len(cfg, F * K)
    want, _ = coracle.from_mel_batch(cfg, [mels[u] for u in sample], [oracle.x0_splitmix(seeds[u], L) for u in sample], threads=min(16, os.cpu_count() or 1))
    eng = mel_obj(fields).engine()
    assert eng.fp64_iterations() == AUTO_60
    odef = _batch_twice(gomel_amd, eng, mels, seeds)                        # the DEFAULT call: nothing is set on the handle
    eng.set_precision(64)
    o64 = _batch_twice(gomel_amd, eng, mels, seeds)
    e64 = np.array([rms(o64[u] - w) for u, w in zip(sample, want)])
    print("%s, 64-bit mode vs C oracle on %d sampled utterances: max %.2e" % (name, len(sample), e64.max()))
    assert e64.max() < 1e-9
    edef_o = np.array([rms(odef[u] - w) for u, w in zip(sample, want)])
    edef = np.array([rms(a - r) for a, r in zip(odef, o64)])
    print("%s, DEFAULT call (%d fp64 + %d fp32 iterations): vs C oracle on the sample max %.2e; vs the fp64 kernels on ALL %d utterances median %.2e p99 %.2e max %.2e" % (
        name, AUTO_60, 60 - AUTO_60, edef_o.max(), U, np.median(edef), np.percentile(edef, 99), edef.max()))
    assert edef_o.max() < RMS_TOL, (edef_o.max(), sample[int(np.argmax(edef_o))])
    assert edef.max() < RMS_TOL - 1e-8, (edef.max(), int(np.argmax(edef)))   # HARD bar on every utterance (1e-9 of slack for the fp64 kernels' own distance)
    assert all(o.shape == (L,) and np.isfinite(o).all() for o in odef)
    eng.set_precision(32)                                                   # the opt-in fast mode: statistical check, named as such
    o32 = _batch_twice(gomel_amd, eng, mels, seeds)
    e32 = np.array([rms(a - r) for a, r in zip(o32, o64)])
    over = np.flatnonzero(e32 > RMS_TOL)
    print("%s, fp32 throughout (opt-in fast mode) vs the fp64 kernels: median %.2e p90 %.2e p99 %.2e max %.2e, over 1e-4: %s" % (
        name, np.median(e32), np.percentile(e32, 90), np.percentile(e32, 99), e32.max(), over.tolist()))
    assert np.median(e32) < 1e-5 and np.percentile(e32, 90) < 5e-5 and len(over) <= max(1, U // 100) + 1, over
    hop = fields["Window"]
    for u in np.flatnonzero(e32 > 3e-5):                                    # an utterance above 3e-5 owes it to isolated hop blocks: no drift
        d = (o32[u] - o64[u])[:(L // hop) * hop].reshape(-1, hop)
        blocks = np.sqrt(np.mean(d * d, axis=1))
        assert np.median(blocks) < 3e-6, (u, np.median(blocks))
    eng.close()


def test_config1_towav_on_a_png(oracle, tmp_path):
    """BASELINE configs[0]: `towav` on a PNG -- 44.1 kHz, NumMels 192, Resolut 4096, 2 iterations (cmd/towav/main.go:27-45).  The PNG
    the config names is the reference's README banner, 183 x 80: with NumMels = 192 its 14 640 pairs are no whole number of frames --
    the reference panics in undomel (mel/impl.go:366), the C ABI returns GOMEL_ERR_BAD_LENGTH.  The pipeline itself runs on the
    repository's 192-row fixture: ToWavPng = loadpng -> +VolumeBoost -> FromMel -> trim -> 16-bit PCM, against the oracle's."""
    import gomel_amd
    from gomel_amd import codec
    m = mel_obj(T, MelFmin=0.0, YReverse=True, GriffinLimIterations=2, VolumeBoost=0.0, SampleRate=0)
    banner, _, _ = codec.loadpng(os.path.join(HERE, "golden", "reference_banner_183x80.png"), True)
    assert len(banner) == 183 * 80
    with pytest.raises(gomel_amd.GomelError) as ei:
        m.FromMel(banner, seed=1)
    assert ei.value.status == 3                                              # GOMEL_ERR_BAD_LENGTH
    png = os.path.join(HERE, "golden", "chirp_192.png")
    buf, samples, sr = oracle.loadpng_pixels(codec.read_png(png), True)
    cfg = oracle.towav_config()
    x0 = oracle.x0_splitmix(42, oracle.out_len(cfg, len(buf)))
    want = oracle.from_mel(cfg, buf.copy(), x0)                              # (the oracle keeps the reference's in-place exp side effect)
    assert m.engine().fp64_iterations() == 2                                 # the shipped 2 iterations run in fp64 throughout
    got = m.FromMel(buf.copy(), x0=x0)
    print("config 1 (PNG fixture, %d frames, 2 iterations), default call vs oracle: %.2e" % (len(buf) // 192, rms(got - want)))
    assert rms(got - want) < 1e-9
    out = str(tmp_path / "o.wav")
    assert m.ToWavPng(png, out, x0=x0) is None
    pcm = np.frombuffer(open(out, "rb").read()[44:], dtype="<i2")
    if int(samples) > 0 and oracle.is_padded(int(samples), len(want), 1280) and len(want) > int(samples):    # mel.go:228-230
        want = want[:int(samples)]
    wpcm = oracle.wav_pcm16(want)
    assert len(pcm) == len(wpcm) and np.max(np.abs(pcm.astype(int) - wpcm.astype(int))) <= 1 and np.count_nonzero(pcm != wpcm) <= 2   # fp64 on both sides: the same 16-bit samples (a sample exactly on a step may round either way)


def test_config2_single_10s_utterance(oracle, coracle):
    """BASELINE configs[1]: one 10 s utterance @44.1 kHz (342 frames of the survey's chirp + AM tone + noise through ToMel), 60
    iterations, as a caller gets it without a knob: the default schedule on the persistent kernels (the library's own path choice)."""
    import gomel_amd
    fields = dict(T, GriffinLimIterations=60)
    m = mel_obj(fields)
    sig = oracle.config2_signal()
    mel = m.ToMel(sig)
    assert len(mel) == 342 * 192
    cfg = oracle.MelConfig(**fields)
    assert np.max(np.abs(mel - oracle.to_mel(cfg, sig))) < 1e-9             # the forward path at full size too
    x0 = oracle.x0_splitmix(1, oracle.out_len(cfg, len(mel)))
    want, _ = coracle.from_mel(cfg, mel, x0)
    got = m.FromMel(mel.copy(), x0=x0)
    b = gomel_amd.Batch(m.engine(), [len(mel)])
    b.upload([mel], [x0], None)
    tm = b.run_timed()
    assert tm["fast_path"] == 2 and np.array_equal(b.download()[0], got)     # the persistent kernels, and the host-to-host call takes the same path
    b.close()
    print("config 2 (342 frames x 60 iterations), default call vs C oracle: %.2e (signal rms %.3f)" % (rms(got - want), rms(want)))
    assert rms(got - want) < 1e-5
    e = m.engine()
    e.set_precision(64)
    assert rms(e.from_mel(mel, x0=x0) - want) < 1e-9


def test_config3_batch_of_256(oracle, coracle):
    """BASELINE configs[2]: 256 x (192 mels x 400 frames), N 4096 / hop 1280, 60 iterations -- the workload bench.py's `value` is quoted on."""
    _full_size_batch(oracle, coracle, T, 192, "config 3", sample=list(range(0, 256, 11)))          # 24 utterances through the C oracle


def test_config4_tts_geometry(oracle, coracle):
    """BASELINE configs[3]: 256 x 400 frames at N 1024 / hop 256 / 80 mels, 60 iterations."""
    _full_size_batch(oracle, coracle, S, 80, "config 4", sample=list(range(0, 256, 4)))            # 64 utterances through the C oracle


def test_config5_one_rank_shard_of_the_8192_corpus(oracle, coracle):
    """BASELINE configs[4] shards an 8192-utterance corpus over 8 GPUs: 1024 utterances x 400 frames per rank.  This is
    rank 3's shard (utterances 3072 ... 4095, seeds = global indices) on the one GPU here, through gomel_from_mel_batch in the
    calls of 256 bench.py makes, in the default mode: lengths, finiteness, call-to-call determinism, and utterances against the oracle."""
    fields = dict(T, GriffinLimIterations=60)
    U, F, K, first = 1024, 400, 192, 3072
    eng = mel_obj(fields).engine()
    cfg = oracle.MelConfig(**fields)
    L = oracle.out_len(cfg, F * K)
    mels, seeds = cfg3_inputs(U, F, K, first=first)
    outs = []
    for b0 in range(0, U, 256):
        outs += eng.from_mel_batch(mels[b0:b0 + 256], None, seeds[b0:b0 + 256])
    assert len(outs) == U and all(o.shape == (L,) and np.isfinite(o).all() for o in outs)
    again = eng.from_mel_batch(mels[512:768], None, seeds[512:768])
    assert all(np.array_equal(a, b) for a, b in zip(again, outs[512:768]))                      # call-to-call bitwise
    pick = [0, 517, 1023]
    want, _ = coracle.from_mel_batch(cfg, [mels[u] for u in pick], [oracle.x0_splitmix(seeds[u], L) for u in pick], threads=3)
    errs = [rms(outs[u] - w) for u, w in zip(pick, want)]
    print("config 5 shard (utterances %d..%d), default call: rms vs C oracle %s" % (first, first + U - 1, ["%.2e" % e for e in errs]))
    assert max(errs) < RMS_TOL
    eng.set_precision(64)                                       # ... and the same three in the mode that promises the tolerance on every utterance
    o64 = eng.from_mel_batch([mels[u] for u in pick], None, [seeds[u] for u in pick])
    assert max(rms(a - w) for a, w in zip(o64, want)) < 1e-9
    sig = np.array([rms(o) for o in outs])
    assert 0.1 < sig.min() and sig.max() < 0.5                                                  # every utterance a signal of the expected level (0.25)
    eng.close()
