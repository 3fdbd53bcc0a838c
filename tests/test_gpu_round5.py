"""GPU tests added in round 5: the DEFAULT call at FULL size on the geometries the schedule rule had no hard test for -- the
reference's own default geometry (NewMel, mel/mel.go:30-41: the second schedule table) and a geometry on the interpolated stretch of
the overlap rule: a sample of the utterances against the C oracle, EVERY utterance against the fp64 kernels that the sample pins to the
oracle at 1e-9 (tests/test_gpu_configs.py explains the scheme)."""
import math
import os

import numpy as np
import pytest

pytestmark = [pytest.mark.gpu]
RMS_TOL = 1e-4          # north_star: sample-for-sample within 1e-4 RMS


def rms(a):
    return float(np.sqrt(np.mean(np.square(a))))


def mel_obj(**kw):
    import gomel_amd
    m = gomel_amd.NewMel()
    for k, v in kw.items():
        setattr(m, k, v)
    return m


# (Window, expected fp64 iterations of the default call at 60): NewMel's defaults overlap eight-fold -> the second table's 50;
# hop 400 overlaps 5.12-fold -> the interpolated 44 (gomel_hip.hip: auto_fp64_iterations)
@pytest.mark.parametrize("hop,k_expected", [(256, 50), (400, 44)])
def test_default_call_full_size_at_the_newmel_geometry(oracle, coracle, monkeypatch, hop, k_expected):
    """256 x 400 frames, Resolut 2048 / 160 mels / fmax 8000 (`NewMel`, mel/mel.go:30-41) at hop 256 (the reference's default) and
    hop 400, 60 Griffin-Lim iterations, the handle as a caller gets it -- no knob, no environment: EVERY utterance under the HARD bar
    of 1e-4 RMS (against the fp64 kernels, themselves < 1e-9 from the C restatement of mel.go:76-152 on a sample); run-to-run bitwise."""
    import gomel_amd
    monkeypatch.delenv("GOMEL_PRECISION", raising=False)
    m = mel_obj(GriffinLimIterations=60, Window=hop)
    assert (m.NumMels, m.Resolut, m.MelFmax) == (160, 2048, 8000.0)          # NewMel's defaults
    U, F, K = 256, 400, m.NumMels
    mels = [np.random.default_rng(u).uniform(math.log(1e-5), math.log(50.0), size=(F * K, 2)) for u in range(U)]
    seeds = [1000 + u for u in range(U)]
    eng = m.engine()
    assert eng.fp64_iterations() == k_expected
    cfg = oracle.MelConfig(NumMels=K, MelFmax=m.MelFmax, Window=hop, Resolut=m.Resolut, GriffinLimIterations=60)
    L = oracle.out_len(cfg, F * K)
    sample = list(range(0, U, 8))                                           # 32 utterances through the C oracle (seconds); the rest through the fp64 kernels
    want, _ = coracle.from_mel_batch(cfg, [mels[u] for u in sample], [oracle.x0_splitmix(seeds[u], L) for u in sample], threads=min(16, os.cpu_count() or 1))

    def run(bits):
        if bits != "default":
            eng.set_precision(bits)
        b = gomel_amd.Batch(eng, [F * K] * U)
        b.upload(mels, None, seeds)
        b.run()
        o1 = b.download()
        b.run()
        o2 = b.download()
        b.close()
        assert all(np.array_equal(a, c) for a, c in zip(o1, o2))            # run-to-run bitwise (no atomics)
        return o1

    odef = run("default")
    o64 = run(64)
    e64 = np.array([rms(o64[u] - w) for u, w in zip(sample, want)])
    print("N 2048 / hop %d, 64-bit mode vs C oracle on %d sampled utterances: max %.2e" % (hop, len(sample), e64.max()))
    assert e64.max() < 1e-9                                                 # two implementations of the same float64 arithmetic ...
    edef = np.array([rms(a - r) for a, r in zip(odef, o64)])                # ... so the fp64 kernels are the reference for EVERY utterance
    edef_o = np.array([rms(odef[u] - w) for u, w in zip(sample, want)])
    print("N 2048 / hop %d, DEFAULT call (%d fp64 + %d fp32 iterations): vs C oracle on the sample max %.2e; vs the fp64 kernels on all %d: median %.2e p99 %.2e max %.2e" % (
        hop, k_expected, 60 - k_expected, edef_o.max(), U, np.median(edef), np.percentile(edef, 99), edef.max()))
    assert edef_o.max() < RMS_TOL and edef.max() < RMS_TOL - 1e-8, (edef.max(), int(np.argmax(edef)))
