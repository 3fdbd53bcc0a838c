"""Package phase: CPU tests of the oracle restatement, GPU parity tests of gomel_to_phase / gomel_from_phase."""
import numpy as np
import pytest


def rms(a):
    return float(np.sqrt(np.mean(np.square(a))))


@pytest.fixture(scope="module")
def po():
    import phase_oracle
    return phase_oracle


def test_shrink_grow_semantics(po):
    o = np.arange(24.0).reshape(12, 2)
    s = po.shrink(o, 4, 3)                                   # phase/impl.go:383-391
    assert s[:, 0].tolist() == [0, 2, 4, 8, 10, 12, 16, 18, 20]
    g = po.grow(s, 3, 5)                                     # phase/impl.go:392-404: the last pair of every block repeated
    assert g[:, 0].tolist() == [0, 2, 4, 4, 4, 8, 10, 12, 12, 12, 16, 18, 20, 20, 20]
    assert np.array_equal(po.grow(o, 4, 4), o) and np.array_equal(po.shrink(o, 4, 4), o)


def test_oracle_round_trip_and_normalisation(oracle, po):
    cfg = po.PhaseConfig(NumFreqs=2048)
    sig = oracle.config2_signal(seconds=0.8)
    p = po.to_phase(cfg, sig)
    F = oracle.num_frames(len(oracle.pad(sig, 1280)), 4096, 1280)
    assert p.shape == (F * 2048, 2)
    w = po.from_phase(cfg, p)
    assert len(w) == po.from_phase_len(cfg, len(p)) == 4096 + (F - 1) * 1280
    pad = oracle.pad(sig, 1280)
    core = slice(4096, len(w) - 4096)
    # all N/2 pairs kept: only each frame's DC is lost (bin 0 is never stored, phase.go:52-56)
    assert rms(w[core] - pad[core]) < 1e-3 * rms(pad[core]) + 3e-4
    # the window-sum rule: interior divided by sum w^2 (1.19..1.21 at hop 1280), edges faded, never amplified
    spec = np.zeros((F, 4096), dtype=complex)
    ones = po.istft(oracle.hann(4096), 1280, spec)
    assert np.all(ones == 0)
    cfg.VolumeBoost = 2.0
    assert np.allclose(po.from_phase(cfg, p), 2.0 * w)


@pytest.mark.gpu
@pytest.mark.parametrize("N,hop,nf", [(4096, 1280, 768), (4096, 1280, 2048), (2048, 256, 836), (1024, 256, 512), (4096, 1280, 1), (2048, 275, 700), (1024, 441, 300), (512, 128, 200), (512, 160, 256), (256, 64, 100)])
def test_to_phase_from_phase_match_oracle(oracle, po, N, hop, nf):
    import gomel_amd
    m = gomel_amd.NewPhase()
    m.NumFreqs, m.Window, m.Resolut = nf, hop, N
    cfg = po.PhaseConfig(NumFreqs=nf, Window=hop, Resolut=N)
    sig = oracle.config2_signal(seconds=0.6)
    want = po.to_phase(cfg, sig)
    got = m.ToPhase(sig)
    assert got.shape == want.shape
    assert np.max(np.abs(got - want)) < 1e-9 * max(1.0, float(np.abs(want).max()))
    back_want = po.from_phase(cfg, want)
    back = m.FromPhase(want)
    assert back.shape == back_want.shape
    assert np.max(np.abs(back - back_want)) < 1e-11 * max(1.0, float(np.abs(back_want).max())) + 1e-12
    m.VolumeBoost = 1.666
    cfg.VolumeBoost = 1.666
    assert np.max(np.abs(m.FromPhase(want) - po.from_phase(cfg, want))) < 1e-11


@pytest.mark.gpu
def test_phase_errors_and_defaults():
    import gomel_amd
    m = gomel_amd.NewPhase()
    assert (m.NumFreqs, m.Window, m.Resolut, m.VolumeBoost) == (768, 1280, 4096, 0)     # phase/phase.go:21-28
    with pytest.raises(gomel_amd.GomelError) as ei:
        m.FromPhase(np.zeros((100, 2)))                       # not a multiple of NumFreqs
    assert ei.value.status == 3
    with pytest.raises(gomel_amd.GomelError) as ei:
        m.FromPhase(np.zeros((0, 2)))
    assert ei.value.status == 4
    m.NumFreqs = 5000                                          # more than N/2: ToPhase keeps all N/2 pairs, FromPhase refuses
    sig = np.random.default_rng(0).standard_normal(30000)
    assert len(m.ToPhase(sig)) % 2048 == 0
    with pytest.raises(gomel_amd.GomelError):
        m.FromPhase(np.zeros((5000, 2)))
