"""CPU oracle (NumPy, fp64) for the reference's package `phase` core: ToPhase / FromPhase.  TEST INFRASTRUCTURE ONLY
(same rules as gomel_oracle.py: only tests/ may import it; the product never does).

PARITY UNPINNED, as for the mel path: the Go reference cannot be built or run here and holds no fixtures for this
path.  The restatement follows phase/phase.go line by line (file:line relative to /root/reference) and was read against the
reference's own NumPy port (phase.py:113-220, which agrees on the STFT pick, the spectrum rebuild and the window-sum
normalisation; it differs only in applying VolumeBoost for > 0 instead of != 0 -- the Go source is followed here).
"""
import numpy as np

import gomel_oracle as go


class PhaseConfig:
    """type Phase + NewPhase defaults, phase/phase.go:8-29."""

    def __init__(self, **kw):
        self.NumFreqs = 768
        self.Window = 1280
        self.Resolut = 4096
        self.YReverse = False
        self.SampleRate = 0
        self.VolumeBoost = 0.0
        self.IHS = False
        self.HDR = False
        for k, v in kw.items():
            if not hasattr(self, k):
                raise AttributeError(k)
            setattr(self, k, v)


def shrink(ospectrum, imels, omels):
    """phase/impl.go:383-391: keep the first omels entries of every block of imels."""
    o = np.asarray(ospectrum, dtype=np.float64)
    j = np.arange(len(o)) % imels
    return o[j < omels]


def grow(ospectrum, imels, omels):
    """phase/impl.go:392-404: after every block of imels entries, repeat its last entry omels - imels times."""
    o = np.asarray(ospectrum, dtype=np.float64)
    if omels <= imels:
        return o.copy()
    F = len(o) // imels
    full = o[:F * imels].reshape(F, imels, 2)
    rep = np.repeat(full[:, -1:, :], omels - imels, axis=1)
    out = np.concatenate([full, rep], axis=1).reshape(F * omels, 2)
    rest = o[F * imels:]                      # a trailing partial block is copied as is (its last index never hits imels-1)
    return np.concatenate([out, rest]) if len(rest) else out


def to_phase(cfg, buf, window=None):
    """(*Phase).ToPhase, phase/phase.go:41-72."""
    N, hop = cfg.Resolut, cfg.Window
    buf = go.pad(np.asarray(buf, dtype=np.float64), hop)         # phase.go:43 (pad is the same function, impl.go:424-450)
    if window is None:
        window = go.hann(N)
    F = go.num_frames(len(buf), N, hop)
    if F <= 0:
        return np.zeros((0, 2))
    idx = np.arange(N)[None, :] + hop * np.arange(F)[:, None]
    S = np.fft.fft(buf[idx] * window[None, :], axis=1)            # phase.go:47 (gossp STFT)
    j = np.arange(N // 2)
    o = np.stack([S[:, j + 1].imag, S[:, N - j - 1].real], axis=2)  # phase.go:52-63
    return shrink(o.reshape(F * (N // 2), 2), N // 2, cfg.NumFreqs)  # phase.go:67


def undospectrum(cfg, ospectrum):
    """(*Phase).undospectrum, phase/phase.go:74-94 (the two writes of j = N/2-1 hit the same bin; the second one stays)."""
    N = cfg.Resolut
    h = N // 2
    F = len(ospectrum) // h
    o = np.asarray(ospectrum, dtype=np.float64)[:F * h].reshape(F, h, 2)
    spec = np.zeros((F, N), dtype=np.complex128)
    j = np.arange(h)
    spec[:, j + 1] = o[:, :, 1] + 1j * o[:, :, 0]
    spec[:, N - j - 1] = o[:, :, 1] - 1j * o[:, :, 0]
    return spec


def istft(window, hop, spectrogram):
    """phase.ISTFT, phase/phase.go:96-134: IFFT, synthesis window, overlap-add, division by the window-square sum where it
    exceeds half its maximum, proportional fade below."""
    F, N = spectrogram.shape
    L = N + (F - 1) * hop
    x = np.zeros(L)
    ws = np.zeros(L)
    y = np.fft.ifft(spectrogram, axis=1).real * window[None, :]
    w2 = window * window
    for i in range(F):
        x[i * hop:i * hop + N] += y[i]
        ws[i * hop:i * hop + N] += w2
    thr = ws.max() * 0.5
    out = x.copy()
    hi = ws > thr
    out[hi] = x[hi] / ws[hi]
    mid = (~hi) & (ws > 1e-21)
    out[mid] = x[mid] / ws[mid] * (ws[mid] / thr)
    return out


def from_phase_len(cfg, n_pairs):
    F = n_pairs // cfg.NumFreqs
    return cfg.Resolut + (F - 1) * cfg.Window


def from_phase(cfg, ospectrum, window=None):
    """(*Phase).FromPhase, phase/phase.go:137-155."""
    if len(ospectrum) == 0:
        raise IndexError("empty spectrogram (reference: index out of range, phase.go:98)")
    if window is None:
        window = go.hann(cfg.Resolut)
    g = grow(ospectrum, cfg.NumFreqs, cfg.Resolut // 2)          # phase.go:141
    out = istft(window, cfg.Window, undospectrum(cfg, g))
    if cfg.VolumeBoost != 0:
        out = out * cfg.VolumeBoost
    return out
