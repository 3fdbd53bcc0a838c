"""Host-side mirror of the reference's Go package `phase` (phase/phase.go) for its two numeric entry points, ToPhase and
FromPhase, over the C ABI (gomel_to_phase / gomel_from_phase in include/gomel_hip.h).  The phase package's PNG variants (HDR,
inverse-hyperbolic-sine passes) and its zero-stuffing resampler are host-side file formats outside the accelerated path and
are not mirrored here."""
import ctypes

import numpy as np

from . import _lib
from ._lib import GomelConfig, c_double_p, check
from .mel import Engine, ErrFileNotLoaded  # noqa: F401  (same sentinel as phase/phase.go:39)


class Phase:
    """`type Phase` (phase/phase.go:8-18); fields may be changed between calls."""

    def __init__(self):
        # NewPhase defaults, phase/phase.go:21-28
        self.NumFreqs = 768
        self.Window = 1280
        self.Resolut = 4096
        self.YReverse = False
        self.SampleRate = 0
        self.VolumeBoost = 0.0
        self.IHS = False
        self.HDR = False
        self.Device = -1
        self._engines = {}

    def _config(self):
        c = GomelConfig()
        _lib.lib().gomel_default_config(ctypes.byref(c))
        c.window, c.resolut, c.device = int(self.Window), int(self.Resolut), int(self.Device)
        c.num_mels = 1                      # the mel tables of the handle are not used by the phase entry points
        return c

    def engine(self):
        c = self._config()
        key = (c.window, c.resolut, c.device)
        e = self._engines.get(key)
        if e is None:
            e = Engine(c)
            self._engines = {key: e}
        return e

    # phase/phase.go:41-72
    def ToPhase(self, buf):
        e = self.engine()
        w = np.ascontiguousarray(buf, dtype=np.float64)
        L = _lib.lib()
        cap = L.gomel_phase_pairs(ctypes.byref(e.cfg), int(self.NumFreqs), w.size)
        out = np.empty((max(cap, 1), 2), dtype=np.float64)
        got = ctypes.c_size_t(0)
        check(L.gomel_to_phase(e._h, int(self.NumFreqs), w.ctypes.data_as(c_double_p), w.size, out.ctypes.data_as(c_double_p), cap, ctypes.byref(got)),
              "gomel_to_phase")
        return out[:got.value]

    # phase/phase.go:137-155
    def FromPhase(self, ospectrum):
        e = self.engine()
        p = np.ascontiguousarray(ospectrum, dtype=np.float64)
        if p.ndim != 2 or p.shape[1] != 2:
            raise ValueError("expected an [n, 2] float64 array (Go [][2]float64)")
        L = _lib.lib()
        n = L.gomel_phase_out_len(ctypes.byref(e.cfg), int(self.NumFreqs), len(p)) if len(p) else 0
        out = np.empty(max(n, 1), dtype=np.float64)
        got = ctypes.c_size_t(0)
        check(L.gomel_from_phase(e._h, int(self.NumFreqs), float(self.VolumeBoost), p.ctypes.data_as(c_double_p), len(p),
                                 out.ctypes.data_as(c_double_p), out.size, ctypes.byref(got)), "gomel_from_phase")
        return out[:got.value]


def NewPhase():
    """phase/phase.go:21-28."""
    return Phase()
