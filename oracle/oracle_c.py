"""ctypes wrapper of oracle/gomel_oracle.c (test infrastructure / CPU baseline only)."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libgomel_oracle.so")
dp = ctypes.POINTER(ctypes.c_double)


class Cfg(ctypes.Structure):
    _fields_ = [("num_mels", ctypes.c_int32), ("window", ctypes.c_int32), ("resolut", ctypes.c_int32), ("iterations", ctypes.c_int32),
                ("mel_fmin", ctypes.c_double), ("mel_fmax", ctypes.c_double), ("tune_mul", ctypes.c_double), ("tune_add", ctypes.c_double)]


def build(force=False):
    src = os.path.join(HERE, "gomel_oracle.c")
    if force or not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
        subprocess.run(["gcc", "-O2", "-fPIC", "-fopenmp", "-shared", "-o", SO, src, "-lm"], check=True)
    return SO


class COracle:
    def __init__(self, lib):
        self.lib = lib
        lib.oracle_out_len.restype = ctypes.c_size_t
        lib.oracle_to_mel_pairs.restype = ctypes.c_size_t
        lib.oracle_pad_len.restype = ctypes.c_size_t
        lib.oracle_out_len.argtypes = [ctypes.POINTER(Cfg), ctypes.c_size_t]
        lib.oracle_to_mel_pairs.argtypes = [ctypes.POINTER(Cfg), ctypes.c_size_t]
        lib.oracle_pad_len.argtypes = [ctypes.c_size_t, ctypes.c_int]

    @staticmethod
    def cfg(c):
        return Cfg(c.NumMels, c.Window, c.Resolut, c.GriffinLimIterations, c.MelFmin, c.MelFmax, c.TuneMul, c.TuneAdd)

    def hann(self, n, symmetric=True):
        w = np.zeros(n)
        self.lib.oracle_hann(w.ctypes.data_as(dp), n, 1 if symmetric else 0)
        return w

    def fft(self, z, inverse=False):
        re = np.ascontiguousarray(z.real, dtype=np.float64).copy()
        im = np.ascontiguousarray(z.imag, dtype=np.float64).copy()
        rc = self.lib.oracle_fft(re.ctypes.data_as(dp), im.ctypes.data_as(dp), len(re), 1 if inverse else 0)
        if rc:
            raise ValueError("oracle_fft rc=%d" % rc)
        return re + 1j * im

    def from_mel(self, c, mel, x0, window=None):
        """Returns (wave, exp'd mel copy).  Raises IndexError/ValueError like the reference would panic."""
        cc = self.cfg(c)
        mel = np.ascontiguousarray(mel, dtype=np.float64).copy()
        if len(mel) == 0 or len(mel) % c.NumMels:
            raise IndexError("bad length")
        w = self.hann(c.Resolut) if window is None else np.ascontiguousarray(window, dtype=np.float64)
        L = self.lib.oracle_out_len(ctypes.byref(cc), len(mel))
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        assert x0.shape == (L,)
        out = np.zeros(L)
        rc = self.lib.oracle_from_mel(ctypes.byref(cc), mel.ctypes.data_as(dp), ctypes.c_size_t(len(mel)), w.ctypes.data_as(dp),
                                      x0.ctypes.data_as(dp), out.ctypes.data_as(dp))
        if rc:
            raise IndexError("oracle_from_mel rc=%d" % rc)
        return out, mel

    def from_mel_batch(self, c, mels, x0s, threads=0, window=None):
        cc = self.cfg(c)
        n = len(mels)
        ms = [np.ascontiguousarray(m, dtype=np.float64).copy() for m in mels]
        xs = [np.ascontiguousarray(x, dtype=np.float64) for x in x0s]
        w = self.hann(c.Resolut) if window is None else np.ascontiguousarray(window, dtype=np.float64)
        outs = [np.zeros(self.lib.oracle_out_len(ctypes.byref(cc), len(m))) for m in ms]
        mp = (dp * n)(*[m.ctypes.data_as(dp) for m in ms])
        xp = (dp * n)(*[x.ctypes.data_as(dp) for x in xs])
        op = (dp * n)(*[o.ctypes.data_as(dp) for o in outs])
        npairs = (ctypes.c_size_t * n)(*[len(m) for m in ms])
        used = self.lib.oracle_from_mel_batch(ctypes.byref(cc), n, mp, npairs, w.ctypes.data_as(dp), xp, op, int(threads))
        if used < 0:
            raise IndexError("oracle_from_mel_batch rc=%d" % -used)
        return outs, used

    def to_mel(self, c, wave, window=None):
        cc = self.cfg(c)
        wave = np.ascontiguousarray(wave, dtype=np.float64)
        w = self.hann(c.Resolut) if window is None else np.ascontiguousarray(window, dtype=np.float64)
        n = self.lib.oracle_to_mel_pairs(ctypes.byref(cc), ctypes.c_size_t(len(wave)))
        out = np.zeros((n, 2))
        rc = self.lib.oracle_to_mel(ctypes.byref(cc), wave.ctypes.data_as(dp), ctypes.c_size_t(len(wave)), w.ctypes.data_as(dp), out.ctypes.data_as(dp))
        if rc:
            raise IndexError("oracle_to_mel rc=%d" % rc)
        return out


def load():
    return COracle(ctypes.CDLL(build()))
