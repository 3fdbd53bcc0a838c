import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from gomel_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import gomel_amd
for N, K, hop in ((4096, 192, 1280), (2048, 160, 256), (1024, 80, 256)):
    m = gomel_amd.NewMel(); m.Resolut, m.NumMels, m.Window = N, K, hop
    e = m.engine()
    fr = np.random.default_rng(0).standard_normal((64, N)).astype(np.float32)
    got = e.rfft_frames(fr); ref = np.fft.rfft(fr.astype(np.float64), axis=1)
    err = np.abs(got - ref); sc = np.abs(ref).max()
    back = e.irfft_frames(ref)
    print(N, "rfft max rel err %.3e  rms rel %.3e | irfft max err %.3e" % (err.max() / sc, np.sqrt((err ** 2).mean()) / sc, np.abs(back - fr).max()),
          "sum", float(np.abs(got).sum()))
