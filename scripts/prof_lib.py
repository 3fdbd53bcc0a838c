"""Run prof_run.py against an alternative build of the library (ablation experiments)."""
import os, sys, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gomel_amd import _lib

def lenient(_lib, path):
    """Older builds of the library lack the newest entry points: drop what a build does not export from the symbol table."""
    import ctypes
    _lib._share_hip_runtime()
    l = ctypes.CDLL(path)
    _lib.SYMBOLS = {k: v for k, v in _lib.SYMBOLS.items() if hasattr(l, k)}
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
lenient(_lib, _lib.LIB_PATH)
sys.argv = [sys.argv[0]] + sys.argv[2:]
exec(open(os.path.join(ROOT, "scripts", "prof_run.py")).read())
