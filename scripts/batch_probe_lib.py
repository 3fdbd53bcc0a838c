import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gomel_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = [sys.argv[0]] + sys.argv[2:]
exec(open(os.path.join(ROOT, "scripts", "batch_probe.py")).read())
