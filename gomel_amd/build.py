"""Builds libgomel_hip.so (the C-ABI library, HIP for gfx950) in-tree with hipcc.  `python -m gomel_amd.build`."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "gomel_hip.hip")
DEPS = [SRC, os.path.join(HERE, "csrc", "gl_fft.hpp"), os.path.join(HERE, "csrc", "gl_kernels.hpp"),
        os.path.join(HERE, "..", "include", "gomel_hip.h")]
OUT = os.path.join(HERE, "libgomel_hip.so")


def hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 into gomel_amd/libgomel_hip.so.  Returns the path."""
    if not force and not needs_build():
        return OUT
    cmd = [hipcc(), "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value",
           "-o", OUT + ".tmp", SRC]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    os.replace(OUT + ".tmp", OUT)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
