"""Builds libgomel_hip.so (the C-ABI library) in-tree: the HIP translation unit (kernels, launches, device entry points) with hipcc
for gfx950, the device-free host logic (csrc/gomel_host.cpp: table builder, schedule, planners, registry, host-only entry points) with
g++ as plain C++17, linked into one shared object.  `python -m gomel_amd.build [--force]`."""
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SRC = os.path.join(CSRC, "gomel_hip.hip")
HOST_SRC = os.path.join(CSRC, "gomel_host.cpp")
DEPS = [SRC, HOST_SRC] + [os.path.join(CSRC, n) for n in ("gomel_host.hpp", "gl_types.hpp", "gl_fft.hpp", "gl_kernels.hpp", "gl_any.hpp", "gl_big.hpp")] + [
    os.path.join(HERE, "..", "include", "gomel_hip.h")]
OUT = os.path.join(HERE, "libgomel_hip.so")
# -target-feature -load-store-opt: switches off the SI load/store optimizer for the device code.  It merges neighbouring
# ds_read_b64 into ds_read2(st64)_b64, which the LDS serves at half the rate of two single reads (8 vs 2 x 2 array cycles per
# wave-instruction, MI355X_MICROARCH.md "LDS"); 66 of the 90 LDS reads per frame of k_gl_fused were merged that way.  The
# host pass of hipcc does not know the feature and says so in a warning (-Wno-... has no switch for it): harmless.
HIP_FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Xclang", "-target-feature", "-Xclang", "-load-store-opt"]
HOST_FLAGS = ["-O2", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def needs_build(out=OUT):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in DEPS)


def _run(cmd, verbose):
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
    err = "\n".join(l for l in r.stderr.splitlines() if "is not a recognized feature for this target" not in l)
    if err.strip():
        print(err, file=sys.stderr)
    if r.returncode:
        raise subprocess.CalledProcessError(r.returncode, cmd)


def build(force=False, verbose=False, out=OUT, extra_flags=()):
    """Compile every source into `out` (default gomel_amd/libgomel_hip.so).  extra_flags: -D knobs of an A/B build (both
    translation units get them).  Returns the path."""
    if not force and not extra_flags and not needs_build(out):
        return out
    with tempfile.TemporaryDirectory(prefix="gomel_build_") as tmp:
        host_o, hip_o = os.path.join(tmp, "gomel_host.o"), os.path.join(tmp, "gomel_hip.o")
        cxx = shutil.which("g++") or shutil.which("c++")
        host_cmd = ([cxx] if cxx else [hipcc(), "-x", "c++"]) + HOST_FLAGS + list(extra_flags) + ["-c", HOST_SRC, "-o", host_o]   # (no g++: the host pass of hipcc's clang)
        if verbose:
            print(" ".join(host_cmd), file=sys.stderr)
        host = subprocess.Popen(host_cmd, stderr=subprocess.PIPE, text=True)                 # beside the device compile
        try:
            _run([hipcc()] + HIP_FLAGS + list(extra_flags) + ["-c", SRC, "-o", hip_o], verbose)
            _, host_err = host.communicate()
            if host_err.strip():
                print(host_err, file=sys.stderr)
            if host.returncode:
                raise subprocess.CalledProcessError(host.returncode, host_cmd)
        finally:
            if host.poll() is None:        # the device compile failed: do not leave the host compile writing into a directory that is about to go
                host.kill()
                host.communicate()
        _run([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out + ".tmp", hip_o, host_o], verbose)
    os.replace(out + ".tmp", out)
    return out


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "--force"]
    out = OUT
    if "-o" in args:
        i = args.index("-o"); out = os.path.abspath(args[i + 1]); del args[i:i + 2]
    print(build(force="--force" in sys.argv, verbose=True, out=out, extra_flags=tuple(args)))
