"""The host-side logic of libgomel_hip.so under threads, on a machine without a GPU (the handle must be safe under
concurrent cgo calls; the reference itself is single-goroutine): tests/host_stress.cpp -- pin registry against a fake backend, table
builder, chunk planner, automatic schedule, device-buffer cache against host malloc -- built with g++ against the product library
and run.  The same program under AddressSanitizer + UBSan and under ThreadSanitizer is scripts/sanitize_host.sh (a clean log of the
final build: profiles/r04/sanitizers.log); GOMEL_RUN_SANITIZERS=1 runs that script from here (about five minutes)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_stress_program(tmp_path):
    from gomel_amd import _lib
    _lib.lib()
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("g++ not available")
    exe = str(tmp_path / "host_stress")
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.run([gxx, "-O1", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-Wno-unused-function", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "host_stress.cpp"),
                    "-o", exe, "-L", libdir, "-lgomel_hip", "-Wl,-rpath," + libdir, "-pthread"], check=True)
    r = subprocess.run([exe, "150"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "host_stress: 0 failures" in r.stdout and "violations 0" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-2000:])


@pytest.mark.skipif(not os.environ.get("GOMEL_RUN_SANITIZERS"), reason="set GOMEL_RUN_SANITIZERS=1 (builds the library twice with sanitizers: ~5 minutes)")
def test_host_side_is_clean_under_the_sanitizers(tmp_path):
    r = subprocess.run(["bash", os.path.join(ROOT, "scripts", "sanitize_host.sh"), str(tmp_path / "san"), "300"], cwd=ROOT, capture_output=True, text=True, timeout=3000)
    log = open(str(tmp_path / "san" / "sanitizers.log")).read()
    assert r.returncode == 0 and "== result: CLEAN" in log, log[-3000:]
