"""Test-session plumbing.

* The process that runs the tests never imports torch: the library is loaded on the HIP runtime it is linked against and ships on
  (/opt/rocm: GOMEL_HIP_RUNTIME=system), exactly what a Go host gets.  The few tests about torch interop / torch.distributed run
  their torch part in CHILD processes (tests/_children.py), collected last, each with a wall-clock budget.
* Collection order (pytest_collection_modifyitems): one oracle comparison per BASELINE config in the DEFAULT arithmetic mode first,
  then every other oracle / fp64 parity test, then the API surface, then thread stress, then the child-process torch tests -- so that
  a stall in an interop test can no longer hide the parity of the hot path (round 5's driver run: VERDICT.md).
* A per-test wall-clock guard: on expiry every thread's stack is written to stderr and the test is failed from a signal handler; a
  main thread that stays stuck in native code for another grace period ends the session with exit status 3 (never a re-exec).
"""
import faulthandler
import os
import signal
import sys
import threading
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

# the product's runtime: /opt/rocm's libamdhip64 through libgomel_hip.so's own RUNPATH, not the copy a PyTorch wheel bundles
os.environ.setdefault("GOMEL_HIP_RUNTIME", "system")

GUARD_DEFAULT_S = float(os.environ.get("GOMEL_TEST_GUARD_S", "240"))
GUARD_GRACE_S = float(os.environ.get("GOMEL_TEST_GUARD_GRACE_S", "30"))


_CONFIG = None


def pytest_configure(config):
    global _CONFIG
    _CONFIG = config
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "guard(seconds): wall-clock budget of this test (default %d s)" % GUARD_DEFAULT_S)
    config.addinivalue_line("markers", "torch_child: the test's torch part runs in a child process; collected last")
    config.addinivalue_line("markers", "stress: thread / concurrency stress; collected after the parity tests")


# ------------------------------------------------------------------------------------------------ collection order
_CONFIG_TESTS = "tests/test_gpu_configs.py"
_PARITY_FILES = ("test_gpu_parity.py", "test_reference_vectors.py")
_PARITY_NAMES = ("test_fused_fp64_kernel_matches_the_oracle", "test_fp64_persistent_equals_fp64_generic_bit_for_bit",
                 "test_fp64_single_utterance_matches_the_oracle_to_rounding", "test_persistent_equals_generic_bit_for_bit",
                 "test_default_mode_on_every_path_and_shape", "test_the_default_call_is_the_calibrated_schedule",
                 "test_default_call_full_size_at_the_newmel_geometry", "test_any_hop_fused_kernel_in_64_bit_mode",
                 "test_any_hop_fused_kernel_matches_generic_and_oracle", "test_precision_schedule_is_fp64_then_fp32",
                 "test_from_mel_against_committed_golden", "test_to_mel_matches_oracle", "test_host_to_host_call_in_the_parity_modes",
                 "test_to_phase_from_phase_match_oracle")


def _tier(item):
    if item.get_closest_marker("gpu") is None:
        return 2                                            # CPU tests keep their file order between themselves
    if item.get_closest_marker("torch_child") is not None:
        return 5
    if item.get_closest_marker("stress") is not None:
        return 4
    path = item.nodeid.split("::")[0]
    name = item.name.split("[")[0]
    if path == _CONFIG_TESTS:
        return 0
    if os.path.basename(path) in _PARITY_FILES or name in _PARITY_NAMES:
        return 1
    return 3


def pytest_collection_modifyitems(session, config, items):
    items.sort(key=_tier)                                   # stable: the order inside a tier stays the collection order


# ------------------------------------------------------------------------------------------------ page-cache warm-up for the torch children
def pytest_collection_finish(session):
    """`import torch` maps ~5 GB of shared objects; on a fresh box whose image pages in lazily that alone has taken minutes.  When GPU
    torch-child tests were selected, one throw-away child imports torch (host side only) while the parity tests run (tests/_children.py)."""
    if any(i.get_closest_marker("torch_child") is not None and i.get_closest_marker("gpu") is not None for i in session.items):
        import _children
        _children.start_torch_warmup()


def pytest_sessionfinish(session, exitstatus):
    if "_children" in sys.modules:
        sys.modules["_children"].stop_torch_warmup()


def pytest_terminal_summary(terminalreporter):
    if "_children" in sys.modules:
        state, secs = sys.modules["_children"].torch_warmup_state()
        if state != "none":
            terminalreporter.write_line("torch warm-up import (child process, host side only): %s after %.1f s" % (state, secs))


# ------------------------------------------------------------------------------------------------ per-test wall-clock guard
class _Guard:
    def __init__(self):
        self.lock = threading.Lock()
        self.deadline = None
        self.nodeid = None
        self.fired = False
        self.thread = None
        self.main = threading.main_thread().ident

    def start(self):
        if self.thread is None:
            self.thread = threading.Thread(target=self._watch, name="test-guard", daemon=True)
            self.thread.start()

    def arm(self, nodeid, seconds):
        with self.lock:
            self.nodeid, self.deadline, self.fired = nodeid, time.monotonic() + seconds, False

    def disarm(self):
        with self.lock:
            self.deadline = None

    def _watch(self):
        while True:
            time.sleep(0.5)
            with self.lock:
                due = self.deadline is not None and time.monotonic() > self.deadline
                nodeid, fired = self.nodeid, self.fired
                if due and not fired:
                    self.fired = True
                    self.deadline = time.monotonic() + GUARD_GRACE_S
            if not due:
                continue
            # what the guard has to say goes to the REAL stderr: pytest's capture holds fd 2 during a test, and a session that ends through
            # os._exit would take the captured text with it
            capman = _CONFIG.pluginmanager.getplugin("capturemanager") if _CONFIG is not None else None
            if capman is not None:
                try:
                    capman.suspend_global_capture(in_=False)
                except Exception:  # noqa: BLE001 -- diagnostics must not depend on the capture plugin's state
                    capman = None
            if not fired:
                sys.stderr.write("\n[test guard] %s exceeded its wall-clock budget; stacks of all threads follow\n" % nodeid)
                faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
                sys.stderr.flush()
                if capman is not None:
                    try:
                        capman.resume_global_capture()
                    except Exception:  # noqa: BLE001
                        pass
                signal.pthread_kill(self.main, signal.SIGALRM)       # fails the test if the interpreter gets control back
            else:
                sys.stderr.write("\n[test guard] %s: the main thread is still inside native code %d s after the alarm; ending the session "
                                 "(exit status 3)\n" % (nodeid, GUARD_GRACE_S))
                sys.stderr.flush()
                os._exit(3)


_guard = _Guard()


class TestBudgetExceeded(Exception):
    pass


def _on_alarm(signum, frame):
    raise TestBudgetExceeded("wall-clock budget exceeded (tests/conftest.py guard)")


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_call(item):
    if threading.current_thread() is not threading.main_thread():
        yield
        return
    m = item.get_closest_marker("guard")
    seconds = float(m.args[0]) if m and m.args else GUARD_DEFAULT_S
    old = signal.signal(signal.SIGALRM, _on_alarm)
    _guard.start()
    _guard.arm(item.nodeid, seconds)
    try:
        yield
    finally:
        _guard.disarm()
        signal.signal(signal.SIGALRM, old)


# ------------------------------------------------------------------------------------------------ fixtures
@pytest.fixture(scope="session")
def oracle():
    import gomel_oracle
    return gomel_oracle


@pytest.fixture(scope="session")
def coracle():
    """The C restatement (oracle/libgomel_oracle.so), built on demand."""
    import oracle_c
    return oracle_c.load()


@pytest.fixture
def fp32_start_mode(monkeypatch):
    """Since round 4 a handle's default arithmetic is the calibrated fp64-then-fp32 schedule (gomel_set_precision(h, 0)).  The test
    modules written for the fp32 kernels -- now the opt-in fast mode -- start their handles in that mode through the library's
    GOMEL_PRECISION knob (read at gomel_create); a test that selects a mode itself overrides it.  tests/test_gpu_configs.py,
    tests/test_gpu_round4.py and test_gpu_round5.py check what a caller gets WITHOUT any knob."""
    monkeypatch.setenv("GOMEL_PRECISION", "32")


@pytest.fixture(scope="session")
def hip():
    """The HIP runtime the library is running on, for the few raw calls a test needs (hipGetDevice, hipMalloc ...): resolved from
    the objects already mapped into this process, so it is the same copy libgomel_hip.so uses."""
    import ctypes
    import gomel_amd
    gomel_amd._lib.lib()
    path = gomel_amd.hip_runtime_path()
    assert path, "no libamdhip64 mapped after loading libgomel_hip.so"
    return ctypes.CDLL(path)
