"""CPU tests of the test session's own plumbing (tests/conftest.py, tests/_children.py): a harness that is there to make a stall visible and
bounded must itself be known to work -- the per-test wall-clock guard, the collection order, and what happens to a child process that raises,
overruns its budget or dies."""
import os
import subprocess
import sys
import textwrap
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_pytest(tmp_path, body, *args, env=None):
    """A throw-away test directory that uses THIS repository's conftest.py."""
    d = tmp_path / "t"
    d.mkdir()
    (d / "conftest.py").write_text(open(os.path.join(ROOT, "tests", "conftest.py")).read().replace(
        'ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))', 'ROOT = %r' % ROOT))
    (d / "_children.py").write_text(open(os.path.join(ROOT, "tests", "_children.py")).read())
    (d / "test_x.py").write_text(textwrap.dedent(body))
    return subprocess.run([sys.executable, "-m", "pytest", str(d), "-q", "-p", "no:cacheprovider", *args], capture_output=True, text=True, timeout=300,
                          env=dict(os.environ, **(env or {})), cwd=str(d))


def test_guard_fails_a_test_that_overruns_its_budget_and_the_session_goes_on(tmp_path):
    r = _run_pytest(tmp_path, """
        import time, pytest
        @pytest.mark.guard(1)
        def test_sleeps():
            time.sleep(30)
        def test_after():
            assert True
    """)
    assert r.returncode == 1, (r.returncode, r.stdout[-800:], r.stderr[-800:])
    assert "1 failed, 1 passed" in r.stdout
    assert "TestBudgetExceeded" in r.stdout and "[test guard]" in r.stderr and "test_sleeps" in r.stderr       # the stacks went to stderr


def test_guard_ends_a_session_whose_main_thread_is_stuck_in_native_code(tmp_path):
    """A main thread that never returns to the interpreter (here: blocked in a C-level read with signals restarted) cannot be failed by an
    exception: after the grace period the guard ends the session with exit status 3 instead of hanging until the driver's limit."""
    r = _run_pytest(tmp_path, """
        import ctypes, os, pytest, signal
        @pytest.mark.guard(1)
        def test_stuck():
            signal.siginterrupt(signal.SIGALRM, False)
            rd, wr = os.pipe()
            libc = ctypes.CDLL(None)
            buf = ctypes.create_string_buffer(8)
            libc.read(rd, buf, 8)            # blocks in native code: nobody writes
    """, env={"GOMEL_TEST_GUARD_GRACE_S": "2"})
    assert r.returncode == 3, (r.returncode, r.stdout[-500:], r.stderr[-1200:])
    assert "still inside native code" in r.stderr


def test_collection_order_puts_configs_first_and_torch_children_last(tmp_path):
    r = _run_pytest(tmp_path, """
        import pytest
        pytestmark = pytest.mark.gpu
        @pytest.mark.torch_child
        def test_a_child(): pass
        @pytest.mark.stress
        def test_b_stress(): pass
        def test_c_api(): pass
        def test_from_mel_against_committed_golden(): pass
    """, "--collect-only", "-m", "gpu")
    names = [l.split("::")[-1] for l in r.stdout.splitlines() if "::" in l]
    assert names == ["test_from_mel_against_committed_golden", "test_c_api", "test_b_stress", "test_a_child"], r.stdout


def _raises(rep):
    rep.stage("about to raise")
    raise RuntimeError("boom")


def _overruns(rep):
    rep.stage("sleeping")
    time.sleep(60)


def _dies(rep):
    rep.stage("dying")
    os._exit(7)


def _fine(rep, x):
    rep.result(x * 2)


def test_children_report_failures_by_stage(monkeypatch):
    """run_child: a result comes back; a child that raises fails the test with its traceback; one that overruns its budget is killed and the
    failure names the last stage it reported; one that dies is reported with its exit code.  (The children import torch first -- host side only.)"""
    import _children
    assert _children.run_child(_fine, (21,)) == 42
    with pytest.raises(pytest.fail.Exception) as e1:
        _children.run_child(_raises)
    assert "boom" in str(e1.value)
    monkeypatch.setattr(_children, "RUN_BUDGET_S", 2.0)
    t0 = time.time()
    with pytest.raises(pytest.fail.Exception) as e2:
        _children.run_child(_overruns, run_budget_s=2.0)
    assert "sleeping" in str(e2.value) and time.time() - t0 < 50
    with pytest.raises(pytest.fail.Exception) as e3:
        _children.run_child(_dies)
    assert "7" in str(e3.value) and "dying" in str(e3.value)


def test_a_child_still_importing_torch_is_a_skip_not_a_failure(monkeypatch):
    import _children
    monkeypatch.setattr(_children, "IMPORT_BUDGET_S", 0.0)          # nobody imports torch in no time
    with pytest.raises(pytest.skip.Exception) as e:
        _children.run_child(_fine, (1,))
    assert "import torch" in str(e.value)
