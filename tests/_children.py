"""Child processes for the tests whose subject needs torch (torch CUDA tensors handed to the C ABI, torch.distributed ranks).

The process that runs pytest never imports torch (tests/conftest.py): the library under test runs on the HIP runtime it ships on.  A
test marked `torch_child` starts its torch part here -- stdlib multiprocessing, spawn context, the children import torch FIRST (the
dynamic loader then resolves libgomel_hip.so's libamdhip64.so.7 to torch's loaded copy: one runtime per process) -- and watches it with a
wall-clock budget.  Children report stages; a child that has not finished is killed by PID and the test fails naming the last stage
reached, except when that stage is still "importing torch": `import torch` maps ~5 GB of shared objects and has taken minutes on a
fresh box whose image pages in lazily -- that says nothing about this library, so the test is SKIPPED with the time it took.
"""
import multiprocessing as mp
import os
import subprocess
import sys
import time

import pytest

IMPORT_BUDGET_S = float(os.environ.get("GOMEL_TEST_TORCH_IMPORT_S", "300"))
RUN_BUDGET_S = float(os.environ.get("GOMEL_TEST_CHILD_S", "240"))

_warm = {"proc": None, "t0": 0.0}


def start_torch_warmup():
    """One throw-away child imports torch (host side only: `import torch` opens no GPU) while the parity tests run, so that the
    children at the end of the session find torch's shared objects in the page cache."""
    if _warm["proc"] is not None:
        return
    try:
        _warm["t0"] = time.time()
        _warm["proc"] = subprocess.Popen([sys.executable, "-c", "import torch"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                         env=dict(os.environ, OMP_NUM_THREADS="1"))
    except OSError:
        _warm["proc"] = None


def stop_torch_warmup():
    p = _warm["proc"]
    if p is not None and p.poll() is None:
        p.kill()                                            # the exact child this session started
        p.wait()


def torch_warmup_state():
    """(state, seconds): 'none' | 'running' | 'done' | 'failed', and how long the warm-up import has run / ran."""
    p = _warm["proc"]
    if p is None:
        return "none", 0.0
    rc = p.poll()
    if rc is None:
        return "running", time.time() - _warm["t0"]
    if "t1" not in _warm:
        _warm["t1"] = time.time()
    return ("done" if rc == 0 else "failed"), _warm["t1"] - _warm["t0"]


def _await_warmup():
    """Before the first torch child: give a warm-up import that is still running the rest of its budget."""
    while True:
        state, secs = torch_warmup_state()
        if state != "running":
            return state, secs
        if secs > IMPORT_BUDGET_S:
            stop_torch_warmup()
            pytest.skip("`import torch` (host side only, a child started at session begin) had not finished after %.0f s on this box: "
                        "the torch-interop part of this test was not reached; nothing of libgomel_hip.so was involved" % secs)
        time.sleep(1.0)


class Reporter:
    """What a child gets as its first argument: stage(name) and result(value) go to the parent -- synchronously, down a pipe (a queue's feeder
    thread would lose the last reports of a child that dies)."""

    def __init__(self, conn, rank=0):
        self.conn, self.rank = conn, rank

    def stage(self, name):
        self.conn.send(("stage", self.rank, name, time.time()))

    def result(self, value):
        self.conn.send(("result", self.rank, value, time.time()))


def _child_entry(target, conn, rank, args):
    rep = Reporter(conn, rank)
    rep.stage("importing torch")
    try:
        import torch  # noqa: F401  (first: see the module docstring)
        rep.stage("torch imported")
        target(rep, *args)
    except BaseException as ex:  # noqa: BLE001
        import traceback
        conn.send(("error", rank, "%r\n%s" % (ex, traceback.format_exc()), time.time()))
        raise


def run_children(target, per_rank_args, run_budget_s=None):
    """Start one child per entry of `per_rank_args`, each running target(reporter, *args) after importing torch.  Returns the list of
    results by rank.  See the module docstring for what happens to a child that does not finish."""
    from multiprocessing.connection import wait as wait_conns
    run_budget_s = RUN_BUDGET_S if run_budget_s is None else run_budget_s
    _await_warmup()
    ctx = mp.get_context("spawn")
    n = len(per_rank_args)
    pipes = [ctx.Pipe(duplex=False) for _ in range(n)]
    procs = [ctx.Process(target=_child_entry, args=(target, pipes[r][1], r, tuple(a))) for r, a in enumerate(per_rank_args)]
    t0 = time.time()
    for p in procs:
        p.start()
    for _, w in pipes:
        w.close()                                               # the parent's copy of the write end: a dead child then reads as EOF
    readers = {pipes[r][0]: r for r in range(n)}
    stages = {r: ("started", t0) for r in range(n)}
    results = {}
    failure = None
    try:
        while len(results) < n and failure is None:
            importing = [r for r in range(n) if r not in results and stages[r][0] in ("started", "importing torch")]
            if importing:
                deadline, what = t0 + IMPORT_BUDGET_S, "skip"
            else:                                               # everybody has torch: the clock of the part under test runs from the last report
                deadline, what = max(t for s, t in stages.values()) + run_budget_s, "fail"
            names = {r: s for r, (s, _) in stages.items()}
            ready = wait_conns(list(readers), timeout=max(0.05, min(deadline - time.time(), 2.0))) if readers else []
            for c in ready:
                try:
                    kind, rank, payload, t = c.recv()
                except (EOFError, OSError):                     # the child is gone and its pipe is drained
                    r = readers.pop(c)
                    procs[r].join(timeout=10)
                    if r not in results and failure is None:
                        failure = ("fail", "child %d exited with %s before reporting a result; last stages: %s" % (r, procs[r].exitcode, names))
                    continue
                if kind == "stage":
                    stages[rank] = (payload, time.time())
                elif kind == "result":
                    results[rank] = payload
                elif failure is None:
                    failure = ("fail", "child %d raised: %s" % (rank, payload))
            if not ready and failure is None and time.time() > deadline:
                if what == "skip":
                    failure = ("skip", "`import torch` had not finished in child %s after %.0f s on this box: the torch part of this test was not "
                                       "reached; nothing of libgomel_hip.so was involved" % (importing, time.time() - t0))
                else:
                    failure = ("fail", "child processes exceeded their %.0f s budget after importing torch; last stages: %s" % (run_budget_s, names))
    finally:
        for p in procs:
            p.join(timeout=60 if failure is None else 0.1)
            if p.is_alive():
                p.kill()                                        # the exact children this call started
                p.join()
    if failure is not None:
        (pytest.skip if failure[0] == "skip" else pytest.fail)(failure[1])
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    return [results[r] for r in range(n)]


def run_child(target, args=(), run_budget_s=None):
    return run_children(target, [args], run_budget_s)[0]
