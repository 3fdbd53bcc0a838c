"""CPU tests of the multi-GPU plumbing with gloo, world_size 2: utterance sharding is invisible in the results, and the
table blob ("mel basis") broadcast delivers rank 0's bytes.  The compute callable is the oracle here (no GPU); on a GPU
box the same functions get Engine.from_mel_batch."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_properties():
    from gomel_amd.dist import shard_bounds
    rng = np.random.default_rng(0)
    for world in (1, 2, 3, 4, 8):
        for n in (0, 1, 5, 8, 8192, 37):
            fr = rng.integers(1, 900, size=n).tolist()
            b = shard_bounds(fr, world)
            assert len(b) == world + 1 and b[0] == 0 and b[-1] == n and all(b[i] <= b[i + 1] for i in range(world))
            if n >= 4 * world:
                loads = [sum(fr[b[i]:b[i + 1]]) for i in range(world)]
                assert max(loads) - min(loads) <= 2 * max(fr)       # balanced by frames to within an utterance or two
    assert shard_bounds([400] * 8192, 8) == [1024 * i for i in range(9)]   # BASELINE config 5


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch.distributed as dist
    import gomel_oracle as go
    from gomel_amd import dist as gd
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        cfg = go.MelConfig(NumMels=8, MelFmax=8000.0, Window=20, Resolut=64, GriffinLimIterations=3)
        frames = [5, 9, 3, 12, 7, 4, 6]
        mels = [go.random_logmel(f, 8, seed=i) for i, f in enumerate(frames)]

        def compute(ms, x0s, seeds):
            return [go.from_mel(cfg, m.copy(), go.x0_splitmix(s, go.out_len(cfg, len(m)))) for m, s in zip(ms, seeds)]

        outs = gd.from_mel_sharded(compute, mels, None, None, rank, world)
        blob = gd.broadcast_bytes(b"mel-basis-\x00\x01\x02" * 1000 if rank == 0 else None, rank, world)

        class FakeEngine:
            def __init__(self, tables=None):
                self.tables = tables if tables is not None else b"tables-built-by-rank-0"

            def export_tables(self):
                return self.tables

        eng = gd.create_engine_broadcast(lambda tables=None: FakeEngine(tables), rank, world)
        q.put((rank, [o.tobytes() for o in outs], blob, eng.tables))
    finally:
        dist.destroy_process_group()


def test_sharded_results_equal_single_process_gloo_world2(oracle):
    import multiprocessing as mp          # (stdlib: the process running pytest never imports torch, tests/conftest.py)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cfg = oracle.MelConfig(NumMels=8, MelFmax=8000.0, Window=20, Resolut=64, GriffinLimIterations=3)
    frames = [5, 9, 3, 12, 7, 4, 6]
    single = [oracle.from_mel(cfg, oracle.random_logmel(f, 8, seed=i), oracle.x0_splitmix(i, oracle.out_len(cfg, f * 8))).tobytes()
              for i, f in enumerate(frames)]
    for rank, outs, blob, tables in res:
        assert outs == single                          # sharding + gather is bitwise invisible, on every rank
        assert blob == b"mel-basis-\x00\x01\x02" * 1000
        assert tables == b"tables-built-by-rank-0"     # every rank created its engine from rank 0's blob


def test_rank_binds_to_the_cpus_of_its_gpus_numa_node(tmp_path):
    """bind_to_gpu_numa_node on a fake sysfs tree: two GPUs on two nodes, the visibility variables mapped through, the affinity mask
    intersected; missing information leaves the process unbound and says why."""
    from gomel_amd import dist as gd
    sysfs = tmp_path / "sys"
    for bdf, node in (("0000:05:00.0", 0), ("0000:85:00.0", 1), ("0000:c5:00.0", -1)):
        d = sysfs / "bus" / "pci" / "devices" / bdf
        d.mkdir(parents=True)
        (d / "numa_node").write_text("%d\n" % node)
    for node, cpus in ((0, "0-3,16-19"), (1, "4-7,20-23")):
        d = sysfs / "devices" / "system" / "node" / ("node%d" % node)
        d.mkdir(parents=True)
        (d / "cpulist").write_text(cpus + "\n")
    ids = ["0000:05:00.0", "0000:85:00.0", "0000:c5:00.0"]
    got = {}
    allowed = set(range(0, 24))
    kw = dict(sysfs=str(sysfs), bus_ids=ids, setaffinity=lambda c: got.__setitem__("cpus", set(c)), getaffinity=lambda: allowed)
    r = gd.bind_to_gpu_numa_node(1, env={}, **kw)
    assert r["bound"] and r["numa_node"] == 1 and got["cpus"] == {4, 5, 6, 7, 20, 21, 22, 23} and r["cpus"] == 8
    r = gd.bind_to_gpu_numa_node(0, env={"HIP_VISIBLE_DEVICES": "1,0"}, **kw)           # ordinal 0 is physical GPU 1
    assert r["bound"] and r["bdf"] == "0000:85:00.0"
    r = gd.bind_to_gpu_numa_node(0, env={"HIP_VISIBLE_DEVICES": "1", "ROCR_VISIBLE_DEVICES": "2,0"}, **kw)   # HIP index 1 of ROCR's list -> GPU 0
    assert r["bound"] and r["bdf"] == "0000:05:00.0" and got["cpus"] == {0, 1, 2, 3, 16, 17, 18, 19}
    # HIP_ and CUDA_VISIBLE_DEVICES are aliases to the HIP runtime: ONE filter (the first set wins), never composed -- both set to
    # the same permutation still means "ordinal 0 is physical GPU 1", not "GPU 0"
    r = gd.bind_to_gpu_numa_node(0, env={"HIP_VISIBLE_DEVICES": "1,0", "CUDA_VISIBLE_DEVICES": "1,0"}, **kw)
    assert r["bound"] and r["bdf"] == "0000:85:00.0"
    r = gd.bind_to_gpu_numa_node(1, env={"CUDA_VISIBLE_DEVICES": "1,0"}, **kw)            # CUDA_ alone is honoured
    assert r["bound"] and r["bdf"] == "0000:05:00.0"
    assert gd.visible_index(0, {"HIP_VISIBLE_DEVICES": "2", "CUDA_VISIBLE_DEVICES": "0,1"}) == 2     # HIP_ before CUDA_
    assert gd.visible_index_why(3, {"HIP_VISIBLE_DEVICES": "0,1"}) == (None, "HIP_VISIBLE_DEVICES='0,1' lists 2 device(s), ordinal 3 asked for")
    got.clear()
    allowed = {0, 1, 4}
    kw["getaffinity"] = lambda: allowed
    r = gd.bind_to_gpu_numa_node(1, env={}, **kw)                                        # cgroup / taskset mask respected
    assert r["bound"] and got["cpus"] == {4}
    for rank, env, why in ((2, {}, "numa_node = -1"), (5, {}, "no PCI bus id"), (0, {"ROCR_VISIBLE_DEVICES": "GPU-abcdef"}, "ROCR_VISIBLE_DEVICES='GPU-abcdef' is not a list of indices")):
        got.clear()
        r = gd.bind_to_gpu_numa_node(rank, env=env, **kw)
        assert not r["bound"] and why in r["why"] and not got
    allowed = {8, 9}
    r = gd.bind_to_gpu_numa_node(0, env={}, **kw)
    assert not r["bound"] and "outside" in r["why"]
    assert gd.gpu_bus_ids(smi="/nonexistent/rocm-smi") == []
    assert gd._parse_cpulist("0-2,7,9-10\n") == {0, 1, 2, 7, 9, 10}


def test_c_abi_numa_lookup_and_thread_binding_on_a_fake_sysfs(tmp_path):
    """gomel_numa_cpus / gomel_numa_bind_thread (host logic, gomel_host.cpp): what gomel_from_mel_batch_multi's per-device threads use
    -- the one-process multi-GPU path a Go host takes -- on a fake sysfs tree: node and CPU list of a PCI bus id (upper-case
    ids as hipDeviceGetPCIBusId may spell them), the calling THREAD bound to that node's CPUs intersected with its mask (never widened),
    GOMEL_ERR_EMPTY (4) where the machine does not say, the process's other threads untouched."""
    import ctypes
    import threading
    from gomel_amd import _lib
    L = _lib.lib()
    allowed = sorted(os.sched_getaffinity(0))
    sysfs = tmp_path / "sys"
    mine = allowed[: max(1, len(allowed) // 2)]                      # a node made of CPUs this process may really use
    lists = {0: ",".join(str(c) for c in mine), 1: "%d-%d" % (max(allowed) + 100, max(allowed) + 103)}
    for bdf, node in (("0000:05:00.0", 0), ("0000:85:00.0", 1), ("0000:c5:00.0", -1)):
        d = sysfs / "bus" / "pci" / "devices" / bdf
        d.mkdir(parents=True)
        (d / "numa_node").write_text("%d\n" % node)
    for node, cpus in lists.items():
        d = sysfs / "devices" / "system" / "node" / ("node%d" % node)
        d.mkdir(parents=True)
        (d / "cpulist").write_text(cpus + "\n")
    root = str(sysfs).encode()
    node, n = ctypes.c_int32(-9), ctypes.c_size_t(0)
    buf = (ctypes.c_int32 * 64)()
    assert L.gomel_numa_cpus(root, b"0000:05:00.0", ctypes.byref(node), buf, 64, ctypes.byref(n)) == 0
    assert node.value == 0 and list(buf[: n.value]) == mine
    assert L.gomel_numa_cpus(root, b"0000:85:00.0", ctypes.byref(node), buf, 2, ctypes.byref(n)) == 0 and n.value == 4 and node.value == 1   # count beyond cap
    assert list(buf[:2]) == [max(allowed) + 100, max(allowed) + 101]
    assert L.gomel_numa_cpus(root, b"0000:C5:00.0", ctypes.byref(node), buf, 64, ctypes.byref(n)) == 4 and node.value == -1                 # numa_node = -1
    assert L.gomel_numa_cpus(root, b"0000:ff:00.0", None, None, 0, None) == 4                                                                # no such device
    assert L.gomel_numa_cpus(root, None, None, None, 0, None) == 7
    res = {}

    def worker():
        nb, nd = ctypes.c_int32(0), ctypes.c_int32(-9)
        res["rc"] = L.gomel_numa_bind_thread(root, b"0000:05:00.0", ctypes.byref(nd), ctypes.byref(nb))
        res["node"], res["n"] = nd.value, nb.value
        res["mask"] = sorted(os.sched_getaffinity(threading.get_native_id()))
        res["rc_outside"] = L.gomel_numa_bind_thread(root, b"0000:85:00.0", None, ctypes.byref(nb))   # node 1's CPUs are not ours: left alone
        res["mask_after"] = sorted(os.sched_getaffinity(threading.get_native_id()))
    t = threading.Thread(target=worker)
    t.start()
    t.join()
    assert res["rc"] == 0 and res["node"] == 0 and res["n"] == len(mine) and res["mask"] == mine
    assert res["rc_outside"] == 4 and res["mask_after"] == mine
    assert sorted(os.sched_getaffinity(0)) == allowed                # the binding was the thread's, not the process's
