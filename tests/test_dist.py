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
    import torch.multiprocessing as mp
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
