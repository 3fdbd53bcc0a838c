"""Multi-GPU plumbing for the one place the path shards: independent utterances (mel/mel.go:142-152 keeps no state
between FromMel calls).  One process per GPU; contiguous utterance shards balanced by frame count; NO data-path
collective.  The only collective is the one-off broadcast of the table blob ("mel basis": Hann window + undomel/domel
tables) from rank 0 -- over RCCL/xGMI with the "nccl" backend, over gloo in the CPU tests.
"""
import numpy as np


def shard_bounds(frame_counts, world):
    """Contiguous blocks of utterances, balanced by frame count (greedy on the prefix sum).  Returns world+1 bounds.
    Every utterance lands in exactly one block; blocks may be empty when there are fewer utterances than ranks."""
    n = len(frame_counts)
    pre = np.concatenate([[0], np.cumsum(np.asarray(frame_counts, dtype=np.int64))])
    total = int(pre[-1])
    bounds = [0]
    for r in range(1, world):
        target = total * r / world
        i = int(np.searchsorted(pre, target, side="left"))
        if i > 0 and abs(pre[i - 1] - target) <= abs(pre[min(i, n)] - target):
            i -= 1
        bounds.append(min(max(i, bounds[-1]), n))
    bounds.append(n)
    return bounds


def broadcast_bytes(payload, rank, world, device=None, src=0, force_collective=False):
    """Broadcast a bytes object from `src` (length first, then the body as a uint8 tensor on `device`).  A world of one returns the
    payload as it is, unless `force_collective` asks for the collective all the same (an initialised process group of one rank:
    how a one-GPU box exercises the RCCL call path -- communicator init, uint8 CUDA broadcast -- that N ranks take)."""
    if world == 1 and not force_collective:
        return payload
    import torch
    import torch.distributed as dist
    dev = device if device is not None else torch.device("cpu")
    n = torch.tensor([len(payload) if rank == src else 0], dtype=torch.int64, device=dev)
    dist.broadcast(n, src=src)
    if rank == src:
        body = torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(dev)
    else:
        body = torch.empty(int(n.item()), dtype=torch.uint8, device=dev)
    dist.broadcast(body, src=src)
    return bytes(body.cpu().numpy().tobytes())


def create_engine_broadcast(factory, rank, world, device=None, force_collective=False):
    """Rank 0 builds the handle (tables computed in fp64 on its host), exports the blob, broadcasts it; the other ranks
    create their handle from the blob (gomel_create_from_tables), so every GPU clamps to bit-identical tables."""
    if world == 1 and not force_collective:
        return factory()
    if rank == 0:
        eng = factory()
        broadcast_bytes(eng.export_tables(), rank, world, device, force_collective=force_collective)
        return eng
    blob = broadcast_bytes(None, rank, world, device, force_collective=force_collective)
    return factory(tables=blob)


def from_mel_sharded(compute, mels, x0s=None, seeds=None, rank=0, world=1, gather=True):
    """Shard `mels` over ranks, run `compute(mels, x0s, seeds) -> list of waveforms` on the local shard, and (optionally)
    all-gather the waveforms so every rank returns the full list in the original order.  `compute` is the engine's
    from_mel_batch on a GPU box; tests inject other callables."""
    n = len(mels)
    frames = [len(m) for m in mels]
    b = shard_bounds(frames, world)
    lo, hi = b[rank], b[rank + 1]
    if seeds is None:
        seeds = list(range(n))  # the seed of an utterance is its GLOBAL index, so sharding does not change results
    local = compute(mels[lo:hi], None if x0s is None else x0s[lo:hi], seeds[lo:hi]) if hi > lo else []
    if world == 1 or not gather:
        return local
    import torch.distributed as dist
    parts = [None] * world
    dist.all_gather_object(parts, local)
    return [w for p in parts for w in p]
