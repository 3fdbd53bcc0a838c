"""Multi-GPU plumbing for the one place the path shards: independent utterances (mel/mel.go:142-152 keeps no state
between FromMel calls).  One process per GPU; contiguous utterance shards balanced by frame count; NO data-path
collective.  The only collective is the one-off broadcast of the table blob ("mel basis": Hann window + undomel/domel
tables) from rank 0 -- over RCCL/xGMI with the "nccl" backend, over gloo in the CPU tests.
"""
import json
import os
import re
import subprocess

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


# ------------------------------------------------------------------------------------------------ rank -> NUMA node
def _parse_cpulist(txt):
    cpus = set()
    for part in txt.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_bus_ids(smi="rocm-smi", timeout=30):
    """PCI bus ids of the GPUs in the order rocm-smi lists them (`rocm-smi --showbus --json`, a CHILD process: nothing in this
    process touches the GPU).  [] when the tool is missing or says nothing usable."""
    try:
        out = subprocess.run([smi, "--showbus", "--json"], capture_output=True, text=True, timeout=timeout).stdout
        data = json.loads(out[out.index("{"):])
    except (OSError, ValueError, subprocess.SubprocessError):
        return []
    cards = sorted((k for k in data if re.fullmatch(r"card\d+", k)), key=lambda k: int(k[4:]))
    ids = []
    for k in cards:
        bdf = next((str(v) for kk, v in data[k].items() if "bus" in kk.lower()), None)
        if bdf is None:
            return []
        ids.append(bdf.strip().lower())
    return ids


def visible_index_why(local_rank, env=None):
    """(index, why): the physical GPU index behind this rank's device ordinal, or (None, reason) when it cannot be told.
    HIP_VISIBLE_DEVICES and CUDA_VISIBLE_DEVICES are ALIASES to the HIP runtime (one filter, not two: the first of them that is set
    wins, HIP_ before CUDA_); ROCR_VISIBLE_DEVICES filters one layer below (the HIP ordinals index ROCR's list).  Only plain index
    lists -- the launcher's usual form -- are mapped; UUIDs or a list shorter than the ordinal are not guessed at."""
    env = os.environ if env is None else env
    idx = local_rank
    hip_level = next((var for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES") if env.get(var, "").strip()), None)
    for var in (hip_level, "ROCR_VISIBLE_DEVICES"):       # applied innermost first
        v = env.get(var, "").strip() if var else ""
        if not v:
            continue
        parts = [p.strip() for p in v.split(",")]
        if not all(p.isdigit() for p in parts):
            return None, "%s=%r is not a list of indices" % (var, v)
        if idx >= len(parts):
            return None, "%s=%r lists %d device(s), ordinal %d asked for" % (var, v, len(parts), idx)
        idx = int(parts[idx])
    return idx, None


def visible_index(local_rank, env=None):
    """The physical GPU index behind this rank's device ordinal (None when it cannot be told: visible_index_why says why)."""
    return visible_index_why(local_rank, env)[0]


def bind_to_gpu_numa_node(local_rank, sysfs="/sys", bus_ids=None, env=None, setaffinity=None, getaffinity=None):
    """Before any GPU call: restrict this process to the CPUs of the NUMA node its GPU hangs off, so that pinned host buffers
    (gomel_host_alloc) are first-touched in memory local to the GPU's root port and the rank's copy threads run there -- every rank
    moves tens of GB/s of pinned host traffic (PCIe, not xGMI, is the 8-GPU limiter).  The node comes from
    <sysfs>/bus/pci/devices/<bdf>/numa_node, its CPUs from <sysfs>/devices/system/node/node<N>/cpulist, intersected with the
    current affinity mask.  Returns a dict saying what was done ({"bound": False, "why": ...} when anything is missing: no NUMA
    information is not an error).  sysfs / bus_ids / env / the affinity calls are injectable for the CPU tests."""
    setaffinity = setaffinity or (lambda cpus: os.sched_setaffinity(0, cpus))
    getaffinity = getaffinity or (lambda: os.sched_getaffinity(0))
    idx, why = visible_index_why(local_rank, env)
    if idx is None:
        return {"bound": False, "why": "device visibility: " + why}
    ids = gpu_bus_ids() if bus_ids is None else list(bus_ids)
    if idx >= len(ids):
        return {"bound": False, "why": "no PCI bus id for GPU %d (%d listed)" % (idx, len(ids))}
    bdf = ids[idx]
    try:
        node = int(open(os.path.join(sysfs, "bus", "pci", "devices", bdf, "numa_node")).read().strip())
    except (OSError, ValueError):
        return {"bound": False, "why": "no numa_node for %s" % bdf, "bdf": bdf}
    if node < 0:
        return {"bound": False, "why": "numa_node = -1 (single-node machine or not reported)", "bdf": bdf}
    try:
        cpus = _parse_cpulist(open(os.path.join(sysfs, "devices", "system", "node", "node%d" % node, "cpulist")).read())
    except (OSError, ValueError):
        return {"bound": False, "why": "no cpulist for node %d" % node, "bdf": bdf, "numa_node": node}
    have = set(getaffinity())
    use = cpus & have
    if not use:
        return {"bound": False, "why": "node %d's CPUs are outside this process's affinity mask" % node, "bdf": bdf, "numa_node": node}
    setaffinity(use)
    return {"bound": True, "bdf": bdf, "numa_node": node, "cpus": len(use), "of_allowed": len(have)}
