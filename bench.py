#!/usr/bin/env python3
"""bench.py -- Griffin-Lim frames/s (4096-pt, 60 iters) on MI355X, one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg3|cfg2|cfg4|cfg5] [--parity all|sample|none] [--no-cpu]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
`python bench.py --gpus N` with no launcher starts the N ranks itself (torch.distributed.run as a child process, before anything
touches a GPU) and exits with its code.

METRIC (SURVEY 8d / BASELINE.md): frames/s = frames pushed through all 60 Griffin-Lim iterations / wall time of
gomel_from_mel_batch -- host fp64 log-mel in, host fp64 waveforms out, H2D and D2H INCLUDED, handle creation excluded.  A "step" is
one such call over the rank's batch; W warm-up steps, then exactly K timed steps between barrier + synchronize on both sides, MAX
over ranks.  The caller's buffers are pinned host memory (gomel_host_alloc: what a bulk Go caller would hand over); the same call
through pageable buffers is reported beside it (`pageable_frames_per_s`; the buffers are bound to the engine once, Engine.bind_buffers, as
a C caller's pointer arrays are filled once), and the HBM-resident rate -- inputs already in HBM, output
left in HBM -- as `resident_frames_per_s`.  The roofline block is about the dominant kernel: its launch time comes from HIP events
on the library's stream in this run, its algorithmic bytes from SURVEY 8(d).

Workloads (BASELINE.json configs): cfg3 = configs[2], 256 x (192 mels x 400 frames) random log-mel, N 4096 / hop 1280, 60
iterations (the configuration the roofline figure is quoted on; default at N = 1, weak scaling when forced at N > 1).  cfg5 =
configs[4], a FIXED corpus of 8192 such utterances cut into one contiguous shard per rank (strong scaling; default when N > 1); a
rank holds a shard of up to 1024 utterances whole on the host (8 GPUs: 1024 per rank; a larger shard re-uses one 256-utterance host
block with the shard's own seeds), one call per block of 256.  cfg2 = configs[1], one 10 s utterance (342 frames), also reported in every N = 1 line as `single_utterance`.  cfg4 =
configs[3], the 22.05 kHz TTS geometry (N 1024 / hop 256 / 80 mels).  Utterances are independent: no data-path collective; the
one collective is the broadcast of the table blob ("mel basis") from rank 0 over RCCL.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

WORKLOADS = {
    # name: (description, Mel fields, utterances per GPU (corpus size when strong), frames per utterance, sample rate)
    "cfg3": ("configs[2]: 256 x (192 mels x 400 frames) random log-mel, N=4096 hop=1280, 60 GL iters",
             dict(NumMels=192, MelFmin=0.0, MelFmax=16000.0, Window=1280, Resolut=4096, GriffinLimIterations=60), 256, 400, 44100),
    "cfg2": ("configs[1]: single 10 s utterance @44.1kHz (342 frames), N=4096 hop=1280, 60 GL iters",
             dict(NumMels=192, MelFmin=0.0, MelFmax=16000.0, Window=1280, Resolut=4096, GriffinLimIterations=60), 1, 342, 44100),
    "cfg4": ("configs[3]: 256 x (80 mels x 400 frames), 22.05kHz TTS geometry N=1024 hop=256, 60 GL iters",
             dict(NumMels=80, MelFmin=0.0, MelFmax=8000.0, Window=256, Resolut=1024, GriffinLimIterations=60), 256, 400, 22050),
    "cfg5": ("configs[4]: 8192-utterance corpus (192 mels x 400 frames each) sharded over the ranks, N=4096 hop=1280, 60 GL iters",
             dict(NumMels=192, MelFmin=0.0, MelFmax=16000.0, Window=1280, Resolut=4096, GriffinLimIterations=60), 8192, 400, 44100),
}
STRONG = {"cfg5"}
BLOCK = 256                     # utterances per gomel_from_mel_batch call
MATERIALISE = 1024              # shards up to this many utterances are materialised whole on the host (configs[4] at 8 GPUs: 1024 per rank)
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PARITY_SCHEDULE = 30            # iterations in fp64 before the fp32 kernel takes over (gomel_set_precision_schedule), see DESIGN.md section 2
TRAFFIC_JSON = os.path.join(ROOT, "profiles", "traffic.json")


def algorithmic_bytes_per_frame_iter(N, hop):
    """read hop new samples + (N/2+1) target magnitudes, write hop samples, fp32."""
    return hop * 4 + (N // 2 + 1) * 4 + hop * 4


def make_inputs(fields, n_utt, frames, first):
    """Config-3 style synthetic log-mel: iid U[ln 1e-5, ln 50], seed = global utterance index."""
    K = fields["NumMels"]
    return [np.random.default_rng(first + u).uniform(math.log(1e-5), math.log(50.0), size=(frames * K, 2)) for u in range(n_utt)]


def build_engine(fields, device, tables=None):
    import gomel_amd
    m = gomel_amd.NewMel()
    for k, v in fields.items():
        setattr(m, k, v)
    m.Device = device
    return gomel_amd.Engine(m._config(), tables=tables)


PATHS = ("generic", "fused", "persistent", "fused")      # gomel_timing.fast_path (3 = the fused kernel's any-hop form)


def committed_traffic(workload, fast_path):
    """HBM bytes per launch of the dominant kernel from the COMMITTED rocprofv3 --pmc passes (profiles/traffic.json): counters
    cannot be read from inside the timed process, so this figure is a label from an earlier profile of the same kernel."""
    try:
        with open(TRAFFIC_JSON) as f:
            e = json.load(f).get("%s:%s" % (workload, PATHS[fast_path]))
        return (None, None) if e is None else (e.get("hbm_bytes_per_launch"), "profiles/traffic.json [%s] (separate rocprofv3 --pmc passes, not this run)" % e.get("from", "?"))
    except (OSError, ValueError):
        return None, None


def host_cpus():
    """Threads the CPU legs may use: the affinity mask, cut to the cgroup's CPU quota when there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    how = "affinity mask"
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max" and int(q) // int(p) >= 1 and int(q) // int(p) < n:
            n, how = int(q) // int(p), "cgroup cpu.max"
    except (OSError, ValueError):
        pass
    return max(1, n), how


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def go_probe():
    """say whether the box could time the Go reference at all."""
    go = shutil.which("go")
    if not go:
        return "go: not found on PATH (the Go reference cannot be built or timed here; CPU legs are the C / NumPy restatements)"
    try:
        v = subprocess.run([go, "version"], capture_output=True, text=True, timeout=20).stdout.strip()
    except Exception as ex:  # noqa: BLE001
        v = "go present but `go version` failed: %r" % (ex,)
    return v + " (module cache for go-dsp / gossp still needed; not attempted)"


# ------------------------------------------------------------------------------------------------ the measured legs
def host_to_host(engine, fields, n_local, frames, first, steps, warmup, barrier, pinned=True, io32=False):
    """The metric: K calls of gomel_from_mel_batch per BLOCK of the rank's shard, host buffers in, host buffers out.
    (io32: the float32 host-I/O entry point, gomel_from_mel_batch_f32 -- an informational side figure, never `value`.)"""
    import gomel_amd
    import torch
    K = fields["NumMels"]
    nb = min(BLOCK, n_local)
    # a shard of up to MATERIALISE utterances is held whole (every utterance its own data and buffers, 5.4 MB each at the towav
    # geometry); a larger one re-uses one block of BLOCK utterances' buffers with the shard's own seeds
    whole = n_local <= MATERIALISE
    nmat = n_local if whole else nb
    dt = np.float32 if io32 else np.float64
    alloc = (lambda shape: gomel_amd.host_alloc(shape, dtype=dt)) if pinned else (lambda shape: np.zeros(shape, dtype=dt))
    call = engine.from_mel_batch_f32 if io32 else engine.from_mel_batch
    mels = []
    for u, src in enumerate(make_inputs(fields, nmat, frames, first)):
        a = alloc((frames * K, 2))
        a[:] = src
        mels.append(a)
    L = engine.out_len(frames * K)
    outs = [alloc(L) for _ in range(nmat)]
    for o in outs:
        o[:] = 0.0                                                   # touched: what a caller's own slices are
    blocks = [(b0, min(nb, n_local - b0)) for b0 in range(0, n_local, nb)]

    # the caller's buffers are bound once (Engine.bind_buffers: the pointer arrays a C / Go caller fills at start-up), so that a timed
    # call is the C ABI call and the seeds -- not 0.5 ms of Python list handling per 256 utterances
    bound = {}
    if not io32:
        for b0, cnt in blocks:
            m0 = b0 if whole else 0
            bound[(m0, cnt)] = engine.bind_buffers(mels[m0:m0 + cnt], outs[m0:m0 + cnt])

    def step():
        for b0, cnt in blocks:
            seeds = [1000 + first + b0 + u for u in range(cnt)]      # the seed of an utterance is its GLOBAL index
            m0 = b0 if whole else 0
            if io32:
                call(mels[m0:m0 + cnt], None, seeds, outs[m0:m0 + cnt])
            else:
                engine.from_mel_batch_bound(bound[(m0, cnt)], seeds)

    for _ in range(warmup):
        step()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    last_b0, last_cnt = blocks[-1]
    m0 = last_b0 if whole else 0
    return elapsed, mels[m0:m0 + last_cnt], [1000 + first + last_b0 + u for u in range(last_cnt)], [np.array(o) for o in outs[m0:m0 + last_cnt]]


def resident(engine, fields, n_utt, frames, first, steps):
    """Inputs already in HBM, output left in HBM: the device-side rate, and the HIP-event timing of the dominant kernel."""
    import gomel_amd
    K = fields["NumMels"]
    mels = make_inputs(fields, n_utt, frames, first)
    batch = gomel_amd.Batch(engine, [frames * K] * n_utt)
    batch.upload(mels, None, [1000 + first + u for u in range(n_utt)])
    del mels
    batch.run()
    batch.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        batch.run()
    batch.sync()
    dt = time.perf_counter() - t0
    # HIP events on the library's stream around every launch of the dominant kernel, three calls back to back right behind the
    # loop above: the mean over all their launches is the figure (the chip is power-limited under this kernel, and the first
    # call after an idle gap runs through a clock transient -- 600 -> 830 -> 590 us per launch over 60 launches in
    # profiles/r02's kernel trace -- that the following calls do not show)
    tms = [batch.run_timed() for _ in range(3)]
    tm = dict(tms[-1])
    for k in ("dom_ms", "dom64_ms", "iter_launches", "iter64_launches"):
        tm[k] = sum(t[k] for t in tms)
    tm["per_call_launch_ms"] = [t["dom_ms"] / max(1, t["iter_launches"]) for t in tms]
    tm["dom_ms_last_call"] = tms[-1]["dom_ms"]
    batch.close()
    return n_utt * frames * steps / dt, tm


def config1(local_rank):
    """BASELINE configs[0]: towav geometry (cmd/towav/main.go:30-40, 2 iterations) on a PNG.  The PNG named there (the
    reference's README banner, 183x80) cannot run with NumMels = 192: the reference panics, the C ABI returns a status.  The
    same pipeline then runs on the repository's 192-row fixture, on the GPU and through the C restatement on one host core."""
    import gomel_amd
    from gomel_amd import codec, _lib
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import gomel_oracle as go
    import oracle_c
    gold = os.path.join(ROOT, "tests", "golden")
    out = {}
    m = gomel_amd.NewMel()
    m.MelFmin, m.MelFmax, m.YReverse, m.Window, m.NumMels, m.Resolut, m.GriffinLimIterations = 0, 16000, True, 1280, 192, 4096, 2
    m.Device = local_rank
    banner = os.path.join(gold, "reference_banner_183x80.png")
    if os.path.exists(banner):
        buf, _, _ = codec.loadpng(banner, True)
        try:
            m.FromMel(buf, seed=1)
            out["reference_banner_png"] = "ran (unexpected)"
        except gomel_amd.GomelError as e:
            out["reference_banner_png"] = "183x80 pixels = %d pairs, not a multiple of 192: %s (the reference panics, mel/impl.go:366)" % (
                len(buf), _lib.lib().gomel_strerror(e.status).decode())
    buf, samples, sr = codec.loadpng(os.path.join(gold, "chirp_192.png"), True)
    cfg = go.towav_config()
    x0 = go.x0_splitmix(1, go.out_len(cfg, len(buf)))
    m.FromMel(buf.copy(), x0=x0)
    t = time.perf_counter()
    for _ in range(20):
        got = m.FromMel(buf.copy(), x0=x0)
    gpu_ms = (time.perf_counter() - t) / 20 * 1e3
    co = oracle_c.load()
    t = time.perf_counter()
    want, _ = co.from_mel(cfg, buf, x0)
    cpu_ms = (time.perf_counter() - t) * 1e3
    frames = len(buf) // 192
    out.update({"fixture": "tests/golden/chirp_192.png (%d frames)" % frames, "gpu_ms_host_to_host": gpu_ms, "cpu_restatement_ms_1_thread": cpu_ms,
                "rms_vs_cpu_restatement": float(np.sqrt(np.mean((got - want) ** 2)))})
    return out


def cpu_legs(fields, frames, mels, seeds, outs_by_mode, mode):
    """ONE run of the C restatement of the reference algorithm (oracle/gomel_oracle.c) serves two purposes, outside every timed
    region: it is the reported CPU baseline (utterance-parallel host threads, the very utterances the GPU just computed) and it
    is the checker of those GPU results (`parity`), for every precision mode that was timed (`outs_by_mode`: name -> waveforms
    of the same utterances).  `mode`: all | sample | none."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import gomel_oracle as go
    import oracle_c
    co = oracle_c.load()
    cores, how = host_cpus()
    cfg = go.MelConfig(**fields)
    small = go.random_logmel(4, cfg.NumMels, seed=0)
    t = time.perf_counter()
    co.from_mel(cfg, small, go.x0_splitmix(0, go.out_len(cfg, len(small))))
    per_frame = (time.perf_counter() - t) / 4.0                      # one thread, seconds per frame (all iterations)
    est_all = len(mels) * frames * per_frame / cores
    if mode == "all" and est_all > 150.0:
        mode = "sample"                                              # a small host: keep the default run within minutes
    pick = list(range(len(mels)))
    if mode == "sample":
        want_n = int(max(cores, min(len(mels), 25.0 * cores / max(frames * per_frame, 1e-9))))   # ~25 s of CPU work on all cores
        pick = sorted(set(np.linspace(0, len(mels) - 1, min(len(mels), want_n)).astype(int).tolist()))
    L = go.out_len(cfg, len(mels[0]))
    t = time.perf_counter()
    want, used = co.from_mel_batch(cfg, [np.asarray(mels[u]) for u in pick], [go.x0_splitmix(seeds[u], L) for u in pick], threads=cores)
    dt = time.perf_counter() - t
    sig = float(np.mean([np.sqrt(np.mean(w * w)) for w in want]))
    cpu = {"value": len(pick) * frames / dt, "unit": "frames/s", "cores": int(used), "kind": "port",
           "sample": "%d of the %d utterances of the timed batch x %d frames x %d iters, C restatement of mel.go:76-152, %d threads (%s), %.1f s" % (
               len(pick), len(mels), frames, cfg.GriffinLimIterations, int(used), how, dt),
           "single_thread_frames_per_s": 1.0 / per_frame, "cpu_model": cpu_model(), "go_probe": go_probe()}
    # "CPU-strong" (informational): the GPU's formulation on the CPU -- real FFT + X/|X| -- NumPy / pocketfft, 1 process
    M = go.target_magnitudes(cfg, np.exp(go.random_logmel(16, cfg.NumMels, seed=1)))
    x0 = go.x0_splitmix(1, cfg.Resolut + 15 * cfg.Window)
    t = time.perf_counter()
    go.griffin_lim_rfft(go.hann(cfg.Resolut), cfg.Window, M, cfg.GriffinLimIterations, x0)
    cpu["cpu_strong_numpy_rfft_1_process_frames_per_s"] = 16.0 / (time.perf_counter() - t)
    parity = {}
    for name, outs in outs_by_mode.items():
        err = np.array([float(np.sqrt(np.mean((outs[u] - w) ** 2))) for u, w in zip(pick, want)])
        parity[name] = {"checked_utterances": len(pick), "of": len(mels), "tolerance": 1e-4, "max": float(err.max()), "p99": float(np.percentile(err, 99)),
                        "p90": float(np.percentile(err, 90)), "median": float(np.median(err)), "n_over_1e-4": int((err > 1e-4).sum()),
                        "worst_utterances": [int(pick[i]) for i in np.argsort(-err)[:3]], "signal_rms": sig,
                        "oracle": "oracle/gomel_oracle.c (fp64 restatement, parity unpinned: no Go toolchain / reference vectors)"}
    if "fp32" in parity:
        parity["fp32"]["note"] = ("fp32 kernels vs fp64 reference arithmetic on iid-random (inconsistent) spectrograms: see DESIGN.md 'Parity in numbers'; "
                                  "the parity modes (64-bit mode, precision schedule) meet the tolerance on every utterance: `precise` block of the N = 1 line")
    return cpu, parity


def self_launch(args):
    """`bench.py --gpus N` without a launcher: start the N ranks as a child (torch.distributed.run) BEFORE anything touches the GPU."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS), help="default: cfg3 on one GPU, cfg5 (sharded corpus) on several")
    ap.add_argument("--utts-per-gpu", type=int, default=0)
    ap.add_argument("--parity", default=None, choices=["all", "sample", "none"], help="oracle check of the timed batch: a ~25 s sample of its utterances (default), all of them, or none")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU leg (baseline + parity)")
    ap.add_argument("--no-single", action="store_true", help="skip the side measurements (configs[1], configs[0], pageable buffers)")
    ap.add_argument("--no-precise", action="store_true", help="skip the parity modes' legs (64-bit mode, precision schedule)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # One rank per GPU.  GOMEL_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks: collectives run
    # on CPU tensors and ranks wrap around the visible devices (the driver's multi-GPU runs use the default, nccl = RCCL).
    backend = os.environ.get("GOMEL_BENCH_BACKEND", "nccl")
    local_rank = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(local_rank)
    coll_dev = torch.device("cuda", local_rank) if backend == "nccl" else torch.device("cpu")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=coll_dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    def barrier():
        if world > 1:
            dist.barrier()

    wl = args.workload or ("cfg3" if world == 1 else "cfg5")
    desc, fields, n_utt, frames, sr = WORKLOADS[wl]
    strong = wl in STRONG
    if strong:
        first = rank * (n_utt // world)   # contiguous shard of the fixed corpus
        n_utt = n_utt // world
    else:
        first = rank * n_utt
    if args.utts_per_gpu > 0:
        n_utt = args.utts_per_gpu
        first = rank * n_utt

    # table blob ("mel basis"): rank 0 builds, broadcast over RCCL/xGMI, every rank creates its handle from it
    from gomel_amd import dist as gdist
    t0 = time.perf_counter()
    engine = gdist.create_engine_broadcast(lambda tables=None: build_engine(fields, local_rank, tables), rank, world, device=coll_dev)
    bcast_s = time.perf_counter() - t0

    elapsed_local, mels_last, seeds_last, outs_last = host_to_host(engine, fields, n_utt, frames, first, args.steps, args.warmup, barrier)
    elapsed = elapsed_local
    per_rank = [n_utt * frames * args.steps / elapsed_local]
    if world > 1:
        t = torch.tensor([elapsed_local], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        gathered = [None] * world
        dist.all_gather_object(gathered, per_rank[0])
        per_rank = gathered

    # device-side figures (outside the timed region): one BLOCK resident in HBM
    res_fps, tm = resident(engine, fields, min(n_utt, BLOCK), frames, first, max(3, min(args.steps, 10)))
    # inside the timed region the library cuts a call into chunks (from_mel_batch_guarded in gomel_hip.hip: m = frames / 12288 full
    # chunks' worth at Resolut 4096, / 32768 below, cut half : full x (m-1) : half): the same kernel at a full chunk's size
    nb = min(n_utt, BLOCK)
    full = 12288 if fields["Resolut"] >= 4096 else 32768
    m_full = min(16, max(1, nb * frames // full))
    n_chunks = 1 if nb * frames < full else min(m_full + 1, nb)
    tm_chunk = resident(engine, fields, max(1, round(nb / m_full)), frames, first, 3)[1] if n_chunks > 1 else tm

    # Everything below is measured BESIDE the headline (other configurations, other host buffers, the parity modes, the CPU leg): a
    # failure in one of these legs is reported in the line (`leg_errors`) instead of costing the line itself.
    leg_errors = {}

    def guarded(name, fn):
        try:
            return fn()
        except Exception as ex:  # noqa: BLE001 -- reported, not hidden: the JSON line carries it
            leg_errors[name] = repr(ex)
            return None

    side = rank == 0 and world == 1 and not args.no_single           # side measurements: N = 1 only
    single = pageable = cfg1 = f32io = None

    def side_legs():
        nonlocal single, pageable, cfg1
        if wl != "cfg2":
            d2, f2, u2, fr2, sr2 = WORKLOADS["cfg2"]
            e2 = build_engine(f2, local_rank)
            fps2, tm2 = resident(e2, f2, u2, fr2, 0, 20)
            el2, _, _, _ = host_to_host(e2, f2, u2, fr2, 0, 20, 2, lambda: None)
            single = {"workload": d2, "resident_frames_per_s": fps2, "resident_ms_per_utterance": u2 * fr2 / fps2 * 1e3,
                      "host_to_host_ms_per_utterance": el2 / 20 * 1e3, "real_time_factor_host_to_host": u2 * fr2 * 20 / el2 * f2["Window"] / sr2,
                      "kernel_path": PATHS[tm2["fast_path"]]}
            e2.close()
        el_p, _, _, _ = host_to_host(engine, fields, min(n_utt, BLOCK), frames, first, 3, 1, lambda: None, pinned=False)
        pageable = min(n_utt, BLOCK) * frames * 3 / el_p
        cfg1 = config1(local_rank)

    if side:
        guarded("side_measurements", side_legs)
    if rank == 0 and world == 1:                                      # float32 host I/O (half the PCIe bytes): informational, any workload
        def f32_leg():
            el_f, _, _, _ = host_to_host(engine, fields, min(n_utt, BLOCK), frames, first, 3, 1, lambda: None, io32=True)
            return min(n_utt, BLOCK) * frames * 3 / el_f
        f32io = guarded("float32_host_io", f32_leg)
    # The PARITY modes on the same clock (N = 1): the very batch of the fp32 leg through 64-bit mode (fp64 kernels throughout: the
    # mode that agrees with the reference's float64 arithmetic on every input) and through the precision schedule (the first
    # PARITY_SCHEDULE iterations fp64, the rest fp32) -- host to host like `value`, resident, their kernels' launch times.
    outs_by_mode = {"fp32": outs_last}
    precise = None

    def precise_legs():
        nonlocal precise
        precise = {}
        N, hop = fields["Resolut"], fields["Window"]
        b32, b64 = algorithmic_bytes_per_frame_iter(N, hop), 2 * algorithmic_bytes_per_frame_iter(N, hop)
        for name, sched in (("fp64", -1), ("schedule", PARITY_SCHEDULE)):
            engine.set_precision_schedule(sched)
            psteps = max(3, args.steps // 4)
            el, _, _, outs_p = host_to_host(engine, fields, n_utt, frames, first, psteps, 1, lambda: None)
            rfps, tmp = resident(engine, fields, min(n_utt, BLOCK), frames, first, 3)
            outs_by_mode[name] = outs_p
            n64, n32 = tmp["iter64_launches"], tmp["iter_launches"] - tmp["iter64_launches"]
            t64 = tmp["dom64_ms"] / max(1, n64) * 1e-3
            t32 = (tmp["dom_ms"] - tmp["dom64_ms"]) / max(1, n32) * 1e-3
            precise[name] = {
                "what": "gomel_set_precision(h, 64): k_gl_fused_d for all %d iterations" % fields["GriffinLimIterations"] if sched < 0 else
                        "gomel_set_precision_schedule(h, %d): k_gl_fused_d for the first %d iterations, k_gl_fused (fp32) for the other %d" % (sched, sched, fields["GriffinLimIterations"] - sched),
                "value": n_utt * frames * psteps / el, "unit": "frames/s", "steps": psteps, "ms_per_step": el / psteps * 1e3,
                "timed_region": "the same host-to-host call as `value` (pinned fp64 buffers, H2D + all device work + D2H)",
                "resident_frames_per_s": rfps,
                "kernel_fp64": {"name": "k_gl_fused_d", "launches_per_call": n64 // 3, "launch_ms": t64 * 1e3,
                                "frac_in_survey_8d_bytes": tmp["frames"] * b32 / t64 / 1e9 / HBM_PEAK_GBS if n64 else None,
                                "frac_in_fp64_bytes": tmp["frames"] * b64 / t64 / 1e9 / HBM_PEAK_GBS if n64 else None},
                "kernel_fp32": {"name": "k_gl_fused", "launches_per_call": n32 // 3, "launch_ms": t32 * 1e3,
                                "frac": tmp["frames"] * b32 / t32 / 1e9 / HBM_PEAK_GBS} if n32 else None,
            }

    if rank == 0 and world == 1 and not args.no_precise:
        guarded("precise", precise_legs)
        guarded("precise_reset", lambda: engine.set_precision_schedule(0))
    cpu = parity = None
    if rank == 0 and not args.no_cpu:
        mode = args.parity or "sample"                                # bounded (~25 s); --parity all checks every utterance of the batch (~80 s)
        if mode != "none":
            got = guarded("cpu_baseline", lambda: cpu_legs(fields, frames, mels_last, seeds_last, outs_by_mode, mode))
            if got is not None:
                cpu, parity_all = got
                parity = parity_all.get("fp32")
                for name in (precise or {}):
                    if name in parity_all:
                        precise[name]["parity"] = parity_all[name]
        if world > 1:
            cpu = None                                                # the CPU baseline is an N = 1 figure

    if rank == 0:
        N, hop, I = fields["Resolut"], fields["Window"], fields["GriffinLimIterations"]
        total_frames = world * n_utt * frames
        value = total_frames * args.steps / elapsed
        launches = max(1, tm["iter_launches"])
        t_launch = tm["dom_ms"] / launches * 1e-3
        bytes_launch = tm["frames"] * algorithmic_bytes_per_frame_iter(N, hop) * (I if tm["fast_path"] == 2 else 1)   # the persistent kernel: one launch = all I iterations
        achieved = bytes_launch / t_launch / 1e9
        traffic, traffic_source = committed_traffic(wl if wl != "cfg5" else "cfg3", tm["fast_path"])
        per_frame_bytes = I * algorithmic_bytes_per_frame_iter(N, hop) + fields["NumMels"] * 16 + hop * 8
        line = {
            "metric": "Griffin-Lim frames/s (4096-pt, 60 iters)" if N == 4096 else "Griffin-Lim frames/s (%d-pt, %d iters)" % (N, I),
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": desc, "utterances_per_gpu": n_utt, "frames_per_utterance": frames, "resolut": N, "window_hop": hop,
                       "num_mels": fields["NumMels"], "iterations": I, "parallelism": "utterance-sharded x%d" % world,
                       "kernel_path": PATHS[tm["fast_path"]],
                       "timed_region": "gomel_from_mel_batch host-to-host (pinned fp64 buffers, H2D + all device work + D2H), %d utterances per call" % min(n_utt, BLOCK)},
            "real_time_factor": value * hop / sr,
            "resident_frames_per_s": res_fps,
            "pageable_frames_per_s": pageable,
            "float32_host_io_frames_per_s": f32io,   # gomel_from_mel_batch_f32: float log-mel in, float waveforms out (half the PCIe bytes); informational
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_source, "kernel": tm["dom_kernel"], "launch_ms": t_launch * 1e3,
                         "launch_ms_per_timed_call": tm.get("per_call_launch_ms"),
                         "algorithmic_bytes_per_launch": bytes_launch,
                         "launch": "one launch over the whole %d-utterance batch (the configuration as BASELINE names it), HIP events on the library's stream in this process" % min(n_utt, BLOCK),
                         "timed_region_launches": {"what": "the same kernel at the size of a full chunk of the host-to-host call (%d chunks per call, first and last half-size; copy/compute overlap)" % n_chunks,
                                                   "frames_per_launch": tm_chunk["frames"], "launch_ms": tm_chunk["dom_ms"] / max(1, tm_chunk["iter_launches"]),
                                                   "frac": tm_chunk["frames"] * algorithmic_bytes_per_frame_iter(N, hop) * (I if tm_chunk["fast_path"] == 2 else 1) /
                                                           (tm_chunk["dom_ms"] / max(1, tm_chunk["iter_launches"]) * 1e-3) / 1e9 / HBM_PEAK_GBS},
                         "whole_job_frac_resident": res_fps * per_frame_bytes / 1e9 / HBM_PEAK_GBS,
                         "whole_job_frac_host_to_host": value / world * per_frame_bytes / 1e9 / HBM_PEAK_GBS},
            "cpu_baseline": cpu,
            "parity": parity,
            "precise": precise,
            "leg_errors": leg_errors or None,
            "single_utterance": single,
            "config1_towav_png_2_iters": cfg1,
            "world_size": world, "backend": backend if world > 1 else None, "per_rank_frames_per_s": per_rank,
            "table_broadcast_s": bcast_s,
            "device_ms": {"total": tm["total_ms"], "iterations": tm["iter_ms"], "dominant_kernel": tm["dom_ms_last_call"]},
        }
        print(json.dumps(line), flush=True)
    engine.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
