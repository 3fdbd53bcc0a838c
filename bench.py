#!/usr/bin/env python3
"""bench.py -- Griffin-Lim frames/s (4096-pt, 60 iters) on MI355X, one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg3|cfg2|cfg4|cfg5] [--no-cpu]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch: log-mel -> target magnitudes, x0 generation, all 60 Griffin-Lim
iterations, float -> double output -- everything (*Mel).FromMel does (mel/mel.go:142-152) -- with the inputs (fp64
log-mel as the API receives it) already resident in HBM and the output left in HBM.  Utterances are independent, so with
N GPUs every rank runs its own shard of the same size (weak scaling) and there is no data-path collective; the only
collective is the one-off broadcast of the table blob ("mel basis") from rank 0 over RCCL.

Workloads (BASELINE.json configs): cfg3 = configs[2], 256 x (192 mels x 400 frames) random log-mel, N 4096 / hop 1280,
60 iterations (the configuration the HBM-roofline figure is quoted on; default).  cfg2 = configs[1], one 10 s utterance
(342 frames) -- launch-latency bound, reported in the same JSON line as "single_utterance".  cfg4 = configs[3], the
22.05 kHz TTS geometry (N 1024 / hop 256 / 80 mels).  cfg5 = configs[4], a fixed 8192-utterance corpus cut into one
contiguous shard per rank (strong scaling; ~105 GB of HBM at N=1).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

WORKLOADS = {
    # name: (description, Mel fields, utterances per GPU, frames per utterance, sample rate)
    "cfg3": ("configs[2]: 256 x (192 mels x 400 frames) random log-mel, N=4096 hop=1280, 60 GL iters",
             dict(NumMels=192, MelFmin=0.0, MelFmax=16000.0, Window=1280, Resolut=4096, GriffinLimIterations=60), 256, 400, 44100),
    "cfg2": ("configs[1]: single 10 s utterance @44.1kHz (342 frames), N=4096 hop=1280, 60 GL iters",
             dict(NumMels=192, MelFmin=0.0, MelFmax=16000.0, Window=1280, Resolut=4096, GriffinLimIterations=60), 1, 342, 44100),
    "cfg4": ("configs[3]: 256 x (80 mels x 400 frames), 22.05kHz TTS geometry N=1024 hop=256, 60 GL iters",
             dict(NumMels=80, MelFmin=0.0, MelFmax=8000.0, Window=256, Resolut=1024, GriffinLimIterations=60), 256, 400, 22050),
    # configs[4]: a FIXED corpus of 8192 utterances cut into contiguous shards, one per rank (strong scaling; 8192/N per GPU)
    "cfg5": ("configs[4]: 8192-utterance corpus (192 mels x 400 frames each) sharded over the ranks, N=4096 hop=1280, 60 GL iters",
             dict(NumMels=192, MelFmin=0.0, MelFmax=16000.0, Window=1280, Resolut=4096, GriffinLimIterations=60), 8192, 400, 44100),
}
STRONG = {"cfg5"}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes_per_frame_iter(N, hop):
    """read hop new samples + (N/2+1) target magnitudes, write hop samples, fp32."""
    return hop * 4 + (N // 2 + 1) * 4 + hop * 4


def make_inputs(fields, n_utt, frames, rank):
    """Config-3 style synthetic log-mel: iid U[ln 1e-5, ln 50], seed = global utterance index."""
    K = fields["NumMels"]
    mels = []
    for u in range(n_utt):
        rng = np.random.default_rng(rank * n_utt + u)
        mels.append(rng.uniform(math.log(1e-5), math.log(50.0), size=(frames * K, 2)))
    return mels


def build_engine(fields, device, tables=None):
    import gomel_amd
    m = gomel_amd.NewMel()
    for k, v in fields.items():
        setattr(m, k, v)
    m.Device = device
    cfg = m._config()
    return gomel_amd.Engine(cfg, tables=tables)


def run_workload(engine, fields, n_utt, frames, rank, steps, warmup, barrier):
    import gomel_amd
    import torch
    K = fields["NumMels"]
    mels = make_inputs(fields, n_utt, frames, rank)
    batch = gomel_amd.Batch(engine, [frames * K] * n_utt)
    seeds = [1000 + rank * n_utt + u for u in range(n_utt)]
    batch.upload(mels, None, seeds)     # inputs resident in HBM before the clock starts
    del mels
    for _ in range(warmup):
        batch.run()
    batch.sync()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        batch.run()
    batch.sync()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    tm = batch.run_timed()              # HIP events on the library's stream, outside the timed region
    out0 = batch.download()[0]
    batch.close()
    return elapsed, tm, out0


def host_buffer_rate(engine, fields, n_utt, frames, rank):
    """The drop-in entry point as Go would call it in bulk: host fp64 in, host fp64 out (plan + H2D + run + D2H through
    pageable buffers).  Reported next to the HBM-resident figure, never as `value`."""
    mels = make_inputs(fields, n_utt, frames, rank)
    seeds = [1000 + rank * n_utt + u for u in range(n_utt)]
    outs = [np.zeros(engine.out_len(len(m))) for m in mels]     # caller-owned, already touched (what a Go caller's slices are)
    engine.from_mel_batch(mels, None, seeds, outs)              # warm: first-use allocation and page registration costs
    t = time.perf_counter()
    engine.from_mel_batch(mels, None, seeds, outs)
    return n_utt * frames / (time.perf_counter() - t)


def config1(local_rank):
    """BASELINE configs[0]: towav geometry (cmd/towav/main.go:30-40, 2 iterations) on a PNG.  The PNG named there (the
    reference's README banner, 183x80) cannot run with NumMels = 192: the reference panics, the C ABI returns a status.  The
    same pipeline then runs on the repository's 192-row fixture, on the GPU and through the C restatement on one host core."""
    import ctypes
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


def cpu_baseline(fields, frames, max_seconds_hint=30.0):
    """The C restatement of the reference algorithm (oracle/gomel_oracle.c) on the host cores: one 400-frame utterance
    per thread, same geometry and iteration count as the GPU workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import gomel_oracle as go
    import oracle_c
    co = oracle_c.load()
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, 16))  # a one-GPU box gives this job a 16-CPU share whatever the affinity mask says
    cfg = go.MelConfig(**fields)
    # calibrate on a few frames so the sample stays in the 10-30 s range whatever the host is
    small = go.random_logmel(4, cfg.NumMels, seed=0)
    t = time.perf_counter()
    co.from_mel(cfg, small, go.x0_splitmix(0, go.out_len(cfg, len(small))))
    per_frame = (time.perf_counter() - t) / 4.0
    f_s = int(max(8, min(frames, (max_seconds_hint / 2.0) / max(per_frame, 1e-6))))
    reps = int(min(8, max(1, math.ceil(10.0 / max(f_s * per_frame, 1e-3)))))   # aim at 10-30 s of wall time on the sample
    mels = [go.random_logmel(f_s, cfg.NumMels, seed=u) for u in range(cores * reps)]
    x0s = [go.x0_splitmix(1000 + u, go.out_len(cfg, len(m))) for u, m in enumerate(mels)]
    t = time.perf_counter()
    _, used = co.from_mel_batch(cfg, mels, x0s, threads=cores)
    dt = time.perf_counter() - t
    return {"value": len(mels) * f_s / dt, "unit": "frames/s", "cores": int(used), "kind": "port",
            "sample": "%d utterances x %d frames x %d iters on %d threads, C restatement of mel.go:76-152, %.1f s" % (
                len(mels), f_s, cfg.GriffinLimIterations, cores, dt),
            "single_thread_frames_per_s": 1.0 / per_frame}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--utts-per-gpu", type=int, default=0)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-single", action="store_true", help="skip the single-utterance (configs[1]) side measurement")
    args = ap.parse_args()

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

    desc, fields, n_utt, frames, sr = WORKLOADS[args.workload]
    strong = args.workload in STRONG
    if strong:
        n_utt = n_utt // world            # contiguous shard of the fixed corpus
    if args.utts_per_gpu > 0:
        n_utt = args.utts_per_gpu

    # table blob ("mel basis"): rank 0 builds, broadcast over RCCL/xGMI, every rank creates its handle from it
    from gomel_amd import dist as gdist
    t0 = time.perf_counter()
    engine = gdist.create_engine_broadcast(lambda tables=None: build_engine(fields, local_rank, tables), rank, world, device=coll_dev)
    bcast_s = time.perf_counter() - t0

    elapsed, tm, out0 = run_workload(engine, fields, n_utt, frames, rank, args.steps, args.warmup, barrier)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    single = None
    if rank == 0 and world == 1 and not args.no_single and args.workload != "cfg2":
        d2, f2, u2, fr2, sr2 = WORKLOADS["cfg2"]
        e2 = build_engine(f2, local_rank)
        el2, tm2, _ = run_workload(e2, f2, u2, fr2, 0, max(3, args.steps), 1, lambda: None)
        steps2 = max(3, args.steps)
        fps2 = u2 * fr2 * steps2 / el2
        single = {"workload": d2, "frames_per_s": fps2, "ms_per_utterance": el2 / steps2 * 1e3,
                  "real_time_factor": fps2 * f2["Window"] / sr2}
        e2.close()

    side = rank == 0 and world == 1 and not args.no_single       # side measurements and the CPU leg: N = 1 only
    host_rate = host_buffer_rate(engine, fields, n_utt, frames, rank) if side else None
    cfg1 = config1(local_rank) if side else None
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        cpu = cpu_baseline(fields, frames)

    if rank == 0:
        N, hop, I = fields["Resolut"], fields["Window"], fields["GriffinLimIterations"]
        total_frames = world * n_utt * frames
        value = total_frames * args.steps / elapsed
        launches = max(1, tm["iter_launches"])
        t_launch = tm["dom_ms"] / launches * 1e-3
        bytes_launch = tm["frames"] * algorithmic_bytes_per_frame_iter(N, hop)
        achieved = bytes_launch / t_launch / 1e9
        traffic = gdist.committed_traffic(args.workload, tm["fast_path"])
        line = {
            "metric": "Griffin-Lim frames/s (4096-pt, 60 iters)" if N == 4096 else "Griffin-Lim frames/s (%d-pt, %d iters)" % (N, I),
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": desc, "utterances_per_gpu": n_utt, "frames_per_utterance": frames, "resolut": N, "window_hop": hop,
                       "num_mels": fields["NumMels"], "iterations": I, "parallelism": "utterance-sharded x%d" % world,
                       "kernel_path": "fused" if tm["fast_path"] else "generic"},
            "real_time_factor": value * hop / sr,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": tm["dom_kernel"], "launch_ms": t_launch * 1e3,
                         "algorithmic_bytes_per_launch": bytes_launch,
                         "whole_job_frac": value / world * (I * algorithmic_bytes_per_frame_iter(N, hop) + fields["NumMels"] * 16 + hop * 8) / 1e9 / HBM_PEAK_GBS},
            "cpu_baseline": cpu,
            "single_utterance": single,
            "host_buffers_frames_per_s": host_rate,
            "config1_towav_png_2_iters": cfg1,
            "table_broadcast_s": bcast_s,
            "device_ms": {"total": tm["total_ms"], "iterations": tm["iter_ms"], "dominant_kernel": tm["dom_ms"]},
            "output_rms": float(np.sqrt(np.mean(out0 * out0))),
        }
        print(json.dumps(line), flush=True)
    engine.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
