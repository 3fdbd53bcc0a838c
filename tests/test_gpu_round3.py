"""GPU tests added in round 3: a plain C caller that drives the GPU entry points in the cgo shim's exact order, the RCCL call path
on a process group of one rank, a per-GPU shard of BASELINE configs[4] on one GPU, and the parity mode (gomel_set_precision)
held to the north-star tolerance on EVERY utterance of configs[2] and configs[3]."""
import math
import os
import shutil
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("fp32_start_mode")]   # these tests target the fp32 kernels unless they select a mode (conftest.py)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RMS_TOL = 1e-4          # north_star: sample-for-sample within 1e-4 RMS

T = dict(NumMels=192, MelFmax=16000.0, Window=1280, Resolut=4096)     # towav geometry (cmd/towav/main.go:30-37)
S = dict(NumMels=80, MelFmax=8000.0, Window=256, Resolut=1024)        # BASELINE configs[3], the TTS geometry


def rms(a):
    return float(np.sqrt(np.mean(np.square(a))))


def mel_obj(fields, **kw):
    import gomel_amd
    m = gomel_amd.NewMel()
    for k, v in dict(fields, **kw).items():
        setattr(m, k, v)
    return m


def cfg3_inputs(U, F, K, first=0):
    mels = [np.random.default_rng(first + u).uniform(math.log(1e-5), math.log(50.0), size=(F * K, 2)) for u in range(U)]
    return mels, [1000 + first + u for u in range(U)]


# ------------------------------------------------------------------------------------------------ the cgo shim minus Go
def test_c_program_drives_the_gpu_entry_points_like_the_cgo_shim(oracle, tmp_path):
    """tests/c_gpu_caller.c = bindings/go/mel/mel_hip.go:175-205 (FromMel) and :264-317 (FromMelBatch) without Go: gomel_out_len,
    caller-drawn x0, gomel_from_mel, then gomel_from_mel_batch with C-allocated pointer arrays and out_cap = NULL, then the exp
    side effect.  Built as strict C99 with gcc; utterance 0 is the golden `towav_n4096_i60` case (mel/mel.go:142-152)."""
    from gomel_amd import _lib
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("gcc not available")
    exe = str(tmp_path / "c_gpu_caller")
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.run([gcc, "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c_gpu_caller.c"), "-o", exe, "-L", libdir, "-lgomel_hip", "-lm", "-Wl,-rpath," + libdir], check=True)
    gold = np.load(os.path.join(ROOT, "tests", "golden", "golden_v1.npz"))
    fields = dict(T, GriffinLimIterations=60)
    cfg = oracle.MelConfig(**fields)
    F, K, U = 4, 192, 3
    mels = [oracle.random_logmel(F, K, seed=7 + u) for u in range(U)]         # utterance 0: tests/golden/make_golden.py "towav_n4096"
    L = oracle.out_len(cfg, F * K)
    x0s = [oracle.x0_splitmix(8 + u, L) for u in range(U)]
    with open(tmp_path / "in.bin", "wb") as f:
        for m, x in zip(mels, x0s):
            f.write(np.ascontiguousarray(m, dtype=np.float64).tobytes())
            f.write(np.ascontiguousarray(x, dtype=np.float64).tobytes())
    r = subprocess.run([exe, "192", "16000", "1280", "4096", "60", str(F * K), str(U), str(tmp_path / "in.bin"), str(tmp_path / "out.bin")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "c gpu caller ok" in r.stdout, (r.returncode, r.stdout, r.stderr)
    out = np.fromfile(tmp_path / "out.bin", dtype=np.float64)
    assert out.size == L * (1 + U) + F * K * 2
    single, batch, slice0 = out[:L], out[L:L * (1 + U)].reshape(U, L), out[L * (1 + U):]
    want0 = gold["towav_n4096_i60_wave"]
    print("C caller: gomel_from_mel vs golden %.2e, gomel_from_mel_batch[0] vs golden %.2e" % (rms(single - want0), rms(batch[0] - want0)))
    assert rms(single - want0) < RMS_TOL and rms(batch[0] - want0) < RMS_TOL
    for u in range(1, U):
        assert rms(batch[u] - oracle.from_mel(cfg, mels[u].copy(), x0s[u])) < RMS_TOL
    assert np.allclose(slice0, np.exp(mels[0]).ravel(), rtol=4e-16, atol=0.0)  # the caller's slice ends exp'd (mel/mel.go:143); libm's exp vs NumPy's: last ulp


# ------------------------------------------------------------------------------------------------ RCCL on a group of one rank
def _nccl_world1_main(rep, port):
    """Child process (tests/_children.py has imported torch)."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    import gomel_amd
    import gomel_oracle as go
    from gomel_amd import dist as gd
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    rep.stage("init_process_group(nccl)")
    dist.init_process_group("nccl", rank=0, world_size=1, init_method="tcp://127.0.0.1:%d" % port, device_id=dev)   # bench.py's init
    try:
        m = gomel_amd.NewMel()
        for k, v in dict(S, GriffinLimIterations=4).items():
            setattr(m, k, v)
        m.Device = 0
        rep.stage("table broadcast over RCCL")
        eng = gd.create_engine_broadcast(lambda tables=None: gomel_amd.Engine(m._config(), tables=tables), 0, 1, device=dev, force_collective=True)
        blob = gd.broadcast_bytes(eng.export_tables(), 0, 1, device=dev, force_collective=True)   # what a rank > 0 receives
        eng2 = gomel_amd.Engine(m._config(), tables=blob)
        mels = [go.random_logmel(f, 80, seed=i) for i, f in enumerate([30, 7, 64])]
        rep.stage("FromMel")
        a = eng.from_mel_batch(mels, None, [0, 1, 2])
        b = eng2.from_mel_batch(mels, None, [0, 1, 2])
        rep.stage("all_reduce / all_gather_object / barrier")
        t = torch.tensor([1.25], dtype=torch.float64, device=dev)                                 # bench.py's MAX over ranks
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        gathered = [None]
        dist.all_gather_object(gathered, 3.5)
        dist.barrier()
        rep.result((blob == eng.export_tables(), all(np.array_equal(x, y) for x, y in zip(a, b)), float(t.item()), gathered, dist.get_backend()))
        eng.close(); eng2.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.torch_child
@pytest.mark.guard(700)
def test_table_broadcast_through_rccl_on_a_group_of_one_rank():
    """the one collective of the path is the broadcast of the table blob over RCCL.  One GPU cannot show N ranks, but it
    can run the SAME call path once: torch.distributed with the nccl backend (= RCCL) and device_id, a uint8 CUDA-tensor broadcast
    that is forced through the collective (dist.broadcast_bytes short-circuits a world of one otherwise), the float64 all_reduce(MAX)
    and all_gather_object bench.py uses -- and a handle built from the bytes that came back computes the same bits."""
    import _children
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    same_blob, same_bits, tmax, gathered, backend = _children.run_child(_nccl_world1_main, (port,))
    assert backend == "nccl" and same_blob and same_bits and tmax == 1.25 and gathered == [3.5]


# ------------------------------------------------------------------------------------------------ cmd/towav counterpart
def test_towav_command_line(tmp_path, capsys):
    """`python -m gomel_amd.towav <png> [rate]` = cmd/towav/main.go:10-47: fixed towav parameters, <png>.wav beside the input, the
    usage line and exit code 1 without an argument, the error line and exit code 1 for a file that does not load."""
    from gomel_amd import towav
    png = str(tmp_path / "in.png")
    shutil.copy(os.path.join(ROOT, "tests", "golden", "chirp_192.png"), png)
    assert towav.main(["towav", png, "22050"]) == 0
    raw = open(png + ".wav", "rb").read()
    assert raw[:4] == b"RIFF" and int.from_bytes(raw[24:28], "little") == 22050            # the rate given on the command line wins (main.go:39)
    assert (len(raw) - 44) // 2 == 4096 + 38 * 1280                                        # 39 frames, mono 16-bit
    assert towav.main(["towav"]) == 1 and "Usage" in capsys.readouterr().out
    assert towav.main(["towav", str(tmp_path / "missing.png")]) == 1
    assert "Error generating wave from spectrogram: wavNotLoaded" in capsys.readouterr().out


# ------------------------------------------------------------------------------------------------ precision schedule
@pytest.mark.parametrize("fields,frames", [(S, [60, 33, 47, 64, 12, 90]), (T, [40, 25, 64]), (dict(NumMels=160, MelFmax=8000.0, Window=256, Resolut=2048), [50, 70]),
                                           (dict(NumMels=80, MelFmax=8000.0, Window=300, Resolut=1024), [60, 33, 47]),      # any-hop kernels, even hop
                                           (dict(NumMels=80, MelFmax=8000.0, Window=275, Resolut=2048), [50, 70])])         # ... odd hop
def test_precision_schedule_is_fp64_then_fp32(oracle, fields, frames):
    """gomel_set_precision_schedule(h, k): the first k iterations in fp64 (k_gl_fused_d), the signal handed over once (tails folded
    in, rounded to fp32), the remaining iterations in fp32 (k_gl_fused).  That is exactly: a 64-bit call of k iterations, then an
    fp32 call of I - k iterations started from its output -- bit for bit -- on the fused kernels, on the generic pair and (same bits as
    the generic pair) on the persistent kernel; the degenerate schedules are the plain modes."""
    import gomel_amd
    I, k = 7, 3
    K = fields["NumMels"]
    mels = [oracle.random_logmel(f, K, seed=40 + i) for i, f in enumerate(frames)]
    seeds = list(range(len(frames)))

    def run(iters, sched, path, x0s=None):
        m = mel_obj(fields, GriffinLimIterations=iters)
        e = m.engine()
        e.set_path(path)
        e.set_precision_schedule(sched)
        b = gomel_amd.Batch(e, [len(mm) for mm in mels])
        b.upload(mels, x0s, seeds)
        tm = b.run_timed()
        out = b.download()
        b.close(); e.close()
        return out, tm

    mixed, tm = run(I, k, 2)
    assert tm["iter_launches"] == I and tm["iter64_launches"] == k and tm["fast_path"] in (1, 3), tm
    first, tm1 = run(k, -1, 2)                                  # k iterations, fp64 throughout
    assert tm1["iter64_launches"] == k
    second, tm2 = run(I - k, 0, 2, x0s=first)                   # I - k iterations, fp32 throughout, from there
    assert tm2["iter64_launches"] == 0
    assert all(np.array_equal(a, b) for a, b in zip(mixed, second))
    cfg = oracle.MelConfig(GriffinLimIterations=I, **fields)
    for i in (0, len(frames) - 1):
        want = oracle.from_mel(cfg, mels[i].copy(), oracle.x0_splitmix(seeds[i], len(mixed[i])))
        assert rms(mixed[i] - want) < 1e-5
    all64, tm64 = run(I, -1, 2)
    same, tms = run(I, I, 2)                                    # a schedule that covers every iteration is 64-bit mode
    assert tms["iter64_launches"] == I and all(np.array_equal(a, b) for a, b in zip(all64, same))
    gen64, _ = run(I, -1, 1)
    assert max(rms(a - b) for a, b in zip(gen64, all64)) < 1e-12
    # the generic pair runs the same schedule (round 4): fp64 launches for k iterations, the hand-over, fp32 launches for the rest --
    # again exactly the 64-bit call followed by the fp32 call; and the persistent kernel (where the geometry admits it), which sums
    # the frames in the generic kernels' order, gives the generic pair's bits
    gens, tmg = run(I, k, 1)
    assert tmg["iter64_launches"] == k and tmg["fast_path"] == 0
    g1, _ = run(k, -1, 1)
    g2, _ = run(I - k, 0, 1, x0s=g1)
    assert all(np.array_equal(a, b) for a, b in zip(gens, g2))
    try:
        pers, tmp = run(I, k, 3)
    except gomel_amd.GomelError:
        pers = None                                             # hop not a multiple of 4: the persistent kernel does not take it
    if pers is not None:
        assert tmp["fast_path"] == 2 and all(np.array_equal(a, b) for a, b in zip(pers, gens))


# ------------------------------------------------------------------------------------------------ 64-bit mode at hops that are no multiple of N/16
@pytest.mark.parametrize("N,hop,K", [(4096, 882, 96), (4096, 441, 96), (4096, 2050, 96), (4096, 3001, 96), (2048, 276, 80), (2048, 275, 80), (2048, 1025, 80),
                                      (1024, 300, 80), (1024, 275, 80), (1024, 600, 80), (512, 100, 40), (512, 161, 40), (256, 50, 40), (256, 51, 40), (256, 255, 40)])
def test_any_hop_fused_kernel_in_64_bit_mode(oracle, N, hop, K):
    """k_gl_fused_any_d: 64-bit mode keeps a fused kernel at hops that are no multiple of N/16 (even hops: the window moves through LDS
    by pairs, odd hops: by doubles).  Against the generic fp64 kernels (same arithmetic per frame, other summation order of the
    overlap-add) on a ragged batch, and against the fp64 oracle at the level 64-bit mode promises."""
    import gomel_amd
    fields = dict(NumMels=K, MelFmax=8000.0, Window=hop, Resolut=N, GriffinLimIterations=4)
    gmin = (N - 1) // hop
    frames = [gmin * 4 + 3, 1, 2, gmin + 1, 150, 64, 65]
    mels = [oracle.random_logmel(f, K, seed=500 + i) for i, f in enumerate(frames)]
    seeds = list(range(len(frames)))
    outs = {}
    for path in (2, 1):
        m = mel_obj(fields)
        e = m.engine()
        e.set_path(path)
        e.set_precision(64)
        b = gomel_amd.Batch(e, [len(mm) for mm in mels])
        b.upload(mels, None, seeds)
        tm = b.run_timed()
        assert tm["fast_path"] == (3 if path == 2 else 0) and tm["iter64_launches"] == (4 if path == 2 else 0), tm
        outs[path] = b.download()
        b.close(); e.close()
    for a, r in zip(outs[2], outs[1]):
        assert a.shape == r.shape and rms(a - r) < 1e-12 * max(1.0, rms(r)), rms(a - r)
    cfg = oracle.MelConfig(**fields)
    for i in (0, 1, 4):
        want = oracle.from_mel(cfg, mels[i].copy(), oracle.x0_splitmix(seeds[i], len(outs[2][i])))
        assert rms(outs[2][i] - want) < 1e-9 * max(1.0, rms(want))


def test_host_to_host_call_in_the_parity_modes(oracle, coracle):
    """gomel_from_mel_batch (the chunked upload | compute | download pipeline, what bench.py times) in 64-bit mode and under the
    precision schedule: 64 utterances x 200 frames is enough for the call to be cut into chunks, each of which the fused kernels
    take.  64-bit mode agrees with the oracle at rounding level whatever the chunking; the schedule stays far inside the bar; a
    Batch of the same utterances (one whole-batch plan: other segments, other summation order) gives the same within rounding."""
    import gomel_amd
    fields = dict(T, GriffinLimIterations=20)
    U, F, K = 64, 200, 192
    mels, seeds = cfg3_inputs(U, F, K, first=7000)
    m = mel_obj(fields)
    eng = m.engine()
    cfg = oracle.MelConfig(**fields)
    L = oracle.out_len(cfg, F * K)
    assert eng.plan_chunks([F * K] * U)[-1] == U and len(eng.plan_chunks([F * K] * U)) > 2          # more than one chunk
    pick = list(range(0, U, 7))
    want, _ = coracle.from_mel_batch(cfg, [mels[u] for u in pick], [oracle.x0_splitmix(seeds[u], L) for u in pick], threads=min(10, os.cpu_count() or 1))
    eng.set_precision(64)
    o64 = eng.from_mel_batch(mels, None, seeds)
    assert max(rms(o64[u] - w) for u, w in zip(pick, want)) < 1e-9
    b = gomel_amd.Batch(eng, [F * K] * U)
    b.upload(mels, None, seeds)
    b.run()
    r64 = b.download()
    b.close()
    assert max(rms(a - c) for a, c in zip(o64, r64)) < 1e-11
    eng.set_precision_schedule(10)
    osch = eng.from_mel_batch(mels, None, seeds)
    esch = [rms(osch[u] - w) for u, w in zip(pick, want)]
    print("host-to-host, schedule 10 + 10: max %.2e" % max(esch))
    assert max(esch) < 1e-5
    eng.set_precision_schedule(0)
    o32 = eng.from_mel_batch(mels, None, seeds)
    assert max(rms(o32[u] - w) for u, w in zip(pick, want)) < RMS_TOL                                # 20 iterations: fp32 is still at 1e-6


def test_bound_buffers_equal_the_marshalling_call(oracle):
    """Engine.bind_buffers / from_mel_batch_bound: the pointer arrays of gomel_from_mel_batch built once for buffers a caller reuses
    (bench.py's timed loop): same bits as the call that marshals every time, the waveforms land in the bound arrays, x0 arrays bind too."""
    import gomel_amd
    fields = dict(S, GriffinLimIterations=4)
    m = mel_obj(fields)
    eng = m.engine()
    frames = [30, 7, 64, 12]
    mels = [oracle.random_logmel(f, 80, seed=70 + i) for i, f in enumerate(frames)]
    cfg = oracle.MelConfig(**fields)
    outs = [np.zeros(oracle.out_len(cfg, len(mm)) + 5) for mm in mels]            # roomier than needed: the views are trimmed
    b = eng.bind_buffers(mels, outs)
    got = eng.from_mel_batch_bound(b, [3, 4, 5, 6])
    ref = eng.from_mel_batch(mels, None, [3, 4, 5, 6])
    assert all(np.array_equal(a, r) and a.base is not None for a, r in zip(got, ref))
    assert all(np.array_equal(o[:len(r)], r) for o, r in zip(outs, ref))
    x0s = [oracle.x0_splitmix(9 + i, len(r)) for i, r in enumerate(ref)]
    bx = eng.bind_buffers(mels, outs, x0s)
    gotx = eng.from_mel_batch_bound(bx)
    refx = eng.from_mel_batch(mels, x0s, None)
    assert all(np.array_equal(a, r) for a, r in zip(gotx, refx))
    with pytest.raises(ValueError):
        eng.bind_buffers(mels, outs[:2])
    with pytest.raises(ValueError):
        eng.bind_buffers([mm.astype(np.float32) for mm in mels], outs)
