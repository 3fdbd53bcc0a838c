"""CPU tests of the host-to-host pipeline's chunk planner (gomel_plan_chunks: host logic only, no device): how a
gomel_from_mel_batch call is cut into the chunks whose upload / compute / download overlap."""
import ctypes

import numpy as np
import pytest


def plan(L, fields, frames, fp64_iterations=0):
    """fp64_iterations = 0: the plan of the fp32 fast mode (what rounds 2-3 tuned); None: gomel_plan_chunks, i.e. the default precision mode"""
    from gomel_amd._lib import GomelConfig
    c = GomelConfig()
    L.gomel_default_config(ctypes.byref(c))
    for k, v in fields.items():
        setattr(c, k, v)
    n = len(frames)
    npairs = (ctypes.c_size_t * n)(*[f * c.num_mels for f in frames])
    bounds = (ctypes.c_size_t * 66)()
    nch = ctypes.c_size_t(0)
    if fp64_iterations is None:
        rc = L.gomel_plan_chunks(ctypes.byref(c), n, npairs, bounds, 66, ctypes.byref(nch))
    else:
        rc = L.gomel_plan_chunks_mode(ctypes.byref(c), fp64_iterations, n, npairs, bounds, 66, ctypes.byref(nch))
    return rc, [bounds[i] for i in range(nch.value + 1)]


@pytest.fixture(scope="module")
def L():
    from gomel_amd import build, _lib
    build.build()
    return _lib.lib()


T = dict(num_mels=192, window=1280, resolut=4096)
S = dict(num_mels=80, window=256, resolut=1024)


def test_config3_plan_has_half_size_ends_and_round_filling_chunks(L, monkeypatch):
    monkeypatch.delenv("GOMEL_CHUNKS", raising=False)
    monkeypatch.delenv("GOMEL_CHUNK_SHARES", raising=False)
    rc, b = plan(L, T, [400] * 256)
    assert rc == 0 and b[0] == 0 and b[-1] == 256 and b == sorted(set(b))
    sizes = np.diff(b)
    assert len(sizes) == 9                                        # 102 400 frames / 12 288 = 8 full chunks' worth: half + 7 + half
    assert sizes[0] < 0.7 * np.median(sizes) and sizes[-1] < 0.7 * np.median(sizes)
    # a launch lasts ceil(frames / 1024 streams) frame times: 33 utterances of 400 frames fill 13 x 1024 to 99 %, 32 only to 96 %
    fill = lambda f: f / (1024 * -(-f // 1024))
    assert all(fill(s * 400) > 0.97 for s in sizes[1:-1]) and 33 in sizes
    rc, b = plan(L, S, [400] * 256)                               # shorter frames: ~32k-frame chunks
    assert rc == 0 and len(b) - 1 == 4 and b[-1] == 256


def test_small_and_ragged_calls(L, monkeypatch):
    monkeypatch.delenv("GOMEL_CHUNKS", raising=False)
    monkeypatch.delenv("GOMEL_CHUNK_SHARES", raising=False)
    assert plan(L, T, [342]) == (0, [0, 1])                       # one utterance: one chunk
    assert plan(L, T, [400] * 16) == (0, [0, 16])                 # below one full chunk: no pipelining
    rc, b = plan(L, T, [400] * 32)                                # one full chunk's worth: two halves
    assert rc == 0 and len(b) == 3 and b[-1] == 32
    rng = np.random.default_rng(0)
    frames = rng.integers(1, 900, size=500).tolist()
    rc, b = plan(L, T, frames)
    assert rc == 0 and b[0] == 0 and b[-1] == 500 and all(x < y for x, y in zip(b, b[1:]))
    chunk_frames = [sum(frames[lo:hi]) for lo, hi in zip(b, b[1:])]
    mid = np.median(chunk_frames[1:-1])
    assert all(0.6 * mid < f < 1.5 * mid for f in chunk_frames[1:-1])       # boundaries move by at most two utterances
    rc, b = plan(L, T, [20000, 20000, 5])                          # fewer utterances than chunks: still a valid cover
    assert rc == 0 and b[0] == 0 and b[-1] == 3 and all(x < y for x, y in zip(b, b[1:]))


def test_overrides_and_errors(L, monkeypatch):
    monkeypatch.setenv("GOMEL_CHUNKS", "4")
    rc, b = plan(L, T, [400] * 256)
    assert rc == 0 and list(np.diff(b)) == [64] * 4
    monkeypatch.delenv("GOMEL_CHUNKS")
    monkeypatch.setenv("GOMEL_CHUNK_SHARES", "1,2,1")
    rc, b = plan(L, T, [400] * 256)
    assert rc == 0 and list(np.diff(b)) == [64, 128, 64]
    monkeypatch.delenv("GOMEL_CHUNK_SHARES")
    from gomel_amd._lib import GomelConfig
    c = GomelConfig()
    L.gomel_default_config(ctypes.byref(c))
    npairs = (ctypes.c_size_t * 2)(160 * 10, 161)
    bounds = (ctypes.c_size_t * 66)()
    nch = ctypes.c_size_t(0)
    assert L.gomel_plan_chunks(ctypes.byref(c), 2, npairs, bounds, 66, ctypes.byref(nch)) == 3        # not a multiple of NumMels
    assert L.gomel_plan_chunks(ctypes.byref(c), 0, npairs, bounds, 66, ctypes.byref(nch)) == 4        # empty
    assert L.gomel_plan_chunks(None, 2, npairs, bounds, 66, ctypes.byref(nch)) == 7
    npairs = (ctypes.c_size_t * 64)(*([192 * 400] * 64))
    c.num_mels, c.window, c.resolut = 192, 1280, 4096
    assert L.gomel_plan_chunks(ctypes.byref(c), 64, npairs, bounds, 2, ctypes.byref(nch)) == 6        # bounds too small


def test_calls_with_fp64_iterations_take_chunks_twice_the_size(L, monkeypatch):
    """The default precision mode (and any schedule / 64-bit mode) runs fp64 launches that take 1.75-2.4x as long per frame: chunks of
    ~24k frames still hide the copies and halve the number of chunk-sized launches (75.8 ms against 79.0 per config-3 call)."""
    from gomel_amd._lib import SCHEDULE_AUTO
    monkeypatch.delenv("GOMEL_CHUNKS", raising=False)
    monkeypatch.delenv("GOMEL_CHUNK_SHARES", raising=False)
    t60 = dict(T, griffin_lim_iterations=60)
    rc, fast = plan(L, t60, [400] * 256, 0)
    rc2, dflt = plan(L, t60, [400] * 256, None)
    rc3, auto = plan(L, t60, [400] * 256, SCHEDULE_AUTO)
    rc4, full = plan(L, t60, [400] * 256, -1)
    assert rc == rc2 == rc3 == rc4 == 0 and dflt == auto == full
    assert len(fast) - 1 == 9 and len(dflt) - 1 == 5                 # half + 7 + half against half + 3 + half
    sizes = np.diff(dflt)
    assert sizes[0] < 0.7 * np.median(sizes) and sizes[-1] < 0.7 * np.median(sizes) and dflt[-1] == 256
    assert plan(L, t60, [400] * 32, None) == (0, [0, 32])            # below one (doubled) full chunk: no pipelining
    assert L.gomel_plan_chunks_mode(None, 0, 1, None, None, 0, None) == 7
