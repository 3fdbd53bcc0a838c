"""Host-side mirror of the reference's Go package `mel` (mel/mel.go) over the C ABI in include/gomel_hip.h.

Same exported names, argument meaning and error behaviour as the reference for the hot path:
`NewMel`, `Mel.ToMel`, `Mel.FromMel`, `Mel.ToWavPng`, plus `Mel.ToMelWav`, `Mel.Image`, `LoadWav`, `SaveWav`,
`ErrFileNotLoaded`.  All numerics run in libgomel_hip.so on the GPU; this module only marshals buffers and
keeps the reference's host-side side effects (in-place exp, in-place VolumeBoost, SampleRate fallback).
There is no CPU fallback.
"""
import ctypes
import random

import numpy as np

from . import _lib
from ._lib import GomelConfig, GomelError, GomelTiming, c_double_p, check


class ErrFileNotLoaded(Exception):
    """mel/mel.go:43 `var ErrFileNotLoaded = errors.New("wavNotLoaded")`."""

    def __init__(self):
        super().__init__("wavNotLoaded")


class DevicePointer:
    """`numel` float64 elements in the memory of the handle's device at address `ptr` (from hipMalloc, another library, a pipeline stage
    that keeps its spectrograms in HBM): what the batch calls take wherever they take an array (include/gomel_hip.h "Pointers").  The
    caller owns the memory and must have finished writing it (the library copies on its own streams)."""
    is_cuda = True
    dtype = "float64"

    def __init__(self, ptr, numel):
        self._ptr, self._numel = int(ptr), int(numel)

    def data_ptr(self):
        return self._ptr

    def numel(self):
        return self._numel

    def is_contiguous(self):
        return True


def _is_device_tensor(a):
    """A torch CUDA tensor or a DevicePointer: its memory is handed to the library as a device pointer (include/gomel_hip.h "Pointers")."""
    return hasattr(a, "data_ptr") and getattr(a, "is_cuda", False)


def _device_ptr(a, numel, what):
    if str(a.dtype).replace("torch.", "") != "float64" or not a.is_contiguous() or a.numel() != numel:
        raise ValueError("%s: device memory must be contiguous float64 with %d elements" % (what, numel))
    if not isinstance(a, DevicePointer):
        import torch
        torch.cuda.current_stream(a.device).synchronize()   # the library copies on its own stream
    return ctypes.cast(ctypes.c_void_p(a.data_ptr()), c_double_p)


def _as_pairs(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if a.ndim != 2 or a.shape[1] != 2:
        raise ValueError("expected an [n, 2] float64 array (Go [][2]float64)")
    return a


class Engine:
    """One gomel_handle: configuration + device tables + stream (replaces stft.New and undomel's index math)."""

    def __init__(self, cfg, tables=None):
        self._h = ctypes.c_void_p()
        self.cfg = cfg
        L = _lib.lib()
        if tables is None:
            check(L.gomel_create(ctypes.byref(cfg), ctypes.byref(self._h)), "gomel_create")
        else:
            buf = (ctypes.c_char * len(tables)).from_buffer_copy(tables)
            check(L.gomel_create_from_tables(ctypes.byref(cfg), buf, len(tables), ctypes.byref(self._h)), "gomel_create_from_tables")

    def close(self):
        """gomel_destroy.  A handle with submitted jobs that have not been waited for refuses (GOMEL_ERR_BUSY): it STAYS valid and this
        raises -- wait for the jobs, then close again (dropping it here would leak the handle and its device buffers)."""
        if self._h:
            rc = _lib.lib().gomel_destroy(self._h)
            if rc == 13:
                raise GomelError(rc, "gomel_destroy")
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- tables (the "mel basis" that rank 0 broadcasts) ---------------------------------------------
    def export_tables(self):
        L = _lib.lib()
        n = L.gomel_tables_bytes(self._h)
        buf = ctypes.create_string_buffer(n)
        check(L.gomel_tables_export(self._h, buf, n), "gomel_tables_export")
        return bytes(buf.raw)

    def set_window(self, w):
        w = np.ascontiguousarray(w, dtype=np.float64)
        check(_lib.lib().gomel_set_window(self._h, w.ctypes.data_as(c_double_p), w.size), "gomel_set_window")

    def get_window(self):
        w = np.zeros(self.cfg.resolut)
        check(_lib.lib().gomel_get_window(self._h, w.ctypes.data_as(c_double_p), w.size), "gomel_get_window")
        return w

    def set_path(self, mode):
        check(_lib.lib().gomel_set_path(self._h, int(mode)), "gomel_set_path")

    def set_call_mode(self, mode):
        """gomel_set_call_mode: 1 = every synchronous batch call is one whole batch (submit + wait), for callers that overlap calls
        from several threads; 0 (default) = chunked."""
        check(_lib.lib().gomel_set_call_mode(self._h, int(mode)), "gomel_set_call_mode")

    def set_precision(self, bits):
        check(_lib.lib().gomel_set_precision(self._h, int(bits)), "gomel_set_precision")

    def set_precision_schedule(self, fp64_iterations):
        """The first `fp64_iterations` iterations of every call in fp64, the rest in fp32 (gomel_set_precision_schedule);
        "auto" (or _lib.SCHEDULE_AUTO) = the calibrated default."""
        k = _lib.SCHEDULE_AUTO if fp64_iterations == "auto" else int(fp64_iterations)
        check(_lib.lib().gomel_set_precision_schedule(self._h, k), "gomel_set_precision_schedule")

    def fp64_iterations(self, iterations=-1):
        """gomel_get_precision: how many of a call's iterations run in fp64 under the engine's current mode."""
        k = ctypes.c_int32(0)
        check(_lib.lib().gomel_get_precision(self._h, int(iterations), ctypes.byref(k)), "gomel_get_precision")
        return k.value

    def plan_chunks(self, n_pairs, fp64_iterations=None):
        """gomel_plan_chunks_mode: how a gomel_from_mel_batch call over utterances of these lengths is cut -- the utterance indices
        [0, b1, ..., n] at which its chunks begin and end (host logic, no device work) -- for a call with that many fp64 iterations
        (None: what this engine's current precision mode gives; 0: the fp32 fast mode)."""
        n = len(n_pairs)
        np_ = (ctypes.c_size_t * max(n, 1))(*[int(v) for v in n_pairs])
        bounds = (ctypes.c_size_t * (n + 2))()
        nch = ctypes.c_size_t(0)
        k = self.fp64_iterations() if fp64_iterations is None else int(fp64_iterations)
        check(_lib.lib().gomel_plan_chunks_mode(ctypes.byref(self.cfg), k, n, np_, bounds, n + 2, ctypes.byref(nch)), "gomel_plan_chunks_mode")
        return [int(bounds[i]) for i in range(nch.value + 1)]

    def device_name(self):
        buf = ctypes.create_string_buffer(256)
        check(_lib.lib().gomel_device_name(self._h, buf, 256), "gomel_device_name")
        return buf.value.decode()

    def out_len(self, n_pairs):
        return _lib.lib().gomel_out_len(ctypes.byref(self.cfg), n_pairs)

    # -- FromMel -------------------------------------------------------------------------------------
    def from_mel(self, mel_pairs, x0=None, seed=0):
        m = _as_pairs(mel_pairs)
        L = _lib.lib()
        n = self.out_len(len(m)) if len(m) else 0
        out = np.empty(max(n, 1), dtype=np.float64)
        x0p = None
        if x0 is not None:
            x0 = np.ascontiguousarray(x0, dtype=np.float64)
            if x0.shape != (n,):
                raise ValueError("x0 must have length %d" % n)
            x0p = x0.ctypes.data_as(c_double_p)
        got = ctypes.c_size_t(0)
        check(L.gomel_from_mel(self._h, m.ctypes.data_as(c_double_p), len(m), x0p, ctypes.c_uint64(seed & (2 ** 64 - 1)),
                               out.ctypes.data_as(c_double_p), out.size, ctypes.byref(got)), "gomel_from_mel")
        return out[:got.value]

    def istft(self, magnitudes, iterations, x0=None, seed=0, spec=False):
        """gomel_istft: Griffin-Lim from linear magnitudes [F][N/2+1] (the reference's exported ISTFT, mel/mel.go:76-139; an odd N
        keeps bins N/2 and N/2+1: rows of N/2+2).  With spec=True (gomel_istft_spec) also returns what the reference leaves in its
        spectrogram argument in those bins (complex, same row length; None when iterations == 0: the reference's loop does not run)."""
        m = np.ascontiguousarray(magnitudes, dtype=np.float64)
        H1 = self.cfg.resolut // 2 + 1 + (self.cfg.resolut & 1)
        if m.ndim != 2 or m.shape[1] != H1:
            raise ValueError("magnitudes must be [frames, %d]" % H1)
        F = m.shape[0]
        n = self.cfg.resolut + (F - 1) * self.cfg.window if F else 0
        out = np.empty(max(n, 1), dtype=np.float64)
        x0p = None
        if x0 is not None:
            x0 = np.ascontiguousarray(x0, dtype=np.float64)
            if x0.shape != (n,):
                raise ValueError("x0 must have length %d" % n)
            x0p = x0.ctypes.data_as(c_double_p)
        got = ctypes.c_size_t(0)
        if not spec:
            check(_lib.lib().gomel_istft(self._h, m.ctypes.data_as(c_double_p), F, int(iterations), x0p, ctypes.c_uint64(seed & (2 ** 64 - 1)),
                                         out.ctypes.data_as(c_double_p), out.size, ctypes.byref(got)), "gomel_istft")
            return out[:got.value]
        sp = np.zeros((max(F, 1), H1, 2), dtype=np.float64)
        check(_lib.lib().gomel_istft_spec(self._h, m.ctypes.data_as(c_double_p), F, int(iterations), x0p, ctypes.c_uint64(seed & (2 ** 64 - 1)),
                                          out.ctypes.data_as(c_double_p), out.size, ctypes.byref(got), sp.ctypes.data_as(c_double_p)), "gomel_istft_spec")
        return out[:got.value], (None if int(iterations) == 0 else sp[:F, :, 0] + 1j * sp[:F, :, 1])

    def from_mel_batch_f32(self, mels, x0s=None, seeds=None, outs=None):
        """gomel_from_mel_batch_f32: float32 log-mel pairs in, float32 waveforms out (half the PCIe bytes of the float64 call)."""
        ms = [np.ascontiguousarray(m, dtype=np.float32) for m in mels]
        n = len(ms)
        if n == 0:
            return []
        if any(m.ndim != 2 or m.shape[1] != 2 for m in ms):
            raise ValueError("expected [n, 2] arrays")
        lens = [self.out_len(len(m)) for m in ms]
        if outs is None:
            outs = [np.empty(max(L, 1), dtype=np.float32) for L in lens]
        elif len(outs) != n or any(o.dtype != np.float32 or o.ndim != 1 or len(o) < L or not o.flags.c_contiguous for o, L in zip(outs, lens)):
            raise ValueError("outs must be C-contiguous float32 arrays of at least the output lengths")
        xs = None
        if x0s is not None:
            xs = [None if x is None else np.ascontiguousarray(x, dtype=np.float32) for x in x0s]
        fpp = ctypes.POINTER(_lib.c_float_p)
        mp = (_lib.c_float_p * n)(*[m.ctypes.data_as(_lib.c_float_p) for m in ms])
        op = (_lib.c_float_p * n)(*[o.ctypes.data_as(_lib.c_float_p) for o in outs])
        xp = None if xs is None else (_lib.c_float_p * n)(*[None if x is None else x.ctypes.data_as(_lib.c_float_p) for x in xs])
        sp = None if seeds is None else (ctypes.c_uint64 * n)(*[int(s) & (2 ** 64 - 1) for s in seeds])
        npairs = (ctypes.c_size_t * n)(*[len(m) for m in ms])
        caps = (ctypes.c_size_t * n)(*[len(o) for o in outs])
        check(_lib.lib().gomel_from_mel_batch_f32(self._h, n, ctypes.cast(mp, fpp), npairs, None if xp is None else ctypes.cast(xp, fpp), sp,
                                                  ctypes.cast(op, fpp), caps), "gomel_from_mel_batch_f32")
        return [o[:L] for o, L in zip(outs, lens)]

    def from_mel_batch(self, mels, x0s=None, seeds=None, outs=None):
        """gomel_from_mel_batch: host buffers in, host buffers out (chunked, copy/compute overlapped inside the library).
        `outs` may be preallocated float64 arrays (caller-owned buffers); otherwise fresh arrays are returned."""
        ms = [_as_pairs(m) for m in mels]
        n = len(ms)
        if n == 0:
            return []
        K, N, hop = self.cfg.num_mels, self.cfg.resolut, self.cfg.window
        # gomel_out_len without a C call per utterance: N + (F-1) hop, 0 for a length that is no whole number of frames (the C
        # call below refuses those with GOMEL_ERR_BAD_LENGTH)
        lens = [N + (len(m) // K - 1) * hop if len(m) and len(m) % K == 0 else 0 for m in ms]
        if outs is None:
            outs = [np.empty(max(L, 1), dtype=np.float64) for L in lens]
        elif len(outs) != n or any(o.dtype != np.float64 or o.ndim != 1 or len(o) < L or not o.flags.c_contiguous for o, L in zip(outs, lens)):
            raise ValueError("outs must be C-contiguous float64 arrays of at least the output lengths")
        xs = None
        if x0s is not None:
            xs = [None if x is None else np.ascontiguousarray(x, dtype=np.float64) for x in x0s]
            for x, L in zip(xs, lens):
                if x is not None and x.shape != (L,):
                    raise ValueError("x0 length mismatch")
        dpp = ctypes.POINTER(c_double_p)

        def ptrs(arrs):   # one uintp array instead of n ctypes pointer objects (a 256-utterance call spent ~1 ms building those)
            return np.array([0 if a is None else a.__array_interface__["data"][0] for a in arrs], dtype=np.uintp)
        mp, op = ptrs(ms), ptrs(outs)
        xp = None if xs is None else ptrs(xs)
        sp = None if seeds is None else np.array([int(s) & (2 ** 64 - 1) for s in seeds], dtype=np.uint64)
        npairs = np.array([len(m) for m in ms], dtype=np.uintp)
        caps = np.array([len(o) for o in outs], dtype=np.uintp)
        as_p = lambda a, t: a.ctypes.data_as(t)
        check(_lib.lib().gomel_from_mel_batch(self._h, n, as_p(mp, dpp), as_p(npairs, ctypes.POINTER(ctypes.c_size_t)), None if xp is None else as_p(xp, dpp),
                                              None if sp is None else as_p(sp, ctypes.POINTER(ctypes.c_uint64)), as_p(op, dpp),
                                              as_p(caps, ctypes.POINTER(ctypes.c_size_t))), "gomel_from_mel_batch")
        return [o[:L] for o, L in zip(outs, lens)]

    def bind_buffers(self, mels, outs, x0s=None):
        """For callers that reuse their buffers call after call (a serving loop, bench.py): validate the arrays and build the
        pointer / length arrays of gomel_from_mel_batch ONCE -- what a C or Go caller does when it fills its `double**` arrays at
        start-up.  Returns a BoundBuffers for from_mel_batch_bound; it keeps the arrays alive."""
        return BoundBuffers(self, mels, outs, x0s)

    def from_mel_batch_bound(self, bound, seeds=None):
        """gomel_from_mel_batch on buffers bound with bind_buffers: no per-call marshalling beyond the seeds.  The waveforms land in
        the bound `outs` (returned, trimmed to the output lengths)."""
        if bound.engine is not self:
            raise ValueError("buffers were bound to another engine")
        n = bound.n
        sp = None if seeds is None else np.array([int(s_) & (2 ** 64 - 1) for s_ in seeds], dtype=np.uint64)
        if sp is not None and len(sp) != n:
            raise ValueError("one seed per utterance")
        dpp = ctypes.POINTER(c_double_p)
        check(_lib.lib().gomel_from_mel_batch(self._h, n, bound.mp.ctypes.data_as(dpp), bound.npairs.ctypes.data_as(ctypes.POINTER(ctypes.c_size_t)),
                                              None if bound.xp is None else bound.xp.ctypes.data_as(dpp),
                                              None if sp is None else sp.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), bound.op.ctypes.data_as(dpp),
                                              bound.caps.ctypes.data_as(ctypes.POINTER(ctypes.c_size_t))), "gomel_from_mel_batch")
        return bound.views

    def submit_bound(self, bound, seeds=None):
        """gomel_from_mel_batch_submit on bound buffers: queues the call (upload | whole-batch kernels | download on three streams)
        and returns a Job at once; Job.wait() blocks until the waveforms are in the bound `outs`.  Keep two jobs in flight and the
        copies of one hide behind the kernels of the other.  The bound buffers must not be touched until the job has been waited for."""
        if bound.engine is not self:
            raise ValueError("buffers were bound to another engine")
        n = bound.n
        sp = None if seeds is None else np.array([int(s_) & (2 ** 64 - 1) for s_ in seeds], dtype=np.uint64)
        if sp is not None and len(sp) != n:
            raise ValueError("one seed per utterance")
        dpp = ctypes.POINTER(c_double_p)
        job = ctypes.c_void_p()
        check(_lib.lib().gomel_from_mel_batch_submit(self._h, n, bound.mp.ctypes.data_as(dpp), bound.npairs.ctypes.data_as(ctypes.POINTER(ctypes.c_size_t)),
                                                     None if bound.xp is None else bound.xp.ctypes.data_as(dpp),
                                                     None if sp is None else sp.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), bound.op.ctypes.data_as(dpp),
                                                     bound.caps.ctypes.data_as(ctypes.POINTER(ctypes.c_size_t)), ctypes.byref(job)), "gomel_from_mel_batch_submit")
        return Job(job, bound)

    def submit_f32(self, mels, x0s=None, seeds=None, outs=None):
        """gomel_from_mel_batch_submit_f32: the float32 host-I/O call as a job (see submit_bound).  The arrays are kept alive by the
        returned Job; Job.wait() returns the float32 waveforms."""
        ms = [np.ascontiguousarray(m, dtype=np.float32) for m in mels]
        n = len(ms)
        if n == 0 or any(m.ndim != 2 or m.shape[1] != 2 for m in ms):
            raise ValueError("expected a non-empty list of [n, 2] arrays")
        lens = [self.out_len(len(m)) for m in ms]
        if outs is None:
            outs = [np.empty(max(L, 1), dtype=np.float32) for L in lens]
        elif len(outs) != n or any(o.dtype != np.float32 or o.ndim != 1 or len(o) < L or not o.flags.c_contiguous for o, L in zip(outs, lens)):
            raise ValueError("outs must be C-contiguous float32 arrays of at least the output lengths")
        xs = None if x0s is None else [None if x is None else np.ascontiguousarray(x, dtype=np.float32) for x in x0s]
        fpp = ctypes.POINTER(_lib.c_float_p)
        mp = (_lib.c_float_p * n)(*[m.ctypes.data_as(_lib.c_float_p) for m in ms])
        op = (_lib.c_float_p * n)(*[o.ctypes.data_as(_lib.c_float_p) for o in outs])
        xp = None if xs is None else (_lib.c_float_p * n)(*[None if x is None else x.ctypes.data_as(_lib.c_float_p) for x in xs])
        sp = None if seeds is None else (ctypes.c_uint64 * n)(*[int(s_) & (2 ** 64 - 1) for s_ in seeds])
        npairs = (ctypes.c_size_t * n)(*[len(m) for m in ms])
        caps = (ctypes.c_size_t * n)(*[len(o) for o in outs])
        job = ctypes.c_void_p()
        check(_lib.lib().gomel_from_mel_batch_submit_f32(self._h, n, ctypes.cast(mp, fpp), npairs, None if xp is None else ctypes.cast(xp, fpp), sp,
                                                         ctypes.cast(op, fpp), caps, ctypes.byref(job)), "gomel_from_mel_batch_submit_f32")

        class _Views:
            views = [o[:L] for o, L in zip(outs, lens)]
            keep = (ms, xs, outs)
        return Job(job, _Views)

    def target_magnitudes(self, mel_pairs):
        m = _as_pairs(mel_pairs)
        H = self.cfg.resolut // 2
        F = len(m) // self.cfg.num_mels
        out = np.empty((max(F, 1), H + 1), dtype=np.float32)
        check(_lib.lib().gomel_target_magnitudes(self._h, m.ctypes.data_as(c_double_p), len(m), out.ctypes.data_as(_lib.c_float_p)),
              "gomel_target_magnitudes")
        return out

    # -- ToMel ---------------------------------------------------------------------------------------
    def to_mel(self, wave):
        w = np.ascontiguousarray(wave, dtype=np.float64)
        L = _lib.lib()
        cap = L.gomel_to_mel_pairs(ctypes.byref(self.cfg), w.size)
        out = np.empty((max(cap, 1), 2), dtype=np.float64)
        got = ctypes.c_size_t(0)
        check(L.gomel_to_mel(self._h, w.ctypes.data_as(c_double_p), w.size, out.ctypes.data_as(c_double_p), cap, ctypes.byref(got)), "gomel_to_mel")
        return out[:got.value]

    def to_mel_batch(self, waves):
        """gomel_to_mel_batch: many independent ToMel calls in one go (one STFT launch, one domel launch over all frames)."""
        ws = [np.ascontiguousarray(w, dtype=np.float64) for w in waves]
        n = len(ws)
        if n == 0:
            return []
        L = _lib.lib()
        caps = [L.gomel_to_mel_pairs(ctypes.byref(self.cfg), w.size) for w in ws]
        outs = [np.empty((max(c, 1), 2), dtype=np.float64) for c in caps]
        dpp = ctypes.POINTER(c_double_p)
        wp = (c_double_p * n)(*[w.ctypes.data_as(c_double_p) for w in ws])
        op = (c_double_p * n)(*[o.ctypes.data_as(c_double_p) for o in outs])
        ns = (ctypes.c_size_t * n)(*[w.size for w in ws])
        cp = (ctypes.c_size_t * n)(*caps)
        got = (ctypes.c_size_t * n)()
        check(L.gomel_to_mel_batch(self._h, n, ctypes.cast(wp, dpp), ns, ctypes.cast(op, dpp), cp, got), "gomel_to_mel_batch")
        return [o[:g] for o, g in zip(outs, got)]

    # -- FFT test hooks ------------------------------------------------------------------------------
    def rfft_frames(self, frames):
        f = np.ascontiguousarray(frames, dtype=np.float32)
        N = self.cfg.resolut
        assert f.ndim == 2 and f.shape[1] == N
        out = np.empty((f.shape[0], N // 2 + 1, 2), dtype=np.float32)
        check(_lib.lib().gomel_rfft_frames(self._h, f.ctypes.data_as(_lib.c_float_p), f.shape[0], out.ctypes.data_as(_lib.c_float_p)), "gomel_rfft_frames")
        return out[..., 0] + 1j * out[..., 1]

    def irfft_frames(self, spec):
        N = self.cfg.resolut
        s = np.asarray(spec)
        assert s.ndim == 2 and s.shape[1] == N // 2 + 1
        buf = np.ascontiguousarray(np.stack([s.real, s.imag], axis=2), dtype=np.float32)
        out = np.empty((s.shape[0], N), dtype=np.float32)
        check(_lib.lib().gomel_irfft_frames(self._h, buf.ctypes.data_as(_lib.c_float_p), s.shape[0], out.ctypes.data_as(_lib.c_float_p)), "gomel_irfft_frames")
        return out


class Job:
    """A submitted call (gomel_job): wait() returns the bound output views once the waveforms have landed; every job must be waited for."""

    def __init__(self, handle, bound):
        self._j, self._bound = handle, bound

    def done(self):
        if not self._j:
            return True
        f = ctypes.c_int(0)
        check(_lib.lib().gomel_job_poll(self._j, ctypes.byref(f)), "gomel_job_poll")
        return bool(f.value)

    def wait(self, timeout_ms=None):
        """Blocks until the waveforms have landed.  With `timeout_ms` (gomel_job_wait_timeout): raises TimeoutError when the job is still
        running at the deadline -- the job stays valid (wait again); any other outcome finishes and frees it."""
        if self._j:
            if timeout_ms is None:
                j, self._j = self._j, None
                check(_lib.lib().gomel_job_wait(j), "gomel_job_wait")
            else:
                rc = _lib.lib().gomel_job_wait_timeout(self._j, int(timeout_ms))
                if rc == 13:                                   # GOMEL_ERR_BUSY: still running, still ours
                    raise TimeoutError("gomel_job_wait_timeout: the job is still running after %d ms" % int(timeout_ms))
                self._j = None
                check(rc, "gomel_job_wait_timeout")
        return self._bound.views

    def __del__(self):
        try:
            if self._j:
                _lib.lib().gomel_job_wait(self._j)
                self._j = None
        except Exception:
            pass


class Batch:
    """Device-resident batch of independent utterances (create -> upload -> run -> download)."""

    def __init__(self, engine, n_pairs):
        self.engine = engine
        self.n_pairs = [int(n) for n in n_pairs]
        self._b = ctypes.c_void_p()
        arr = (ctypes.c_size_t * len(self.n_pairs))(*self.n_pairs)
        check(_lib.lib().gomel_batch_create(engine._h, len(self.n_pairs), arr, ctypes.byref(self._b)), "gomel_batch_create")
        self.lens = [engine.out_len(n) for n in self.n_pairs]
        self._keep = None

    def upload(self, mels, x0s=None, seeds=None):
        """Each spectrogram / x0 is a host array (numpy) or a torch CUDA tensor of the engine's device (float64, contiguous):
        device tensors are copied HBM -> HBM, nothing crosses PCIe."""
        n = len(self.n_pairs)
        if len(mels) != n:
            raise ValueError("mel shapes do not match the batch plan")
        ms = [m if _is_device_tensor(m) else _as_pairs(m) for m in mels]
        if any((not _is_device_tensor(m)) and len(m) != k for m, k in zip(ms, self.n_pairs)):
            raise ValueError("mel shapes do not match the batch plan")
        mp = (c_double_p * n)(*[_device_ptr(m, 2 * k, "mel") if _is_device_tensor(m) else m.ctypes.data_as(c_double_p)
                                for m, k in zip(ms, self.n_pairs)])
        xp = None
        xs = None
        if x0s is not None:
            xs = [x if (x is None or _is_device_tensor(x)) else np.ascontiguousarray(x, dtype=np.float64) for x in x0s]
            for x, L in zip(xs, self.lens):
                if x is not None and not _is_device_tensor(x) and x.shape != (L,):
                    raise ValueError("x0 length mismatch")
            xp = (c_double_p * n)(*[None if x is None else (_device_ptr(x, L, "x0") if _is_device_tensor(x) else x.ctypes.data_as(c_double_p))
                                    for x, L in zip(xs, self.lens)])
        sp = None
        if seeds is not None:
            sp = (ctypes.c_uint64 * n)(*[int(s) & (2 ** 64 - 1) for s in seeds])
        self._keep = (ms, xs)
        check(_lib.lib().gomel_batch_upload(self._b, mp, xp, sp), "gomel_batch_upload")
        self._keep = None

    def run(self):
        check(_lib.lib().gomel_batch_run(self._b), "gomel_batch_run")

    def run_timed(self):
        t = GomelTiming()
        check(_lib.lib().gomel_batch_run_timed(self._b, ctypes.byref(t)), "gomel_batch_run_timed")
        return {"total_ms": t.total_ms, "iter_ms": t.iter_ms, "dom_ms": t.dom_ms, "iter_launches": t.iter_launches,
                "fast_path": t.fast_path, "frames": t.frames, "dom64_ms": t.dom64_ms, "iter64_launches": t.iter64_launches,
                "dom_kernel": ("k_gl_frames", "k_gl_fused", "k_gl_persist", "k_gl_fused_any")[t.fast_path]}

    def sync(self):
        check(_lib.lib().gomel_batch_sync(self._b), "gomel_batch_sync")

    def download(self, outs=None):
        """Copy the waveforms back.  `outs` may be a list of preallocated float64 arrays (caller-owned buffers, as the Go
        side would pass) or torch CUDA tensors (the audio stays in HBM); otherwise fresh arrays are returned."""
        n = len(self.n_pairs)
        if outs is None:
            outs = [np.empty(L, dtype=np.float64) for L in self.lens]
        elif len(outs) != n or any((not _is_device_tensor(o)) and (o.dtype != np.float64 or o.shape != (L,) or not o.flags.c_contiguous)
                                   for o, L in zip(outs, self.lens)):
            raise ValueError("outs must be C-contiguous float64 arrays (or CUDA tensors) of the planned lengths")
        op = (c_double_p * n)(*[_device_ptr(o, L, "out") if _is_device_tensor(o) else o.ctypes.data_as(c_double_p)
                                for o, L in zip(outs, self.lens)])
        caps = (ctypes.c_size_t * n)(*self.lens)
        check(_lib.lib().gomel_batch_download(self._b, op, caps), "gomel_batch_download")
        return outs

    def close(self):
        if self._b:
            _lib.lib().gomel_batch_destroy(self._b)
            self._b = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Mel:
    """`type Mel` (mel/mel.go:10-27).  Fields are plain attributes and may be changed between calls, as in Go."""

    def __init__(self):
        # NewMel defaults, mel/mel.go:30-41
        self.NumMels = 160
        self.MelFmin = 0.0
        self.MelFmax = 8000.0
        self.TuneMul = 1.0
        self.TuneAdd = 0.0
        self.Window = 256
        self.Resolut = 2048
        self.YReverse = False
        self.GriffinLimIterations = 2
        self.VolumeBoost = 0.0
        self.SampleRate = 0
        # not in the reference: which GPU, and which Hann definition (see DESIGN.md "window")
        self.Device = -1
        self.HannPeriodic = False
        # arithmetic of the iterations (gomel_set_precision): 0 = AUTOMATIC (default): the calibrated fp64-then-fp32 schedule, within
        # 1e-4 RMS of the reference's float64 loop (mel/mel.go:76-139); 32 = fp32 throughout, the opt-in fast mode; 64 = fp64 throughout
        self.Precision = 0
        self.Fp64Iterations = None    # an int: explicit schedule -- that many iterations in fp64, the rest fp32 (gomel_set_precision_schedule); overrides Precision
        self._engines = {}

    def _config(self):
        c = GomelConfig()
        c.num_mels, c.window, c.resolut = int(self.NumMels), int(self.Window), int(self.Resolut)
        c.griffin_lim_iterations = int(self.GriffinLimIterations)
        c.mel_fmin, c.mel_fmax = float(self.MelFmin), float(self.MelFmax)
        c.tune_mul, c.tune_add = float(self.TuneMul), float(self.TuneAdd)
        c.device, c.hann_periodic = int(self.Device), int(bool(self.HannPeriodic))
        return c

    def engine(self):
        c = self._config()
        key = (c.num_mels, c.window, c.resolut, c.griffin_lim_iterations, c.mel_fmin, c.mel_fmax, c.tune_mul, c.tune_add,
               c.device, c.hann_periodic)
        e = self._engines.get(key)
        if e is None:
            for old in self._engines.values():
                old.close()           # one live handle per Mel object: the one being replaced is released now, not at GC time
            e = Engine(c)
            self._engines = {key: e}
            e._precision = (0, None)
        want = (int(self.Precision), None if self.Fp64Iterations is None else int(self.Fp64Iterations))
        if e._precision != want:
            if want[1] is None:
                e.set_precision(want[0])
            else:
                e.set_precision_schedule(want[1])
            e._precision = want
        return e

    # mel/mel.go:46-74
    def ToMel(self, buf):
        return self.engine().to_mel(buf)

    # mel/mel.go:142-152.  `x0`/`seed` stand in for Go's global math/rand source (mel/mel.go:81-83).
    def FromMel(self, ospectrum, x0=None, seed=None):
        if not (isinstance(ospectrum, np.ndarray) and ospectrum.dtype == np.float64 and ospectrum.flags.c_contiguous):
            raise TypeError("FromMel wants a C-contiguous float64 [n,2] array: it is exp'd in place like the Go slice")
        if seed is None:
            seed = random.getrandbits(64)
        wave = self.engine().from_mel(ospectrum, x0=x0, seed=seed)
        np.exp(ospectrum, out=ospectrum)  # spectral_denormalize side effect on the caller's slice (mel/mel.go:143)
        return wave

    def ToMelBatch(self, bufs):
        return self.engine().to_mel_batch(bufs)

    def FromMelBatch(self, spectra, x0s=None, seeds=None):
        outs = self.engine().from_mel_batch(spectra, x0s, seeds)
        for s in spectra:
            np.exp(s, out=s)
        return outs

    # mel/mel.go:171-173
    def Image(self, buf):
        from . import codec
        return codec.dumpbuffer(buf, self.NumMels)

    # mel/mel.go:194-209
    def ToMelWav(self, inputFile, outputFile):
        from . import codec
        buf, sr = codec.loadwav(inputFile)
        if len(buf) == 0:
            raise ErrFileNotLoaded()
        osp = self.ToMel(buf)
        codec.dumpimage(outputFile, osp, self.NumMels, self.YReverse, float(len(buf) * self.NumMels) / float(len(osp)), float(sr))

    # mel/mel.go:176-192
    def ToMelFlac(self, inputFile, outputFile):
        from . import codec
        buf, sr = codec.loadflac(inputFile)
        if len(buf) == 0:
            raise ErrFileNotLoaded()
        osp = self.ToMel(buf)
        codec.dumpimage(outputFile, osp, self.NumMels, self.YReverse, float(len(buf) * self.NumMels) / float(len(osp)), float(sr))

    # mel/mel.go:211-238
    def ToWavPng(self, inputFile, outputFile, x0=None, seed=None):
        from . import codec
        buf, samples, samplerate = codec.loadpng(inputFile, self.YReverse)
        if len(buf) == 0:
            raise ErrFileNotLoaded()
        buf += self.VolumeBoost  # mel.go:218-221, log domain, in place
        owave = self.FromMel(buf, x0=x0, seed=seed)
        if int(samples) > 0 and _lib.lib().gomel_is_padded(int(samples), len(owave), int(self.Window)) and len(owave) > int(samples):
            owave = owave[:int(samples)]
        if samplerate != 0 and self.SampleRate == 0:
            self.SampleRate = int(samplerate)
        codec.dumpwav(outputFile, owave, self.SampleRate)  # the reference drops this error (mel.go:235)
        return None


class BoundBuffers:
    """Caller-owned spectrogram / waveform (/ x0) arrays with the pointer arrays gomel_from_mel_batch takes, built once
    (Engine.bind_buffers).  Holds references to the arrays: they cannot go away, or be resized, while bound."""

    def __init__(self, engine, mels, outs, x0s=None):
        ms = list(mels)
        if any(not isinstance(m, np.ndarray) or m.dtype != np.float64 or m.ndim != 2 or m.shape[1] != 2 or not m.flags.c_contiguous for m in ms):
            raise ValueError("mels must be C-contiguous float64 [n, 2] arrays (they are bound, not copied)")
        K, N, hop = engine.cfg.num_mels, engine.cfg.resolut, engine.cfg.window
        lens = [N + (len(m) // K - 1) * hop if len(m) and len(m) % K == 0 else 0 for m in ms]
        os_ = list(outs)
        if len(os_) != len(ms) or any(not isinstance(o, np.ndarray) or o.dtype != np.float64 or o.ndim != 1 or len(o) < L or not o.flags.c_contiguous
                                      for o, L in zip(os_, lens)):
            raise ValueError("outs must be C-contiguous float64 arrays of at least the output lengths")
        xs = None
        if x0s is not None:
            xs = list(x0s)
            if len(xs) != len(ms) or any(x is not None and (not isinstance(x, np.ndarray) or x.dtype != np.float64 or x.shape != (L,) or not x.flags.c_contiguous)
                                         for x, L in zip(xs, lens)):
                raise ValueError("x0s must be C-contiguous float64 arrays of the output lengths (or None)")
        ptrs = lambda arrs: np.array([0 if a is None else a.__array_interface__["data"][0] for a in arrs], dtype=np.uintp)
        self.engine, self.n = engine, len(ms)
        self._keep = (ms, os_, xs)
        self.mp, self.op = ptrs(ms), ptrs(os_)
        self.xp = None if xs is None else ptrs(xs)
        self.npairs = np.array([len(m) for m in ms], dtype=np.uintp)
        self.caps = np.array([len(o) for o in os_], dtype=np.uintp)
        self.views = [o[:L] for o, L in zip(os_, lens)]


class _PinnedOwner:
    def __init__(self, ptr):
        self.ptr = ptr

    def __del__(self):
        try:
            if self.ptr:
                _lib.lib().gomel_host_free(ctypes.c_void_p(self.ptr))
                self.ptr = None
        except Exception:
            pass


def host_alloc(shape, dtype=np.float64):
    """A numpy array in pinned host memory (gomel_host_alloc): buffers a bulk caller reuses across gomel_from_mel_batch calls, so
    that no call pays hipHostRegister.  Freed with the array."""
    dt = np.dtype(dtype)
    n = int(np.prod(shape)) if np.ndim(shape) else int(shape)
    ptr = _lib.lib().gomel_host_alloc(max(n, 1) * dt.itemsize)
    if not ptr:
        raise MemoryError("gomel_host_alloc(%d bytes) failed" % (n * dt.itemsize))
    owner = _PinnedOwner(ptr)
    buf = (ctypes.c_char * (max(n, 1) * dt.itemsize)).from_address(ptr)
    buf._owner = owner                       # the ctypes view keeps the allocation alive; numpy keeps the view alive
    a = np.frombuffer(buf, dtype=dt, count=n).reshape(shape)
    return a


def from_mel_batch_multi(engines, mels, x0s=None, seeds=None, outs=None):
    """gomel_from_mel_batch_multi: one call over several handles (one per GPU): contiguous utterance shards balanced by frame
    count, one host thread per device inside the library, no collective."""
    e0 = engines[0]
    ms = [_as_pairs(m) for m in mels]
    n = len(ms)
    lens = [e0.out_len(len(m)) for m in ms]
    if outs is None:
        outs = [np.empty(max(L, 1), dtype=np.float64) for L in lens]
    xs = None if x0s is None else [None if x is None else np.ascontiguousarray(x, dtype=np.float64) for x in x0s]
    dpp = ctypes.POINTER(c_double_p)
    hp = (ctypes.c_void_p * len(engines))(*[e._h for e in engines])
    mp = (c_double_p * n)(*[m.ctypes.data_as(c_double_p) for m in ms])
    op = (c_double_p * n)(*[o.ctypes.data_as(c_double_p) for o in outs])
    xp = None if xs is None else (c_double_p * n)(*[None if x is None else x.ctypes.data_as(c_double_p) for x in xs])
    sp = None if seeds is None else (ctypes.c_uint64 * n)(*[int(s) & (2 ** 64 - 1) for s in seeds])
    npairs = (ctypes.c_size_t * n)(*[len(m) for m in ms])
    caps = (ctypes.c_size_t * n)(*[len(o) for o in outs])
    check(_lib.lib().gomel_from_mel_batch_multi(hp, len(engines), n, ctypes.cast(mp, dpp), npairs, None if xp is None else ctypes.cast(xp, dpp), sp,
                                                ctypes.cast(op, dpp), caps), "gomel_from_mel_batch_multi")
    return [o[:L] for o, L in zip(outs, lens)]


def build_tables(cfg):
    """gomel_tables_build: the table blob (Hann, undomel rows, domel rows) from a configuration, on the host alone -- no GPU
    is touched.  The same bytes `Engine.export_tables()` returns for a handle of that configuration."""
    L = _lib.lib()
    n = L.gomel_tables_bytes_for(ctypes.byref(cfg))
    buf = ctypes.create_string_buffer(max(n, 1))
    check(L.gomel_tables_build(ctypes.byref(cfg), buf, n), "gomel_tables_build")
    return bytes(buf.raw[:n])


def ISTFT(s, spectrogram, numIterations, x0=None, seed=None, device=-1):
    """`func ISTFT(s *stft.STFT, spectrogram [][]complex128, numIterations int) []float64`, mel/mel.go:76-139.

    `s` is anything with the two fields of gossp's STFT the reference reads: `FrameShift` (hop) and `Window` (frameLen
    doubles).  `spectrogram` is [F][frameLen] complex (or real magnitudes); as in the reference only |spectrogram[f][j]|,
    j <= frameLen/2, matter.  The initial signal is `x0`, or the documented SplitMix64 stream of `seed`, or (neither given)
    Python's `random`, where the reference draws from math/rand.  As in the reference (mel/mel.go:98-108) a complex128 numpy
    `spectrogram` is overwritten with the last iteration's phase estimate on its own magnitudes (lower half + conjugate fill);
    other inputs (lists, real arrays) cannot be written to and are left alone."""
    spec = np.asarray(spectrogram)
    if spec.ndim != 2 or spec.shape[0] == 0:
        raise GomelError(4, "ISTFT")          # the reference panics on spectrogram[0] (mel/mel.go:78)
    N = spec.shape[1]
    window = np.ascontiguousarray(s.Window, dtype=np.float64)
    if window.shape != (N,):
        raise ValueError("len(s.Window) must equal the frame length %d" % N)
    cfg = GomelConfig()
    check(_lib.lib().gomel_default_config(ctypes.byref(cfg)), "gomel_default_config")
    cfg.resolut, cfg.window, cfg.device = N, int(s.FrameShift), int(device)
    e = Engine(cfg)
    try:
        e.set_window(window)
        if x0 is None and seed is None:
            seed = random.getrandbits(64)
        writable = isinstance(spectrogram, np.ndarray) and spectrogram.dtype == np.complex128 and spectrogram.flags.writeable
        own = N // 2 + 1 + (N & 1)                                         # bins the conjugate fill (mel/mel.go:105-108) leaves alone
        if not writable or int(numIterations) == 0:
            return e.istft(np.abs(spec[:, :own]), numIterations, x0=x0, seed=seed or 0)
        wave, sp = e.istft(np.abs(spec[:, :own]), numIterations, x0=x0, seed=seed or 0, spec=True)
        spectrogram[:, :own] = sp                                          # mel/mel.go:98-102, the bins with their own magnitude
        j = np.arange(1, N // 2)
        spectrogram[:, N - j] = np.conj(spectrogram[:, j])                 # mel/mel.go:105-108
        return wave
    finally:
        e.close()


def NewMel():
    """mel/mel.go:30-41."""
    return Mel()


def LoadWav(inputFile):
    """mel/mel.go:161-164."""
    from . import codec
    return codec.loadwav(inputFile)[0]


def LoadFlac(inputFile):
    """mel/mel.go:155-158."""
    from . import codec
    return codec.loadflac(inputFile)[0]


def SaveWav(outputFile, vec, sr):
    """mel/mel.go:167-169."""
    from . import codec
    return codec.dumpwav(outputFile, vec, sr)
