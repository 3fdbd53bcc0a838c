"""ctypes binding of include/gomel_hip.h.  There is no CPU fallback: if libgomel_hip.so is missing or a call
fails, this raises."""
import ctypes
import importlib.util
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GOMEL_LIB") or os.path.join(HERE, "libgomel_hip.so")   # GOMEL_LIB: another build of the library (A/B timing, sanitizer builds)

c_double_p = ctypes.POINTER(ctypes.c_double)
c_float_p = ctypes.POINTER(ctypes.c_float)
c_size_p = ctypes.POINTER(ctypes.c_size_t)
c_u64_p = ctypes.POINTER(ctypes.c_uint64)


class GomelConfig(ctypes.Structure):
    """struct gomel_config (mirrors `type Mel`, mel/mel.go:10-27)."""
    _fields_ = [("num_mels", ctypes.c_int32), ("window", ctypes.c_int32), ("resolut", ctypes.c_int32),
                ("griffin_lim_iterations", ctypes.c_int32), ("mel_fmin", ctypes.c_double), ("mel_fmax", ctypes.c_double),
                ("tune_mul", ctypes.c_double), ("tune_add", ctypes.c_double), ("device", ctypes.c_int32),
                ("hann_periodic", ctypes.c_int32)]


class GomelTiming(ctypes.Structure):
    _fields_ = [("total_ms", ctypes.c_float), ("iter_ms", ctypes.c_float), ("dom_ms", ctypes.c_float), ("iter_launches", ctypes.c_int32),
                ("fast_path", ctypes.c_int32), ("frames", ctypes.c_int64), ("dom64_ms", ctypes.c_float), ("iter64_launches", ctypes.c_int32)]


# every symbol include/gomel_hip.h declares: (restype, argtypes)
_vp = ctypes.c_void_p
_cfg_p = ctypes.POINTER(GomelConfig)
SYMBOLS = {
    "gomel_default_config": (ctypes.c_int, [_cfg_p]),
    "gomel_create": (ctypes.c_int, [_cfg_p, ctypes.POINTER(_vp)]),
    "gomel_create_from_tables": (ctypes.c_int, [_cfg_p, _vp, ctypes.c_size_t, ctypes.POINTER(_vp)]),
    "gomel_tables_bytes": (ctypes.c_size_t, [_vp]),
    "gomel_tables_export": (ctypes.c_int, [_vp, _vp, ctypes.c_size_t]),
    "gomel_tables_bytes_for": (ctypes.c_size_t, [_cfg_p]),
    "gomel_tables_build": (ctypes.c_int, [_cfg_p, _vp, ctypes.c_size_t]),
    "gomel_set_window": (ctypes.c_int, [_vp, c_double_p, ctypes.c_size_t]),
    "gomel_get_window": (ctypes.c_int, [_vp, c_double_p, ctypes.c_size_t]),
    "gomel_destroy": (ctypes.c_int, [_vp]),
    "gomel_out_len": (ctypes.c_size_t, [_cfg_p, ctypes.c_size_t]),
    "gomel_from_mel": (ctypes.c_int, [_vp, c_double_p, ctypes.c_size_t, c_double_p, ctypes.c_uint64, c_double_p,
                                      ctypes.c_size_t, c_size_p]),
    "gomel_istft_row_len": (ctypes.c_size_t, [_cfg_p]),
    "gomel_istft": (ctypes.c_int, [_vp, c_double_p, ctypes.c_size_t, ctypes.c_int32, c_double_p, ctypes.c_uint64, c_double_p,
                                   ctypes.c_size_t, c_size_p]),
    "gomel_istft_spec": (ctypes.c_int, [_vp, c_double_p, ctypes.c_size_t, ctypes.c_int32, c_double_p, ctypes.c_uint64, c_double_p,
                                        ctypes.c_size_t, c_size_p, c_double_p]),
    "gomel_from_mel_batch_f32": (ctypes.c_int, [_vp, ctypes.c_size_t, ctypes.POINTER(c_float_p), c_size_p,
                                                ctypes.POINTER(c_float_p), c_u64_p, ctypes.POINTER(c_float_p), c_size_p]),
    "gomel_from_mel_batch_multi": (ctypes.c_int, [ctypes.POINTER(_vp), ctypes.c_size_t, ctypes.c_size_t, ctypes.POINTER(c_double_p), c_size_p,
                                                  ctypes.POINTER(c_double_p), c_u64_p, ctypes.POINTER(c_double_p), c_size_p]),
    "gomel_from_mel_batch_submit": (ctypes.c_int, [_vp, ctypes.c_size_t, ctypes.POINTER(c_double_p), c_size_p, ctypes.POINTER(c_double_p), c_u64_p,
                                                   ctypes.POINTER(c_double_p), c_size_p, ctypes.POINTER(_vp)]),
    "gomel_from_mel_batch_submit_f32": (ctypes.c_int, [_vp, ctypes.c_size_t, ctypes.POINTER(c_float_p), c_size_p, ctypes.POINTER(c_float_p), c_u64_p,
                                                       ctypes.POINTER(c_float_p), c_size_p, ctypes.POINTER(_vp)]),
    "gomel_job_wait": (ctypes.c_int, [_vp]),
    "gomel_job_wait_timeout": (ctypes.c_int, [_vp, ctypes.c_int64]),
    "gomel_set_call_mode": (ctypes.c_int, [_vp, ctypes.c_int]),
    "gomel_job_poll": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_int)]),
    "gomel_plan_chunks": (ctypes.c_int, [_cfg_p, ctypes.c_size_t, c_size_p, c_size_p, ctypes.c_size_t, c_size_p]),
    "gomel_plan_chunks_mode": (ctypes.c_int, [_cfg_p, ctypes.c_int32, ctypes.c_size_t, c_size_p, c_size_p, ctypes.c_size_t, c_size_p]),
    "gomel_host_alloc": (_vp, [ctypes.c_size_t]),
    "gomel_host_free": (ctypes.c_int, [_vp]),
    "gomel_host_register": (ctypes.c_int, [_vp, ctypes.c_size_t]),
    "gomel_host_unregister": (ctypes.c_int, [_vp]),
    "gomel_batch_create_f32": (ctypes.c_int, [_vp, ctypes.c_size_t, c_size_p, ctypes.POINTER(_vp)]),
    "gomel_batch_upload_f32": (ctypes.c_int, [_vp, ctypes.POINTER(c_float_p), ctypes.POINTER(c_float_p), c_u64_p]),
    "gomel_batch_download_f32": (ctypes.c_int, [_vp, ctypes.POINTER(c_float_p), c_size_p]),
    "gomel_to_mel_batch": (ctypes.c_int, [_vp, ctypes.c_size_t, ctypes.POINTER(c_double_p), c_size_p, ctypes.POINTER(c_double_p), c_size_p, c_size_p]),
    "gomel_from_mel_batch": (ctypes.c_int, [_vp, ctypes.c_size_t, ctypes.POINTER(c_double_p), c_size_p,
                                            ctypes.POINTER(c_double_p), c_u64_p, ctypes.POINTER(c_double_p), c_size_p]),
    "gomel_batch_create": (ctypes.c_int, [_vp, ctypes.c_size_t, c_size_p, ctypes.POINTER(_vp)]),
    "gomel_batch_upload": (ctypes.c_int, [_vp, ctypes.POINTER(c_double_p), ctypes.POINTER(c_double_p), c_u64_p]),
    "gomel_batch_run": (ctypes.c_int, [_vp]),
    "gomel_batch_run_timed": (ctypes.c_int, [_vp, ctypes.POINTER(GomelTiming)]),
    "gomel_batch_sync": (ctypes.c_int, [_vp]),
    "gomel_batch_download": (ctypes.c_int, [_vp, ctypes.POINTER(c_double_p), c_size_p]),
    "gomel_batch_destroy": (ctypes.c_int, [_vp]),
    "gomel_set_path": (ctypes.c_int, [_vp, ctypes.c_int]),
    "gomel_set_precision": (ctypes.c_int, [_vp, ctypes.c_int]),
    "gomel_set_precision_schedule": (ctypes.c_int, [_vp, ctypes.c_int32]),
    "gomel_auto_fp64_iterations": (ctypes.c_int32, [_cfg_p, ctypes.c_int32]),
    "gomel_get_precision": (ctypes.c_int, [_vp, ctypes.c_int32, ctypes.POINTER(ctypes.c_int32)]),
    "gomel_to_mel_pairs": (ctypes.c_size_t, [_cfg_p, ctypes.c_size_t]),
    "gomel_to_mel": (ctypes.c_int, [_vp, c_double_p, ctypes.c_size_t, c_double_p, ctypes.c_size_t, c_size_p]),
    "gomel_phase_pairs": (ctypes.c_size_t, [_cfg_p, ctypes.c_int32, ctypes.c_size_t]),
    "gomel_phase_out_len": (ctypes.c_size_t, [_cfg_p, ctypes.c_int32, ctypes.c_size_t]),
    "gomel_to_phase": (ctypes.c_int, [_vp, ctypes.c_int32, c_double_p, ctypes.c_size_t, c_double_p, ctypes.c_size_t, c_size_p]),
    "gomel_from_phase": (ctypes.c_int, [_vp, ctypes.c_int32, ctypes.c_double, c_double_p, ctypes.c_size_t, c_double_p, ctypes.c_size_t, c_size_p]),
    "gomel_pad_len": (ctypes.c_size_t, [ctypes.c_size_t, ctypes.c_int32]),
    "gomel_is_padded": (ctypes.c_int, [ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int32]),
    "gomel_rfft_frames": (ctypes.c_int, [_vp, c_float_p, ctypes.c_size_t, c_float_p]),
    "gomel_irfft_frames": (ctypes.c_int, [_vp, c_float_p, ctypes.c_size_t, c_float_p]),
    "gomel_target_magnitudes": (ctypes.c_int, [_vp, c_double_p, ctypes.c_size_t, c_float_p]),
    "gomel_strerror": (ctypes.c_char_p, [ctypes.c_int]),
    "gomel_last_hip_error": (ctypes.c_int, []),
    "gomel_abi_version": (ctypes.c_int, []),
    "gomel_numa_cpus": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32), ctypes.c_size_t, c_size_p]),
    "gomel_numa_bind_thread": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]),
    "gomel_device_count": (ctypes.c_int, []),
    "gomel_device_name": (ctypes.c_int, [_vp, ctypes.c_char_p, ctypes.c_size_t]),
    "gomel_device_synchronize": (ctypes.c_int, [ctypes.c_int32]),
}

SCHEDULE_AUTO = -2 ** 31      # GOMEL_SCHEDULE_AUTO

_lib = None


class GomelError(RuntimeError):
    def __init__(self, status, where):
        self.status = status
        msg = lib().gomel_strerror(status).decode()
        if status == 8:
            msg += " [hipError_t %d]" % lib().gomel_last_hip_error()
        super().__init__("%s: %s (status %d)" % (where, msg, status))


_hip_runtime = None


def _share_hip_runtime():
    """Which HIP runtime the library runs on.  libgomel_hip.so is linked against /opt/rocm's libamdhip64.so.7 (RUNPATH) and that is
    what it gets by default -- the runtime the product ships on, the one a Go host loads.  A PyTorch-ROCm wheel bundles its own
    libamdhip64.so with the same SONAME: when torch is ALREADY imported the dynamic loader resolves the library's DT_NEEDED to that
    loaded copy by itself (one runtime per process, device pointers and streams interchangeable: torch interop works).  The one
    order that does not work by itself is gomel_amd first, torch later -- torch would then bring a second runtime beside the system
    one and report no GPU; a Python host that wants that order sets GOMEL_HIP_RUNTIME=torch, which loads torch's copy up front."""
    global _hip_runtime
    if _hip_runtime is not None or "torch" in sys.modules or os.environ.get("GOMEL_HIP_RUNTIME", "system") != "torch":
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(path):
        try:
            _hip_runtime = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            _hip_runtime = None


def hip_runtime_path():
    """Path of the libamdhip64 mapped into this process (from /proc/self/maps), None before the library is loaded.  More than one
    copy mapped is reported as the paths joined by ' + ' -- a misconfiguration worth seeing (see _share_hip_runtime)."""
    found = []
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                path = line.rstrip("\n").split(None, 5)[-1] if line.count("/") else ""
                if "libamdhip64" in os.path.basename(path) and path not in found:
                    found.append(path)
    except OSError:
        return None
    return " + ".join(os.path.realpath(p) for p in found) if found else None


def lib():
    """Load libgomel_hip.so (built in-tree by gomel_amd.build).  Raises if it is absent: no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("gomel_amd: %s not found -- run `python -m gomel_amd.build` (there is no CPU fallback)" % LIB_PATH)
        _share_hip_runtime()
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(l, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(status, where):
    if status != 0:
        raise GomelError(status, where)
