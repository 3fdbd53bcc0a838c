"""gomel_amd -- MI355X-native Griffin-Lim mel inversion behind gomel's `mel.Mel` API.

Only what the hot path needs: `csrc/` (HIP kernels + the C ABI declared in include/gomel_hip.h), and the host-side
mirror of the reference's Go package `mel`.  The shared library is loaded lazily (first call); importing this package
on a machine without a GPU works, calling into it raises.
"""
from .mel import (ISTFT, Batch, DevicePointer, build_tables, Engine, ErrFileNotLoaded, LoadFlac, LoadWav, Mel, NewMel, SaveWav, from_mel_batch_multi, host_alloc)  # noqa: F401
from ._lib import GomelConfig, GomelError, hip_runtime_path  # noqa: F401
from .phase import NewPhase, Phase  # noqa: F401

__all__ = ["NewMel", "Mel", "ISTFT", "build_tables", "Engine", "Batch", "ErrFileNotLoaded", "LoadWav", "LoadFlac", "SaveWav", "GomelConfig", "GomelError", "NewPhase", "Phase", "from_mel_batch_multi", "host_alloc", "DevicePointer", "hip_runtime_path"]
