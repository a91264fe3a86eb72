"""
ctypes binding of ``libbspy_amd.so`` (C ABI: ``include/bspy_amd.h``).

This is the only way the package computes anything: there is no CPU fallback.
If the shared library is missing or cannot be loaded, every compute entry point
raises ``NativeLibraryError`` with build instructions.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# BSPY_AMD_LIB: alternative build of the same library (e.g. the -DBSK_ABLATE timing build)
LIB_PATH = os.environ.get("BSPY_AMD_LIB") or os.path.join(_HERE, "csrc", "libbspy_amd.so")

BSK_F32, BSK_F64 = 0, 1
BSK_HOST, BSK_DEVICE = 0, 1
BSK_OK, BSK_ERR_INVALID, BSK_ERR_DOMAIN, BSK_ERR_HIP, BSK_ERR_NO_DEVICE, BSK_ERR_UNSUPPORTED = range(6)
BSK_MAX_NIND, BSK_MAX_ORDER = 8, 16

# every symbol of the product ABI in include/bspy_amd.h (tests check the library exports them all) ...
PRODUCT_SYMBOLS = (
    "bsk_version", "bsk_last_error", "bsk_device_count",
    "bsk_spline_create", "bsk_spline_update", "bsk_spline_destroy",
    "bsk_evaluate", "bsk_jacobian", "bsk_normal", "bsk_curvature", "bsk_evaluate_grid", "bsk_tessellate",
    "bsk_domain_status", "bsk_bspline_values", "bsk_last_kernel",
    "bsk_multi_create", "bsk_multi_destroy", "bsk_multi_shard_plan", "bsk_multi_stream", "bsk_multi_evaluate",
    "bsk_multi_jacobian",
)
# ... and the measurement hooks of its BSK_INTERNAL section (bench.py, tools/: not used by the product path)
INTERNAL_SYMBOLS = ("bsk_debug_probe", "bsk_debug_stage_times")
SYMBOLS = PRODUCT_SYMBOLS + INTERNAL_SYMBOLS


class NativeLibraryError(RuntimeError):
    """libbspy_amd.so is missing or failed to load."""


class BskError(RuntimeError):
    """A libbspy_amd call failed (status + library message)."""

    def __init__(self, status, message):
        super().__init__(f"libbspy_amd status {status}: {message}")
        self.status = status


class DomainError(Exception):
    """A parameter lies outside the spline's domain; ``index`` = flat index of the first offender."""

    def __init__(self, index):
        super().__init__(f"parameter outside domain at flat index {index}")
        self.index = int(index)


_lib = None

_vp = ctypes.c_void_p
_vpp = ctypes.POINTER(ctypes.c_void_p)
_ip = ctypes.POINTER(ctypes.c_int)
_i64 = ctypes.c_int64
_i64p = ctypes.POINTER(ctypes.c_int64)
_i32p = ctypes.POINTER(ctypes.c_int32)


def lib():
    """Load the shared library once; fail loudly when it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            f"{LIB_PATH} not found. Build it with `make -C {os.path.dirname(LIB_PATH)}` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). bspy_amd has no CPU fallback.")
    if not os.environ.get("BSPY_AMD_NO_TORCH"):
        # PyTorch-ROCm ships its own libamdhip64; if it is going to be used in this process it
        # must be the first HIP runtime loaded (a second copy finds "no HIP GPUs"), so let it
        # load before libbspy_amd.so resolves the same SONAME.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    try:
        L = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover - depends on the machine
        raise NativeLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    L.bsk_version.restype = ctypes.c_int
    L.bsk_last_error.restype = ctypes.c_char_p
    L.bsk_device_count.argtypes = [_ip]
    L.bsk_spline_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _ip, _ip, _vpp, _vp,
                                    ctypes.POINTER(_vp)]
    L.bsk_spline_update.argtypes = [_vp, _vpp, _vp]
    L.bsk_spline_destroy.argtypes = [_vp]
    L.bsk_evaluate.argtypes = [_vp, _ip, _vpp, _i64, ctypes.c_int, _vp, _vp, _i64p]
    L.bsk_jacobian.argtypes = [_vp, _vpp, _i64, ctypes.c_int, _vp, _vp, _i64p]
    L.bsk_normal.argtypes = [_vp, _vpp, _i64, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, _vp, _i64p]
    L.bsk_curvature.argtypes = [_vp, _vpp, _i64, ctypes.c_int, _vp, _vp, _i64p]
    L.bsk_evaluate_grid.argtypes = [_vp, _ip, _vpp, _i64p, ctypes.c_int, _vp, _vp, _i64p]
    L.bsk_tessellate.argtypes = [_vpp, ctypes.c_int, _vpp, _i64p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, _vp, _vp,
                                 _i64p]
    L.bsk_domain_status.argtypes = [_vp, _vp, _i64p]
    L.bsk_bspline_values.argtypes = [ctypes.c_int, ctypes.c_int, _vp, ctypes.c_int, ctypes.c_int, _vp, _i64,
                                     ctypes.c_int, ctypes.c_int, _i32p, _i32p, _vp]
    L.bsk_multi_create.argtypes = [ctypes.c_int, ctypes.c_int, _ip, ctypes.c_int, ctypes.c_int, _ip, _ip, _vpp, _vp,
                                   ctypes.POINTER(_vp)]
    L.bsk_multi_destroy.argtypes = [_vp]
    L.bsk_multi_shard_plan.argtypes = [_vp, _i64, _i64p]
    L.bsk_multi_stream.argtypes = [_vp, ctypes.c_int, ctypes.POINTER(_vp)]
    L.bsk_multi_evaluate.argtypes = [_vp, _ip, _vpp, _i64, ctypes.c_int, _vpp, ctypes.c_int, _i64p]
    L.bsk_multi_jacobian.argtypes = [_vp, _vpp, _i64, ctypes.c_int, _vpp, ctypes.c_int, _i64p]
    L.bsk_last_kernel.argtypes = [_vp]
    L.bsk_last_kernel.restype = ctypes.c_char_p
    L.bsk_debug_probe.argtypes = [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _i64, _vp, _vp, _i64, _vp, _vp]
    L.bsk_debug_stage_times.argtypes = [_vp, ctypes.c_int, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_char_p),
                                        ctypes.c_int, _ip]
    for name in SYMBOLS:
        if name not in ("bsk_version", "bsk_last_error", "bsk_last_kernel"):
            getattr(L, name).restype = ctypes.c_int
    _lib = L
    return L


def check(status, first_bad=None):
    if status == BSK_OK:
        return
    if status == BSK_ERR_DOMAIN:
        raise DomainError(first_bad.value if first_bad is not None else -1)
    raise BskError(status, lib().bsk_last_error().decode("utf-8", "replace"))


def device_count():
    n = ctypes.c_int(0)
    st = lib().bsk_device_count(ctypes.byref(n))
    if st != BSK_OK:
        return 0
    return n.value


def dtype_code(dtype):
    return BSK_F32 if np.dtype(dtype) == np.float32 else BSK_F64


def ptr_array(pointers):
    arr = (ctypes.c_void_p * max(len(pointers), 1))()
    for i, p in enumerate(pointers):
        arr[i] = p
    return arr


def int_array(values):
    return (ctypes.c_int * max(len(values), 1))(*[int(v) for v in values])
