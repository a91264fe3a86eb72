"""
bspy_amd - MI355X-native batched B-spline evaluation, a drop-in for the evaluation path
of ericbrec/BSpy (``Spline.bspline_values`` de Boor recursion + ``Spline.evaluate`` /
``derivative`` / ``jacobian`` tensor-product contraction).

    from bspy_amd import Spline
    s = Spline(2, 3, (4, 4), (64, 64), (ku, kv), coefs)
    x, y, z = s(u, v)                       # NumPy in -> NumPy out (through the GPU)
    x, y, z = s(u_cuda, v_cuda)             # torch CUDA tensors in -> results stay on the GPU

Layers: ``Spline`` (reference API) -> ``_spline_evaluation`` (reference module functions)
-> ``DeviceSpline`` (device tables + raw batched calls) -> ``_native`` (ctypes) ->
``csrc/libbspy_amd.so`` (HIP kernels for gfx950, C ABI in include/bspy_amd.h).
There is no CPU compute path: without the shared library every call raises.
"""
from . import _native
from ._native import BskError, DomainError, NativeLibraryError
from .device_spline import DeviceSpline, bspline_values_batch, get_device, set_device
from .spline import Spline
from .collocation import collocation_matrix
from .spline_block import SplineBlock

__all__ = ["Spline", "SplineBlock", "DeviceSpline", "collocation_matrix", "bspline_values_batch", "set_device", "get_device",
           "BskError", "DomainError", "NativeLibraryError"]
__version__ = "0.1.0"
