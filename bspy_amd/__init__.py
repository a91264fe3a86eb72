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
from .device_spline import DeviceSpline, MultiDeviceSpline, bspline_values_batch, get_device, set_device
from .device_spline import tessellate as tessellate_tables
from .spline import Spline
from .collocation import collocation_matrix
from .spline_block import SplineBlock


def tessellate(splines, u, v, normals=True, normalize=True):
    """Positions (and unit normals) of a batch of ``Spline`` surface patches on the grid ``u x v`` in one
    launch: ``(len(splines), 3, len(u), len(v))`` arrays (NumPy in -> NumPy out, CUDA tensors in -> tensors
    out).  The patches must share orders, nCoef and knots (e.g. the 32 patches of the Utah teapot); each
    patch's ``metadata['negateNormal']`` must agree.  See ``bspy_amd.device_spline.tessellate``."""
    from . import _spline_evaluation as _se
    splines = list(splines)
    neg = {bool(getattr(s, 'metadata', {}).get('negateNormal', False)) for s in splines}
    if len(neg) > 1:
        raise ValueError('the patches of a batch must agree on negateNormal')
    dev = u.device.index if type(u).__module__.split('.')[0] == 'torch' else None
    tables = [_se.device_tables(s, dev) for s in splines]
    return tessellate_tables(tables, (u, v), normals=normals, normalize=normalize, negate=neg.pop() if neg else False)

__all__ = ["Spline", "SplineBlock", "DeviceSpline", "MultiDeviceSpline", "tessellate", "collocation_matrix", "bspline_values_batch", "set_device", "get_device",
           "BskError", "DomainError", "NativeLibraryError"]
__version__ = "0.1.0"
