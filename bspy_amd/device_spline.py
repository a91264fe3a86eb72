"""
``DeviceSpline``: the device-resident tables of one spline (a ``bsk_spline`` handle)
and the raw batched calls on it.  This is the thinnest Python layer over the C ABI;
``bspy_amd.Spline`` (the drop-in for ``bspy.Spline``) is built on it.

All arrays are SoA: parameters one array per independent variable, results
``(nDep, N)`` / ``(nDep, nInd, N)``.
"""
import ctypes
import weakref

import numpy as np

from . import _native as nv
from .result_pool import default_pool

_default_device = None


def set_device(index):
    """Default GPU for new device tables (one process per GPU: set it to LOCAL_RANK)."""
    global _default_device
    _default_device = int(index)


def get_device():
    global _default_device
    if _default_device is None:
        import os
        _default_device = int(os.environ.get("BSPY_AMD_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        n = nv.device_count()
        if n > 0:
            _default_device %= n
    return _default_device


def _destroy(handle):
    try:
        nv.lib().bsk_spline_destroy(handle)
    except Exception:  # noqa: BLE001 - interpreter shutdown
        pass


def _is_torch(x):
    return type(x).__module__.startswith("torch") and hasattr(x, "data_ptr")


class DeviceSpline:
    """Knots + coefficients of one spline on one GPU.

    Parameters mirror the canonical in-memory form of the reference ``Spline``
    (bspy/spline.py:46-76): ``knots`` a sequence of ``nInd`` 1-D arrays,
    ``coefs`` of shape ``(nDep, *nCoef)``.  ``dtype`` is the arithmetic type of
    every call on this object (float32 or float64).
    """

    def __init__(self, order, nCoef, knots, coefs, dtype=np.float64, device=None):
        self.dtype = np.dtype(dtype)
        if self.dtype not in (np.dtype(np.float32), np.dtype(np.float64)):
            raise ValueError("dtype must be float32 or float64")
        self.device = get_device() if device is None else int(device)
        self.order = tuple(int(o) for o in order)
        self.nCoef = tuple(int(c) for c in nCoef)
        self.nInd = len(self.order)
        coefs = np.asarray(coefs)
        self.nDep = int(coefs.shape[0])
        self._ctype = ctypes.c_float if self.dtype == np.float32 else ctypes.c_double
        ks, cf = self._host_tables(knots, coefs)
        handle = ctypes.c_void_p()
        st = nv.lib().bsk_spline_create(nv.dtype_code(self.dtype), self.device, self.nInd, self.nDep,
                                        nv.int_array(self.order), nv.int_array(self.nCoef),
                                        nv.ptr_array([k.ctypes.data for k in ks]), cf.ctypes.data,
                                        ctypes.byref(handle))
        nv.check(st)
        self._handle = handle
        self._finalizer = weakref.finalize(self, _destroy, handle)

    def _host_tables(self, knots, coefs):
        if len(knots) != self.nInd:
            raise ValueError("len(knots) != nInd")
        ks = [np.ascontiguousarray(k, self.dtype) for k in knots]
        for k, o, c in zip(ks, self.order, self.nCoef):
            if k.shape != (o + c,):
                raise ValueError("knots array has the wrong length")
        cf = np.ascontiguousarray(coefs, self.dtype)
        if cf.shape != (self.nDep, *self.nCoef):
            raise ValueError("coefs must have shape (nDep, *nCoef)")
        return ks, cf

    def update(self, knots, coefs):
        """Re-upload knots/coefficients of the same shapes (after in-place mutation)."""
        ks, cf = self._host_tables(knots, np.asarray(coefs))
        nv.check(nv.lib().bsk_spline_update(self._handle, nv.ptr_array([k.ctypes.data for k in ks]), cf.ctypes.data))

    def close(self):
        self._finalizer()

    def last_kernel(self):
        """Family name of the kernel the most recent point call on this object launched (diagnostic)."""
        return nv.lib().bsk_last_kernel(self._handle).decode()

    # ------------------------------------------------------------------ host (NumPy) calls
    def _host_params(self, points):
        if len(points) != self.nInd:
            raise ValueError(f"Incorrect number of parameter values: {len(points)}")
        ps = [np.ascontiguousarray(p, self.dtype).ravel() for p in points]
        n = ps[0].size if ps else 0
        if any(p.size != n for p in ps):
            raise ValueError("parameter arrays must have the same size")
        return ps, n

    def point(self, uvw, wrt=None, jacobian=False):
        """One parameter point (the reference's single-point API): ``uvw`` = nInd scalars ->
        ndarray (nDep,) or, with ``jacobian``, (nDep, nInd).  Same C-ABI calls as ``evaluate`` /
        ``jacobian`` with N = 1 (the library serves such calls through a pinned, device-mapped
        buffer); only the Python-side argument handling is leaner."""
        a = np.array(uvw, dtype=self.dtype).reshape(-1)
        if a.size != self.nInd:
            raise ValueError(f"Incorrect number of parameter values: {a.size}")
        base, isz = a.ctypes.data, a.itemsize
        ptrs = (ctypes.c_void_p * max(self.nInd, 1))(*[base + i * isz for i in range(self.nInd)])
        bad = ctypes.c_int64(-1)
        if jacobian:
            out = np.empty((self.nDep, self.nInd), self.dtype)
            st = nv.lib().bsk_jacobian(self._handle, ptrs, 1, nv.BSK_HOST, out.ctypes.data, None, ctypes.byref(bad))
        else:
            out = np.empty(self.nDep, self.dtype)
            st = nv.lib().bsk_evaluate(self._handle, nv.int_array(wrt) if wrt is not None else None, ptrs, 1, nv.BSK_HOST,
                                       out.ctypes.data, None, ctypes.byref(bad))
        nv.check(st, bad)
        return out

    def _host_out(self, out, shape):
        if out is None:
            # a fresh array, as the reference returns - its memory recycled from results the caller has dropped
            # (result_pool.py: first-touch page faults are most of a large NumPy call)
            return default_pool().empty(shape, self.dtype)
        if not isinstance(out, np.ndarray) or out.dtype != self.dtype or out.shape != shape or not out.flags.c_contiguous:
            raise ValueError(f"out must be a C-contiguous {np.dtype(self.dtype).name} array of shape {shape}")
        return out

    def evaluate(self, points, wrt=None, out=None):
        """points: nInd arrays of N values -> ndarray (nDep, N).  Raises DomainError.
        ``out``: a result array to write into (a reused array saves the first-touch page faults of
        a fresh one: 10 M cfg2 points 19 -> 8 ms)."""
        ps, n = self._host_params(points)
        out = self._host_out(out, (self.nDep, n))
        bad = ctypes.c_int64(-1)
        st = nv.lib().bsk_evaluate(self._handle, nv.int_array(wrt) if wrt is not None else None,
                                   nv.ptr_array([p.ctypes.data for p in ps]), n, nv.BSK_HOST, out.ctypes.data, None,
                                   ctypes.byref(bad))
        nv.check(st, bad)
        return out

    def jacobian(self, points, out=None):
        """points: nInd arrays of N values -> ndarray (nDep, nInd, N)."""
        ps, n = self._host_params(points)
        out = self._host_out(out, (self.nDep, self.nInd, n))
        bad = ctypes.c_int64(-1)
        st = nv.lib().bsk_jacobian(self._handle, nv.ptr_array([p.ctypes.data for p in ps]), n, nv.BSK_HOST,
                                   out.ctypes.data, None, ctypes.byref(bad))
        nv.check(st, bad)
        return out

    def normal(self, points, normalize=True, negate=False):
        """points: nInd arrays of N values -> ndarray (max(nInd, nDep), N) of normals
        (|nInd - nDep| must be 1)."""
        ps, n = self._host_params(points)
        out = default_pool().empty((max(self.nInd, self.nDep), n), self.dtype)
        bad = ctypes.c_int64(-1)
        st = nv.lib().bsk_normal(self._handle, nv.ptr_array([p.ctypes.data for p in ps]), n, nv.BSK_HOST,
                                 1 if normalize else 0, 1 if negate else 0, out.ctypes.data, None, ctypes.byref(bad))
        nv.check(st, bad)
        return out

    def normal_device(self, points, normalize=True, negate=False, out=None, check=True):
        """points: nInd CUDA tensors -> CUDA tensor (max(nInd, nDep), N)."""
        torch, tdt, ps, n = self._torch_params(points)
        big = max(self.nInd, self.nDep)
        if out is None:
            out = torch.empty((big, n), dtype=tdt, device=ps[0].device)
        elif out.dtype != tdt or out.numel() != big * n or not out.is_contiguous():
            raise ValueError("out must be a contiguous (max(nInd, nDep), N) tensor of the spline's dtype")
        st = nv.lib().bsk_normal(self._handle, nv.ptr_array([p.data_ptr() for p in ps]), n, nv.BSK_DEVICE,
                                 1 if normalize else 0, 1 if negate else 0, out.data_ptr(),
                                 self._stream_ptr(torch, self.device), None)
        nv.check(st)
        if check:
            self.domain_status()
        return out

    def curvature(self, points):
        """points: nInd arrays of N values -> ndarray (N,): curvature of a curve (nDep >= 2) or
        Gaussian curvature of a surface in 3-D."""
        ps, n = self._host_params(points)
        out = default_pool().empty((n,), self.dtype)
        bad = ctypes.c_int64(-1)
        st = nv.lib().bsk_curvature(self._handle, nv.ptr_array([p.ctypes.data for p in ps]), n, nv.BSK_HOST,
                                    out.ctypes.data, None, ctypes.byref(bad))
        nv.check(st, bad)
        return out

    def curvature_device(self, points, out=None, check=True):
        """points: nInd CUDA tensors -> CUDA tensor (N,)."""
        torch, tdt, ps, n = self._torch_params(points)
        if out is None:
            out = torch.empty((n,), dtype=tdt, device=ps[0].device)
        st = nv.lib().bsk_curvature(self._handle, nv.ptr_array([p.data_ptr() for p in ps]), n, nv.BSK_DEVICE,
                                    out.data_ptr(), self._stream_ptr(torch, self.device), None)
        nv.check(st)
        if check:
            self.domain_status()
        return out

    def evaluate_grid(self, axes, wrt=None):
        """axes: nInd 1-D arrays (n_i values) -> ndarray (nDep, n_0, ..., n_{nInd-1})."""
        if len(axes) != self.nInd:
            raise ValueError(f"Incorrect number of parameter values: {len(axes)}")
        gs = [np.ascontiguousarray(a, self.dtype).ravel() for a in axes]
        shape = tuple(g.size for g in gs)
        out = default_pool().empty((self.nDep, *shape), self.dtype)
        bad = ctypes.c_int64(-1)
        ng = (ctypes.c_int64 * max(self.nInd, 1))(*shape)
        st = nv.lib().bsk_evaluate_grid(self._handle, nv.int_array(wrt) if wrt is not None else None,
                                        nv.ptr_array([g.ctypes.data for g in gs]), ng, nv.BSK_HOST, out.ctypes.data,
                                        None, ctypes.byref(bad))
        nv.check(st, bad)
        return out

    # ------------------------------------------------------------------ device (torch) calls
    def _torch_params(self, points):
        import torch
        if len(points) != self.nInd:
            raise ValueError(f"Incorrect number of parameter values: {len(points)}")
        tdt = torch.float32 if self.dtype == np.float32 else torch.float64
        ps = []
        for p in points:
            if not (_is_torch(p) and p.is_cuda):
                raise TypeError("device calls need CUDA/HIP torch tensors for every variable")
            if p.device.index != self.device:
                raise ValueError(f"tensor on device {p.device.index}, spline tables on device {self.device}")
            ps.append(p.to(tdt).contiguous().view(-1))
        n = ps[0].numel() if ps else 0
        if any(p.numel() != n for p in ps):
            raise ValueError("parameter tensors must have the same number of elements")
        return torch, tdt, ps, n

    @staticmethod
    def _stream_ptr(torch, device):
        return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)

    def evaluate_device(self, points, wrt=None, out=None, check=True):
        """points: nInd CUDA tensors -> CUDA tensor (nDep, N), enqueued on torch's current
        stream.  ``check=False`` skips the (synchronising) out-of-domain query; call
        ``domain_status()`` later."""
        torch, tdt, ps, n = self._torch_params(points)
        if out is None:
            out = torch.empty((self.nDep, n), dtype=tdt, device=ps[0].device)
        elif out.dtype != tdt or out.numel() != self.nDep * n or not out.is_contiguous():
            raise ValueError("out must be a contiguous (nDep, N) tensor of the spline's dtype")
        st = nv.lib().bsk_evaluate(self._handle, nv.int_array(wrt) if wrt is not None else None,
                                   nv.ptr_array([p.data_ptr() for p in ps]), n, nv.BSK_DEVICE, out.data_ptr(),
                                   self._stream_ptr(torch, self.device), None)
        nv.check(st)
        if check:
            self.domain_status()
        return out

    def jacobian_device(self, points, out=None, check=True):
        """points: nInd CUDA tensors -> CUDA tensor (nDep, nInd, N)."""
        torch, tdt, ps, n = self._torch_params(points)
        if out is None:
            out = torch.empty((self.nDep, self.nInd, n), dtype=tdt, device=ps[0].device)
        elif out.dtype != tdt or out.numel() != self.nDep * self.nInd * n or not out.is_contiguous():
            raise ValueError("out must be a contiguous (nDep, nInd, N) tensor of the spline's dtype")
        st = nv.lib().bsk_jacobian(self._handle, nv.ptr_array([p.data_ptr() for p in ps]), n, nv.BSK_DEVICE,
                                   out.data_ptr(), self._stream_ptr(torch, self.device), None)
        nv.check(st)
        if check:
            self.domain_status()
        return out

    def evaluate_grid_device(self, axes, wrt=None, out=None, check=True):
        """axes: nInd 1-D CUDA tensors -> CUDA tensor (nDep, n_0, ..., n_{nInd-1})."""
        import torch
        if len(axes) != self.nInd:
            raise ValueError(f"Incorrect number of parameter values: {len(axes)}")
        tdt = torch.float32 if self.dtype == np.float32 else torch.float64
        gs = [a.to(tdt).contiguous().view(-1) for a in axes]
        shape = tuple(int(g.numel()) for g in gs)
        if out is None:
            out = torch.empty((self.nDep, *shape), dtype=tdt, device=gs[0].device)
        ng = (ctypes.c_int64 * max(self.nInd, 1))(*shape)
        st = nv.lib().bsk_evaluate_grid(self._handle, nv.int_array(wrt) if wrt is not None else None,
                                        nv.ptr_array([g.data_ptr() for g in gs]), ng, nv.BSK_DEVICE, out.data_ptr(),
                                        self._stream_ptr(torch, self.device), None)
        nv.check(st)
        if check:
            self.domain_status()
        return out

    def domain_status(self):
        """Synchronise and raise DomainError if any device call since the last query met an
        out-of-domain parameter."""
        stream = None
        try:
            import torch
            if torch.cuda.is_available():
                stream = self._stream_ptr(torch, self.device)
        except ImportError:  # pragma: no cover
            pass
        bad = ctypes.c_int64(-1)
        st = nv.lib().bsk_domain_status(self._handle, stream, ctypes.byref(bad))
        nv.check(st, bad)


def tessellate(patches, axes, normals=True, normalize=True, negate=False, out=None, check=True):
    """Positions and normals of a BATCH of surface patches in 3-D on one parameter grid, one launch
    for all patches (C ABI ``bsk_tessellate``).

    patches : sequence of DeviceSpline with nInd 2, nDep 3 that share dtype, orders, nCoef and knots
    axes    : (u, v) 1-D arrays.  NumPy arrays -> NumPy results; CUDA torch tensors -> the results
              stay on the GPU (``out=(positions, normals)`` may pass preallocated tensors)
    returns : positions ``(len(patches), 3, len(u), len(v))`` and, with ``normals``, the normals
              (cross product of the partial derivatives; unit length when ``normalize``) in the same
              shape.
    """
    patches = list(patches)
    if not patches:
        raise ValueError("tessellate needs at least one patch")
    if len(axes) != 2:
        raise ValueError(f"Incorrect number of parameter values: {len(axes)}")
    first = patches[0]
    handles = (ctypes.c_void_p * len(patches))(*[p._handle for p in patches])
    bad = ctypes.c_int64(-1)
    if any(_is_torch(a) for a in axes):
        import torch
        tdt = torch.float32 if first.dtype == np.float32 else torch.float64
        gs = [a.to(tdt).contiguous().view(-1) for a in axes]
        shape = (len(patches), 3, int(gs[0].numel()), int(gs[1].numel()))
        pos, nrm = out if out is not None else (None, None)
        if pos is None:
            pos = torch.empty(shape, dtype=tdt, device=gs[0].device)
        if normals and nrm is None:
            nrm = torch.empty(shape, dtype=tdt, device=gs[0].device)
        for t in (pos, nrm) if normals else (pos,):
            if t.dtype != tdt or tuple(t.shape) != shape or not t.is_contiguous():
                raise ValueError("out tensors must be contiguous (patches, 3, n_u, n_v) tensors of the patches' dtype")
        ng = (ctypes.c_int64 * 2)(shape[2], shape[3])
        st = nv.lib().bsk_tessellate(handles, len(patches), nv.ptr_array([g.data_ptr() for g in gs]), ng, nv.BSK_DEVICE,
                                     int(bool(normalize)), int(bool(negate)), pos.data_ptr(),
                                     nrm.data_ptr() if normals else None,
                                     DeviceSpline._stream_ptr(torch, first.device), None)
        nv.check(st)
        if check:
            first.domain_status()
        return (pos, nrm) if normals else pos
    gs = [np.ascontiguousarray(a, first.dtype).ravel() for a in axes]
    shape = (len(patches), 3, gs[0].size, gs[1].size)
    pos = default_pool().empty(shape, first.dtype)
    nrm = default_pool().empty(shape, first.dtype) if normals else None
    ng = (ctypes.c_int64 * 2)(shape[2], shape[3])
    st = nv.lib().bsk_tessellate(handles, len(patches), nv.ptr_array([g.ctypes.data for g in gs]), ng, nv.BSK_HOST,
                                 int(bool(normalize)), int(bool(negate)), pos.ctypes.data,
                                 nrm.ctypes.data if normals else None, None, ctypes.byref(bad))
    nv.check(st, bad)
    return (pos, nrm) if normals else pos


def bspline_values_batch(knots, order, u, derivative_order=0, taylor_coefs=False, knot=None, device=None):
    """Batched basis values: returns (ix int32 (N,), basis (N, order)) in the knots' dtype
    (float32 stays float32, everything else is computed in float64)."""
    knots = np.asarray(knots)
    dt = np.float32 if knots.dtype == np.float32 else np.float64
    knots = np.ascontiguousarray(knots, dt)
    u = np.ascontiguousarray(u, dt).ravel()
    n = u.size
    ix = np.empty(n, np.int32)
    basis = np.empty((n, int(order)), dt)
    kin = None
    if knot is not None:
        kin_arr = np.ascontiguousarray(np.broadcast_to(np.asarray(knot, np.int32), (n,)))
        kin = kin_arr.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
    st = nv.lib().bsk_bspline_values(nv.dtype_code(dt), get_device() if device is None else int(device),
                                     knots.ctypes.data, knots.size, int(order), u.ctypes.data, n,
                                     int(derivative_order), 1 if taylor_coefs else 0, kin,
                                     ix.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), basis.ctypes.data)
    nv.check(st)
    return ix, basis


class MultiDeviceSpline:
    """One spline replicated on several GPUs of a node, driven from ONE process through the
    ``bsk_multi_*`` entry points (include/bspy_amd.h): the batch is cut into contiguous shards of
    ``ceil(N / ndev)`` points, every device evaluates its shard, and results are exchanged only on
    request (one grouped RCCL all-gather).  The one-process-per-GPU form is ``bspy_amd.sharding``."""

    def __init__(self, order, nCoef, knots, coefs, dtype=np.float64, devices=None):
        self.dtype = np.dtype(dtype)
        self.order = tuple(int(o) for o in order)
        self.nCoef = tuple(int(c) for c in nCoef)
        self.nInd = len(self.order)
        coefs = np.ascontiguousarray(coefs, self.dtype)
        self.nDep = int(coefs.shape[0])
        ks = [np.ascontiguousarray(k, self.dtype) for k in knots]
        self.devices = list(range(nv.device_count())) if devices is None else [int(d) for d in devices]
        handle = ctypes.c_void_p()
        st = nv.lib().bsk_multi_create(nv.dtype_code(self.dtype), len(self.devices), nv.int_array(self.devices), self.nInd,
                                       self.nDep, nv.int_array(self.order), nv.int_array(self.nCoef),
                                       nv.ptr_array([k.ctypes.data for k in ks]), coefs.ctypes.data, ctypes.byref(handle))
        nv.check(st)
        self._handle = handle
        self._finalizer = weakref.finalize(self, lambda h: nv.lib().bsk_multi_destroy(h), handle)

    def close(self):
        self._finalizer()

    def shard_plan(self, n):
        start = (ctypes.c_int64 * (len(self.devices) + 1))()
        nv.check(nv.lib().bsk_multi_shard_plan(self._handle, int(n), start))
        return list(start)

    def _host(self, jac, points, wrt):
        if len(points) != self.nInd:
            raise ValueError(f"Incorrect number of parameter values: {len(points)}")
        ps = [np.ascontiguousarray(p, self.dtype).ravel() for p in points]
        n = ps[0].size
        rows = self.nDep * self.nInd if jac else self.nDep
        out = default_pool().empty((rows, n), self.dtype)
        bad = ctypes.c_int64(-1)
        outs = nv.ptr_array([out.ctypes.data])
        if jac:
            st = nv.lib().bsk_multi_jacobian(self._handle, nv.ptr_array([p.ctypes.data for p in ps]), n, nv.BSK_HOST, outs, 0,
                                             ctypes.byref(bad))
        else:
            st = nv.lib().bsk_multi_evaluate(self._handle, nv.int_array(wrt) if wrt is not None else None,
                                             nv.ptr_array([p.ctypes.data for p in ps]), n, nv.BSK_HOST, outs, 0, ctypes.byref(bad))
        nv.check(st, bad)
        return out.reshape(self.nDep, self.nInd, n) if jac else out

    def evaluate(self, points, wrt=None):
        """Host arrays in, host (nDep, N) out; every device moves its shard over its own PCIe link."""
        return self._host(False, points, wrt)

    def jacobian(self, points):
        return self._host(True, points, None)

    def evaluate_device(self, shards, n, wrt=None, gather=False, jacobian=False):
        """``shards[d]`` = the nInd CUDA tensors of device slot d's shard (on that device), ``n`` the
        global point count.  Returns one tensor per device: the compact ``(rows, shard)`` block, or with
        ``gather`` the whole ``(rows, ndev * ceil(n / ndev))`` result on every device (columns >= n are
        padding)."""
        import torch
        ndev = len(self.devices)
        plan = self.shard_plan(n)
        tdt = torch.float32 if self.dtype == np.float32 else torch.float64
        rows = self.nDep * self.nInd if jacobian else self.nDep
        chunk = -(-int(n) // ndev) if n > 0 else 0
        flat, outs = [], []
        for d in range(ndev):
            cnt = plan[d + 1] - plan[d]
            for t in shards[d]:
                if t.numel() != cnt or t.dtype != tdt or not t.is_contiguous() or t.device.index != self.devices[d]:
                    raise ValueError(f"shard {d}: need {cnt} contiguous {tdt} values on device {self.devices[d]}")
                flat.append(t.data_ptr())
            outs.append(torch.empty((rows, ndev * chunk if gather else cnt), dtype=tdt, device=torch.device("cuda", self.devices[d])))
        torch.cuda.synchronize()              # the library's streams are not torch's
        bad = ctypes.c_int64(-1)
        optrs = nv.ptr_array([o.data_ptr() for o in outs])
        if jacobian:
            st = nv.lib().bsk_multi_jacobian(self._handle, nv.ptr_array(flat), int(n), nv.BSK_DEVICE, optrs, 1 if gather else 0,
                                             ctypes.byref(bad))
        else:
            st = nv.lib().bsk_multi_evaluate(self._handle, nv.int_array(wrt) if wrt is not None else None, nv.ptr_array(flat),
                                             int(n), nv.BSK_DEVICE, optrs, 1 if gather else 0, ctypes.byref(bad))
        nv.check(st, bad)
        return outs
