"""
Module-function form of the evaluation path, mirroring ``bspy/_spline_evaluation.py``
of the reference (same names, argument meaning and error behaviour) so callers that
bind to the module functions (``bspy/spline_block.py:197,224``) can bind to these:

    bspline_values(knot, knots, splineOrder, u, derivativeOrder=0, taylorCoefs=False)   :4-27
    derivative(self, with_respect_to, uvw)                                              :109-133
    domain(self)                                                                        :135-138
    evaluate(self, uvw)                                                                 :140-164
    jacobian(self, uvw)                                                                 :205-213

``self`` is any object with the reference ``Spline``'s attributes (``nInd, nDep, order,
nCoef, knots, coefs``) - a ``bspy_amd.Spline`` or the reference's own ``Spline``.

Every function computes on the GPU through libbspy_amd.so (span search, Cox-de Boor
recursion and window contraction are HIP kernels); there is no CPU path.  Single-point
calls are batches of one.  ``evaluate_batch`` / ``jacobian_batch`` are the batched
entry points behind ``Spline.evaluate`` / ``Spline.derivative`` (reference:
``np.frompyfunc`` loop, bspy/spline.py:757-770, :936-949).
"""

import numpy as np

from . import _native as nv
from .device_spline import DeviceSpline, _is_torch, bspline_values_batch, get_device


def compute_dtype(self):
    """float32 only when every knot vector AND the coefficients are float32 (the
    reference's all-fp32 path, e.g. examples/teapot.py); otherwise float64.  Mixed
    fp32/fp64 inputs are computed in fp64, which is at least as accurate as the
    reference's fp32 basis (bspy/_spline_evaluation.py:5)."""
    if self.coefs.dtype == np.float32 and all(np.asarray(k).dtype == np.float32 for k in self.knots):
        return np.dtype(np.float32)
    return np.dtype(np.float64)


def _bits(arr):
    """The array's bytes as unsigned integers of its own item size (NaN payloads and signed zeros
    compare like any other bit pattern)."""
    a = np.ascontiguousarray(arr)
    return a.reshape(-1).view(f"u{a.dtype.itemsize}") if a.dtype.itemsize in (4, 8) else a.reshape(-1).view(np.uint8)


def _layout(self):
    return (self.coefs.shape, str(self.coefs.dtype), tuple((len(k), str(np.asarray(k).dtype)) for k in self.knots))


def _unchanged(snapshot, self):
    """Exact comparison of knots and coefficients with the private copies taken at upload time (a
    96 KB table compares in ~6 us; a checksum cheap enough to run on every call would be a linear
    one, which misses swaps of round values)."""
    coefs, knots = snapshot
    cb = _bits(self.coefs)
    if cb.shape != coefs.shape or cb.dtype != coefs.dtype or not np.array_equal(cb, coefs):
        return False
    for k, snap in zip(self.knots, knots):
        kb = _bits(k)
        if kb.shape != snap.shape or kb.dtype != snap.dtype or not np.array_equal(kb, snap):
            return False
    return True


def _snapshot(self):
    return (_bits(self.coefs).copy(), tuple(_bits(k).copy() for k in self.knots))


def device_tables(self, device=None):
    """The DeviceSpline of ``self`` on ``device`` (created on first use, cached on the
    object).  The reference's attributes are mutable (bspy/_spline_fitting.py:529-540
    writes coefs in place), so the cache is validated on every call against private copies of
    knots and coefs (exact comparison) and the tables are re-uploaded when they changed."""
    device = get_device() if device is None else int(device)
    cache = self.__dict__.setdefault("_bsk_cache", {})
    dt = compute_dtype(self)
    layout = _layout(self)
    entry = cache.get(device)
    if entry is not None and entry[1] == layout and entry[0].dtype == dt:
        if _unchanged(entry[2], self):
            return entry[0]
        entry[0].update(self.knots, self.coefs)
        cache[device] = (entry[0], layout, _snapshot(self))
        return entry[0]
    if entry is not None:
        entry[0].close()
    tables = DeviceSpline(self.order, self.nCoef, self.knots, self.coefs, dt, device)
    cache[device] = (tables, layout, _snapshot(self))
    return tables


def bspline_values(knot, knots, splineOrder, u, derivativeOrder=0, taylorCoefs=False):
    """Reference bspy/_spline_evaluation.py:4-27.  Scalar ``u`` returns
    ``(knot, basis[splineOrder])``; an array ``u`` (extension) returns
    ``(knot[N], basis[N, splineOrder])``."""
    knots = np.asarray(knots)
    scalar = np.ndim(u) == 0
    ix, basis = bspline_values_batch(knots, splineOrder, np.atleast_1d(u), derivativeOrder, taylorCoefs, knot)
    if knots.dtype != basis.dtype and np.issubdtype(knots.dtype, np.floating):
        basis = basis.astype(knots.dtype)
    if scalar:
        return int(ix[0]), basis[0]
    return ix, basis


def domain(self):
    """Reference bspy/_spline_evaluation.py:135-138."""
    dom = [[self.knots[i][self.order[i] - 1], self.knots[i][self.nCoef[i]]] for i in range(self.nInd)]
    return np.array(dom)


def _point(self, with_respect_to, uvw):
    uvw = np.atleast_1d(uvw)
    if len(uvw) != self.nInd:
        raise ValueError(f"Incorrect number of parameter values: {len(uvw)}")
    if with_respect_to is not None and len(with_respect_to) != self.nInd:
        raise ValueError(f"Incorrect number of derivative orders: {len(with_respect_to)}")
    tables = device_tables(self)
    try:
        return tables.point(uvw, with_respect_to).astype(self.coefs.dtype, copy=False)   # reference: coefs' dtype
    except nv.DomainError:
        raise ValueError(f"Spline evaluation outside domain: {uvw}") from None


def evaluate(self, uvw):
    """Reference bspy/_spline_evaluation.py:140-164 (one point)."""
    return _point(self, None, uvw)


def derivative(self, with_respect_to, uvw):
    """Reference bspy/_spline_evaluation.py:109-133 (one point)."""
    return _point(self, [int(w) for w in with_respect_to], uvw)


def jacobian(self, uvw):
    """Reference bspy/_spline_evaluation.py:205-213: (nDep, nInd) at one point, in the
    coefficients' dtype.  One fused kernel instead of nInd derivative calls."""
    uvw = np.atleast_1d(uvw)
    if len(uvw) != self.nInd:
        raise ValueError(f"Incorrect number of parameter values: {len(uvw)}")
    tables = device_tables(self)
    try:
        out = tables.point(uvw, jacobian=True)
    except nv.DomainError:
        raise ValueError(f"Spline evaluation outside domain: {uvw}") from None
    return out.astype(self.coefs.dtype, copy=False)


def _select_normal(full, indices, normalize, xp):
    """Reference bspy/_spline_evaluation.py:234-244: only the selected cofactors are built and the result
    is divided by the norm of THAT vector.  ``full`` = the un-normalised cofactor vector, (rows, ...)."""
    sel = full[list(indices)]
    if normalize:
        sel = sel / xp.sqrt((sel * sel).sum(0))
    return sel


def normal(self, uvw, normalize=True, indices=None):
    """Reference bspy/_spline_evaluation.py:215-246: the normal at one point, or (extension)
    at arrays of points -> shape (len(indices or all), *broadcast_shape)."""
    if abs(self.nInd - self.nDep) != 1:
        raise ValueError("The number of independent variables must be one different than the number of dependent variables.")
    negate = bool(getattr(self, "metadata", {}).get("negateNormal", False))
    # with an index subset the GPU returns the area-scaled cofactors; selection and normalisation follow here
    gpu_normalize = normalize and indices is None
    batched = len(uvw) == self.nInd and self.nInd > 0 and not np.isscalar(uvw[0]) and \
        (uvw[0].ndim if hasattr(uvw[0], "ndim") else np.ndim(uvw[0])) > 0
    if batched:
        if any(_is_torch(a) for a in uvw):
            import torch
            dev = next(a for a in uvw if _is_torch(a) and a.is_cuda).device
            ts = torch.broadcast_tensors(*[(a if _is_torch(a) else torch.as_tensor(np.asarray(a))).to(dev) for a in uvw])
            shape = tuple(ts[0].shape)
            try:
                out = device_tables(self, dev.index).normal_device(ts, gpu_normalize, negate)
            except nv.DomainError as e:
                cpu = [t.reshape(-1)[e.index].item() for t in ts]
                raise ValueError(f"Spline evaluation outside domain: {np.atleast_1d(cpu)}") from None
            out = out.view((out.shape[0], *shape))
            return out if indices is None else _select_normal(out, indices, normalize, torch)
        arrays = [np.asarray(a) for a in uvw]
        shape = np.broadcast_shapes(*[a.shape for a in arrays])
        flat = [np.ascontiguousarray(np.broadcast_to(a, shape), compute_dtype(self)).reshape(-1) for a in arrays]
        try:
            out = device_tables(self).normal(flat, gpu_normalize, negate)
        except nv.DomainError as e:
            raise ValueError(_domain_message(arrays, shape, e.index)) from None
        out = out.reshape((out.shape[0], *shape))
        if indices is not None:
            out = _select_normal(out, indices, normalize, np)
        return out.astype(self.coefs.dtype, copy=False)
    uvw = np.atleast_1d(uvw)
    if len(uvw) != self.nInd:
        raise ValueError(f"Incorrect number of parameter values: {len(uvw)}")
    try:
        out = device_tables(self).normal([uvw[i:i + 1] for i in range(self.nInd)], gpu_normalize, negate)
    except nv.DomainError:
        raise ValueError(f"Spline evaluation outside domain: {uvw}") from None
    out = out[:, 0]
    if indices is not None:
        out = _select_normal(out, indices, normalize, np)
    return out.astype(self.coefs.dtype, copy=False)


def _graph(self):
    """The graph (u, f(u)) of a scalar-valued spline as a spline with nDep = nInd + 1 (reference
    bspy/_spline_operations.py:281-288): the Greville abscissae of variable i (reference
    bspy/_spline_evaluation.py:166-178) are the coefficients of the new dependent variable i.  The
    reference clamps the spline first (a knot-insertion operation outside the evaluation path); here
    the knots must already be clamped at both domain ends, as every spline built by the reference's
    fitting and construction routines is."""
    from .spline import Spline
    coefs = np.asarray(self.coefs)
    new = np.zeros((self.nInd + self.nDep, *coefs.shape[1:]), coefs.dtype)
    new[self.nInd:] = coefs
    dom = domain(self)
    for i in range(self.nInd):
        k = np.asarray(self.knots[i])
        o, nc = self.order[i], self.nCoef[i]
        if not (np.all(k[:o] == k[o - 1]) and np.all(k[nc:] == k[nc])):
            raise NotImplementedError("curvature of a scalar-valued spline needs knots clamped at the domain ends "
                                      "(the reference's clamp() is outside this package's evaluation path)")
        if o == 1:
            g = 0.5 * (k[1:] + k[:-1])
        else:
            g = sum(k[ix:ix + nc] for ix in range(1, o)) / (o - 1)
        g = np.minimum(dom[i][1], np.maximum(dom[i][0], g))
        shape = [1] * self.nInd
        shape[i] = nc
        new[i] = g.reshape(shape)
    return Spline(self.nInd, self.nInd + self.nDep, self.order, self.nCoef, self.knots, new, getattr(self, "metadata", {}))


def curvature(self, uv):
    """Reference bspy/_spline_evaluation.py:80-107: curvature of a curve (nDep >= 2; signed in
    2-D) or Gaussian curvature of a surface in 3-D, at one point or (extension) at arrays of
    points; a scalar-valued spline (nDep == 1) stands for the graph of the function, as in the reference."""
    if self.nDep == 1 and self.nInd in (1, 2):
        self = _graph(self)                     # reference :81-82: curvature of the graph of the function
    if not ((self.nInd == 1 and self.nDep >= 2) or (self.nInd == 2 and self.nDep == 3)):
        raise NotImplementedError("curvature needs a curve with nDep >= 2 or a surface with nDep == 3")
    if self.nInd == 1 and (np.isscalar(uv) or np.ndim(uv) == 0):
        uv = [uv]
    batched = len(uv) == self.nInd and not np.isscalar(uv[0]) and \
        (uv[0].ndim if hasattr(uv[0], "ndim") else np.ndim(uv[0])) > 0
    if batched:
        if any(_is_torch(a) for a in uv):
            import torch
            dev = next(a for a in uv if _is_torch(a) and a.is_cuda).device
            ts = torch.broadcast_tensors(*[(a if _is_torch(a) else torch.as_tensor(np.asarray(a))).to(dev) for a in uv])
            return device_tables(self, dev.index).curvature_device(ts).view(tuple(ts[0].shape))
        arrays = [np.asarray(a) for a in uv]
        shape = np.broadcast_shapes(*[a.shape for a in arrays])
        flat = [np.ascontiguousarray(np.broadcast_to(a, shape), compute_dtype(self)).reshape(-1) for a in arrays]
        try:
            out = device_tables(self).curvature(flat)
        except nv.DomainError as e:
            raise ValueError(_domain_message(arrays, shape, e.index)) from None
        return out.reshape(shape)
    uv = np.atleast_1d(uv)
    if len(uv) != self.nInd:
        raise ValueError(f"Incorrect number of parameter values: {len(uv)}")
    try:
        out = device_tables(self).curvature([uv[i:i + 1] for i in range(self.nInd)])
    except nv.DomainError:
        raise ValueError(f"Spline evaluation outside domain: {uv}") from None
    return out[0]


# --------------------------------------------------------------------------------------
# batched entry points
# --------------------------------------------------------------------------------------

def _grid_axes(arrays, shape):
    """If the broadcast of ``arrays`` is a tensor-product grid (every array varies along
    exactly one axis of ``shape``, all of them different), return for each array the axis
    it varies along (None for a single value); else None."""
    axes = []
    for a in arrays:
        nd = a.ndim
        varying = [len(shape) - nd + k for k in range(nd) if a.shape[k] != 1]
        if len(varying) > 1:
            return None
        axes.append(varying[0] if varying else None)
    used = [ax for ax in axes if ax is not None]
    if len(set(used)) != len(used):
        return None
    # every non-trivial axis of the result must be driven by exactly one variable
    for k, n in enumerate(shape):
        if n != 1 and k not in used:
            return None
    return axes


_GRID_MIN_POINTS = 4096


def _domain_message(arrays, shape, index):
    # the reference reports the first offending point in broadcast (C) order, as the
    # tuple of Python floats np.frompyfunc handed to the scalar routine
    pt = [float(np.broadcast_to(a, shape).reshape(-1)[index]) for a in arrays]
    return f"Spline evaluation outside domain: {np.atleast_1d(pt)}"


def evaluate_batch(self, with_respect_to, uvw, device=None, check=True):
    """Batched evaluate/derivative: ``uvw`` = nInd broadcastable arrays (NumPy path,
    results returned as NumPy) or CUDA torch tensors (device path, results stay on the
    GPU).  Returns an array of shape ``(nDep, *broadcast_shape)``."""
    if len(uvw) != self.nInd:
        raise ValueError(f"Incorrect number of parameter values: {len(uvw)}")
    wrt = None if with_respect_to is None else [int(w) for w in with_respect_to]
    if wrt is not None and len(wrt) != self.nInd:
        raise ValueError(f"Incorrect number of derivative orders: {len(wrt)}")
    if any(_is_torch(a) for a in uvw):
        import torch
        dev = next(a for a in uvw if _is_torch(a) and a.is_cuda).device
        ts = [a if _is_torch(a) else torch.as_tensor(np.asarray(a), device=dev) for a in uvw]
        ts = torch.broadcast_tensors(*[t.to(dev) for t in ts])
        shape = tuple(ts[0].shape)
        tables = device_tables(self, dev.index)
        try:
            out = tables.evaluate_device(ts, wrt, check=check)
        except nv.DomainError as e:
            cpu = [t.reshape(-1)[e.index].item() for t in ts]
            raise ValueError(f"Spline evaluation outside domain: {np.atleast_1d(cpu)}") from None
        return out.view((self.nDep, *shape))

    dt = compute_dtype(self)
    arrays = [np.asarray(a) for a in uvw]
    shape = np.broadcast_shapes(*[a.shape for a in arrays])
    total = int(np.prod(shape, dtype=np.int64))
    tables = device_tables(self, device)
    axes = _grid_axes(arrays, shape) if (self.nInd >= 2 and total >= _GRID_MIN_POINTS) else None
    try:
        if axes is not None:
            # tensor-product broadcast (e.g. s(u[:, None], v[None, :])): per-variable basis
            # rows once, then one contraction per grid point
            grid = tables.evaluate_grid([a.reshape(-1) for a in arrays], wrt)      # (nDep, n_0, ..., n_{nInd-1})
            # result axis k is driven by variable var_of[k]; variables with one value drop out
            order_axes = [iv for iv in range(self.nInd) if axes[iv] is not None]
            grid = grid.reshape((self.nDep, *[arrays[iv].size for iv in order_axes]))
            perm = sorted(range(len(order_axes)), key=lambda j: axes[order_axes[j]])
            grid = np.transpose(grid, (0, *[p + 1 for p in perm]))
            return np.ascontiguousarray(grid).reshape((self.nDep, *shape))
        flat = [np.ascontiguousarray(np.broadcast_to(a, shape), dt).reshape(-1) for a in arrays]
        out = tables.evaluate(flat, wrt)
    except nv.DomainError as e:
        index = e.index
        if axes is not None:
            # the grid kernel numbers points in variable order; the reference reports the
            # first offender in broadcast (C) order of the result
            dom = domain(self)
            bad = np.zeros(shape, bool)
            for iv, a in enumerate(arrays):
                bad |= np.broadcast_to((a < dom[iv][0]) | (a > dom[iv][1]), shape)
            index = int(np.argmax(bad.reshape(-1)))
        raise ValueError(_domain_message(arrays, shape, index)) from None
    return out.reshape((self.nDep, *shape))


def jacobian_batch(self, uvw, device=None, check=True):
    """Batched jacobian (extension; the reference's is single-point): returns
    ``(nDep, nInd, *broadcast_shape)``."""
    if len(uvw) != self.nInd:
        raise ValueError(f"Incorrect number of parameter values: {len(uvw)}")
    if any(_is_torch(a) for a in uvw):
        import torch
        dev = next(a for a in uvw if _is_torch(a) and a.is_cuda).device
        ts = torch.broadcast_tensors(*[(a if _is_torch(a) else torch.as_tensor(np.asarray(a))).to(dev) for a in uvw])
        shape = tuple(ts[0].shape)
        tables = device_tables(self, dev.index)
        try:
            out = tables.jacobian_device(ts, check=check)
        except nv.DomainError as e:
            cpu = [t.reshape(-1)[e.index].item() for t in ts]
            raise ValueError(f"Spline evaluation outside domain: {np.atleast_1d(cpu)}") from None
        return out.view((self.nDep, self.nInd, *shape))
    dt = compute_dtype(self)
    arrays = [np.asarray(a) for a in uvw]
    shape = np.broadcast_shapes(*[a.shape for a in arrays])
    flat = [np.ascontiguousarray(np.broadcast_to(a, shape), dt).reshape(-1) for a in arrays]
    try:
        out = device_tables(self, device).jacobian(flat)
    except nv.DomainError as e:
        raise ValueError(_domain_message(arrays, shape, e.index)) from None
    return out.reshape((self.nDep, self.nInd, *shape))
