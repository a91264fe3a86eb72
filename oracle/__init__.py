"""
TEST INFRASTRUCTURE - NOT PRODUCT CODE.

CPU oracle for the batched B-spline evaluation path.  Two restatements of the
reference's algorithm (citations relative to /root/reference):

* ``py_*``  - pure Python/NumPy, one point at a time, structured exactly like
  ``bspy/_spline_evaluation.py:4-27`` (``bspline_values``), ``:109-164``
  (``derivative`` / ``evaluate``) and ``:205-213`` (``jacobian``).  Slow; used on
  small cases and as the "reference-structured" CPU timing.
* ``c_*``   - the plain-C restatement in ``bspline_oracle.c`` (same operation
  order, no FMA contraction), loaded through ctypes.  Fast enough to check the
  HIP path on millions of points and to serve as bench.py's ``cpu_baseline``.

Parity is PINNED: ``tests/test_oracle.py`` checks both against the reference's
own golden tables (``tests/bspy_test.py:15-564`` in the reference, captured in
``tests/golden/reference_tables.npz``) and against outputs of the reference
itself (``tests/golden/parity.npz``, ``basis.npz``, made by
``tests/golden/make_golden.py``).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package.  ``bspy_amd`` never does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    """Compile liboracle.so with gcc (seconds)."""
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("bspline_oracle.c", "bspline_oracle_impl.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def _lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        for suf in ("f64", "f32"):
            getattr(_LIB, "orc_bspline_values_" + suf).restype = ctypes.c_int
            getattr(_LIB, "orc_evaluate_" + suf).restype = ctypes.c_long
            getattr(_LIB, "orc_jacobian_" + suf).restype = ctypes.c_long
            getattr(_LIB, "orc_normal_" + suf).restype = ctypes.c_long
            getattr(_LIB, "orc_curvature_" + suf).restype = ctypes.c_long
    return _LIB


# ----------------------------------------------------------------------------
# pure-Python restatement
# ----------------------------------------------------------------------------

def py_bspline_values(knot, knots, splineOrder, u, derivativeOrder=0, taylorCoefs=False):
    """_spline_evaluation.py:4-27."""
    basis = np.zeros(splineOrder, knots.dtype)
    if knot is None:
        knot = int(np.searchsorted(knots, u, side="right"))
        knot = min(max(knot, splineOrder), len(knots) - splineOrder)
    if derivativeOrder >= splineOrder:
        return knot, basis
    basis[-1] = 1.0
    for degree in range(1, splineOrder - derivativeOrder):
        b = splineOrder - degree
        for i in range(knot - degree, knot):
            alpha = (u - knots[i]) / (knots[i + degree] - knots[i])
            basis[b - 1] += (1.0 - alpha) * basis[b]
            basis[b] *= alpha
            b += 1
    for degree in range(splineOrder - derivativeOrder, splineOrder):
        b = splineOrder - degree
        adjustment = degree / (splineOrder - degree if taylorCoefs else 1.0)
        for i in range(knot - degree, knot):
            alpha = adjustment / (knots[i + degree] - knots[i])
            basis[b - 1] += -alpha * basis[b]
            basis[b] *= alpha
            b += 1
    return knot, basis


def py_domain(order, nCoef, knots):
    """_spline_evaluation.py:135-138."""
    return np.array([[knots[i][order[i] - 1], knots[i][nCoef[i]]] for i in range(len(order))])


def py_derivative(order, nCoef, knots, coefs, with_respect_to, uvw):
    """_spline_evaluation.py:109-133 (and :140-164 when with_respect_to is all zeros)."""
    nInd, nDep = len(order), coefs.shape[0]
    uvw = np.atleast_1d(uvw)
    if len(uvw) != nInd:
        raise ValueError(f"Incorrect number of parameter values: {len(uvw)}")
    dom = py_domain(order, nCoef, knots)
    for ix in range(nInd):
        if uvw[ix] < dom[ix][0] or uvw[ix] > dom[ix][1]:
            raise ValueError(f"Spline evaluation outside domain: {uvw}")
    section = [slice(0, nDep)]
    bvalues = []
    for iv in range(nInd):
        ix, b = py_bspline_values(None, knots[iv], order[iv], uvw[iv], with_respect_to[iv])
        bvalues.append(b)
        section.append(slice(ix - order[iv], ix))
    mine = coefs[tuple(section)]
    for iv in range(nInd - 1, -1, -1):
        mine = mine @ bvalues[iv]
    return mine


def py_evaluate(order, nCoef, knots, coefs, uvw):
    return py_derivative(order, nCoef, knots, coefs, [0] * len(order), uvw)


def py_jacobian(order, nCoef, knots, coefs, uvw):
    """_spline_evaluation.py:205-213."""
    nInd, nDep = len(order), coefs.shape[0]
    value = np.empty((nDep, nInd), coefs.dtype)
    wrt = [0] * nInd
    for i in range(nInd):
        wrt[i] = 1
        value[:, i] = py_derivative(order, nCoef, knots, coefs, wrt, uvw)
        wrt[i] = 0
    return value


def py_batch(order, nCoef, knots, coefs, with_respect_to, points):
    """The reference's batched wrapper (spline.py:757-770 / :936-949): one Python
    call per point, results gathered SoA as (nDep, N) in the coefficients' dtype."""
    n = len(points[0])
    out = np.empty((coefs.shape[0], n), coefs.dtype)
    for i in range(n):
        # frompyfunc hands each element over as a Python float
        out[:, i] = py_derivative(order, nCoef, knots, coefs, with_respect_to, [float(p[i]) for p in points])
    return out


# ----------------------------------------------------------------------------
# C restatement through ctypes
# ----------------------------------------------------------------------------

def _compute_dtype(knots, coefs):
    """fp32 only when knots AND coefficients are fp32 (the all-fp32 reference path);
    everything else is compared in fp64."""
    if all(k.dtype == np.float32 for k in knots) and coefs.dtype == np.float32:
        return np.float32
    return np.float64


def _prep(order, nCoef, knots, coefs, points):
    dt = _compute_dtype(knots, coefs)
    suf = "f32" if dt == np.float32 else "f64"
    ct = ctypes.c_float if dt == np.float32 else ctypes.c_double
    ks = [np.ascontiguousarray(k, dt) for k in knots]
    cf = np.ascontiguousarray(coefs, dt)
    ps = [np.ascontiguousarray(p, dt).ravel() for p in points]
    nInd = len(order)
    PtrArr = ctypes.POINTER(ct) * max(nInd, 1)
    kp = PtrArr(*[k.ctypes.data_as(ctypes.POINTER(ct)) for k in ks])
    pp = PtrArr(*[p.ctypes.data_as(ctypes.POINTER(ct)) for p in ps])
    IntArr = ctypes.c_int * max(nInd, 1)
    return dt, suf, ct, ks, cf, ps, kp, pp, IntArr(*order), IntArr(*nCoef), IntArr


def c_bspline_values(knot, knots, splineOrder, u, derivativeOrder=0, taylorCoefs=False):
    knots = np.ascontiguousarray(knots)
    dt = np.float32 if knots.dtype == np.float32 else np.float64
    knots = knots.astype(dt, copy=False)
    ct = ctypes.c_float if dt == np.float32 else ctypes.c_double
    basis = np.zeros(splineOrder, dt)
    fn = getattr(_lib(), "orc_bspline_values_" + ("f32" if dt == np.float32 else "f64"))
    ix = fn(ctypes.c_int(-1 if knot is None else int(knot)), knots.ctypes.data_as(ctypes.POINTER(ct)),
            ctypes.c_int(len(knots)), ctypes.c_int(splineOrder), ct(float(u)), ctypes.c_int(derivativeOrder),
            ctypes.c_int(1 if taylorCoefs else 0), basis.ctypes.data_as(ctypes.POINTER(ct)))
    return ix, basis


def c_evaluate(order, nCoef, knots, coefs, with_respect_to, points):
    """Batched evaluate/derivative.  Returns (out (nDep, N), first_bad_index or -1)."""
    dt, suf, ct, ks, cf, ps, kp, pp, o, c, IntArr = _prep(order, nCoef, knots, coefs, points)
    n = len(ps[0]) if ps else 1
    out = np.zeros((cf.shape[0], n), dt)
    wrt = IntArr(*with_respect_to) if with_respect_to is not None else None
    bad = getattr(_lib(), "orc_evaluate_" + suf)(
        ctypes.c_int(len(order)), ctypes.c_int(cf.shape[0]), o, c, kp, cf.ctypes.data_as(ctypes.POINTER(ct)),
        wrt, pp, ctypes.c_long(n), out.ctypes.data_as(ctypes.POINTER(ct)))
    return out, int(bad)


def c_jacobian(order, nCoef, knots, coefs, points):
    """Batched jacobian.  Returns (out (nDep, nInd, N), first_bad_index or -1)."""
    dt, suf, ct, ks, cf, ps, kp, pp, o, c, IntArr = _prep(order, nCoef, knots, coefs, points)
    n = len(ps[0]) if ps else 1
    out = np.zeros((cf.shape[0], len(order), n), dt)
    bad = getattr(_lib(), "orc_jacobian_" + suf)(
        ctypes.c_int(len(order)), ctypes.c_int(cf.shape[0]), o, c, kp, cf.ctypes.data_as(ctypes.POINTER(ct)),
        pp, ctypes.c_long(n), out.ctypes.data_as(ctypes.POINTER(ct)))
    return out, int(bad)


def c_normal(order, nCoef, knots, coefs, points, normalize=True, negate=False):
    """Batched normal (bspy/_spline_evaluation.py:215-246).  Returns (out (max(nInd, nDep), N),
    first_bad_index or -1)."""
    dt, suf, ct, ks, cf, ps, kp, pp, o, c, IntArr = _prep(order, nCoef, knots, coefs, points)
    n = len(ps[0]) if ps else 1
    big = max(len(order), cf.shape[0])
    out = np.zeros((big, n), dt)
    bad = getattr(_lib(), "orc_normal_" + suf)(
        ctypes.c_int(len(order)), ctypes.c_int(cf.shape[0]), o, c, kp, cf.ctypes.data_as(ctypes.POINTER(ct)),
        pp, ctypes.c_long(n), ctypes.c_int(1 if normalize else 0), ctypes.c_int(1 if negate else 0),
        out.ctypes.data_as(ctypes.POINTER(ct)))
    return out, int(bad)


def py_normal(order, nCoef, knots, coefs, uvw, normalize=True, negate=False, indices=None):
    """_spline_evaluation.py:215-246, one point."""
    nInd, nDep = len(order), coefs.shape[0]
    if abs(nInd - nDep) != 1:
        raise ValueError("The number of independent variables must be one different than the number of dependent variables.")
    tangentSpace = py_jacobian(order, nCoef, knots, coefs, uvw)
    if nInd > nDep:
        big = nInd
        tangentSpace = tangentSpace.T
    else:
        big = nDep
    sign = -1 if negate else 1
    idx = range(big) if indices is None else indices
    normal = np.empty(len(idx), coefs.dtype)
    for k, i in enumerate(idx):
        normal[k] = sign * ((-1) ** i) * np.linalg.det(tangentSpace[[j for j in range(big) if i != j]])
    if normalize:
        normal /= np.linalg.norm(normal)
    return normal


def c_curvature(order, nCoef, knots, coefs, points):
    """Batched curvature (bspy/_spline_evaluation.py:80-107): curves with nDep >= 2 and
    surfaces in 3-D.  Returns (out (N,), first_bad_index or -1)."""
    dt, suf, ct, ks, cf, ps, kp, pp, o, c, IntArr = _prep(order, nCoef, knots, coefs, points)
    n = len(ps[0]) if ps else 1
    out = np.zeros(n, dt)
    bad = getattr(_lib(), "orc_curvature_" + suf)(
        ctypes.c_int(len(order)), ctypes.c_int(cf.shape[0]), o, c, kp, cf.ctypes.data_as(ctypes.POINTER(ct)),
        pp, ctypes.c_long(n), out.ctypes.data_as(ctypes.POINTER(ct)))
    return out, int(bad)
