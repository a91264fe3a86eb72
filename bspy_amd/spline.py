"""
``bspy_amd.Spline``: drop-in for the evaluation path of the reference's ``bspy.Spline``
(bspy/spline.py) - same constructor, attributes, call conventions, return types and
error messages for

    Spline(nInd, nDep, order, nCoef, knots, coefs, metadata={})   spline.py:46-76
    s(*uvw) / s.evaluate(*uvw)                                    spline.py:78-79, :904-949
    s.derivative(with_respect_to, *uvw)                           spline.py:720-770
    s.jacobian(uvw) / s.tangent_space(uvw)                        spline.py:1354-1377, :2238-2252
    s.domain()                                                    spline.py:794-808
    s.normal(uvw, normalize=True, indices=None)                   spline.py:1648-1682
    Spline.bspline_values(knot, knots, splineOrder, u, ...)       spline.py:207-252
    to_dict / from_dict / load / save (JSON, as an input format)   spline.py:1099-1125, :1542-1583, :1998-2026, :2254-2267

The arithmetic runs on the GPU (bspy_amd/_spline_evaluation.py -> libbspy_amd.so); the
rest of the reference's Spline API (fitting, intersection, CSG, viewer) is out of scope.

Documented deviations from the reference:
  * all-integer knots / coefs are promoted to float64 (the reference keeps int64 and
    silently truncates, e.g. Spline(1,1,[4],[4],[[0,0,0,0,1,1,1,1]],[[0.,1,2,3]])(0.5) -> 0);
  * batched calls with nDep == 1 and N-D inputs return the full broadcast shape (the
    reference returns only column 0 of each row);
  * of the ufunc keyword arguments of the reference's np.frompyfunc wrapper, where= and out= are
    honoured (out= also takes float arrays); the others (casting=, order=, ...) raise TypeError;
  * mixed float32/float64 inputs are computed in float64;
  * CUDA/HIP torch tensors are accepted as parameters (results stay on the GPU) and
    jacobian() accepts arrays of points (the reference's is single-point).
"""
import numpy as np

from . import _spline_evaluation as _ev


def _as_float_array(a):
    a = np.array(a)
    if not np.issubdtype(a.dtype, np.floating):
        a = a.astype(np.float64)
    elif a.dtype not in (np.float32, np.float64):
        a = a.astype(np.float64)
    return a


class Spline:
    """Tensor-product B-spline with nInd independent and nDep dependent variables
    (constructor semantics of the reference, bspy/spline.py:46-76)."""

    def __init__(self, nInd, nDep, order, nCoef, knots, coefs, metadata={}):
        # argument checks in the reference's order, with its messages (spline.py:47-69)
        def need(ok, message):
            if not ok:
                raise ValueError(message)

        need(nInd >= 0, "nInd < 0")
        need(nDep >= 0, "nDep < 0")
        self.nInd, self.nDep = int(nInd), int(nDep)
        need(len(order) == self.nInd, "len(order) != nInd")
        need(len(nCoef) == self.nInd, "len(nCoef) != nInd")
        self.order = tuple(map(int, order))
        self.nCoef = tuple(map(int, nCoef))
        need(len(knots) == nInd, "len(knots) != nInd")
        for variable, (kv, o, n) in enumerate(zip(knots, self.order, self.nCoef)):
            need(len(kv) == o + n, f"Knots array for variable {variable} should have length {o + n}")
        self.knots = tuple(_as_float_array(kv) for kv in knots)
        for kv, o, n in zip(self.knots, self.order, self.nCoef):
            # non-decreasing over the first nCoef + 1 knots, and every basis function has support
            # (knots[i + order] > knots[i]), reference spline.py:61-64
            proper = n == 0 or (bool(np.all(np.diff(kv[:n + 1]) >= 0)) and bool(np.all(kv[o:o + n] > kv[:n])))
            need(proper, "Improper knot order or multiplicity")
        points = int(np.prod(self.nCoef, dtype=np.int64)) if self.nCoef else 1
        need(len(coefs) in (points, self.nDep), f"Length of coefs should be {points} or {self.nDep}")
        table = _as_float_array(coefs)
        canonical = (self.nDep, *self.nCoef)
        if table.shape != canonical:
            if len(table) == points:
                # flat "list of points" form: nDep fastest, then the FIRST variable (spline.py:72-73)
                table = table.reshape((*reversed(self.nCoef), self.nDep)).T
            else:
                # one array per dependent variable, each stored last-variable-first (spline.py:74-75)
                table = np.stack([np.transpose(component) for component in table]).reshape(canonical)
        self.coefs = table
        self.metadata = dict(metadata)

    def __call__(self, *uvw, **kwargs):
        return self.evaluate(*uvw, **kwargs)

    def __repr__(self):
        return f"Spline({self.nInd}, {self.nDep}, {self.order}, {self.nCoef}, ...)"

    # ------------------------------------------------------------------ evaluation
    @staticmethod
    def bspline_values(knot, knots, splineOrder, u, derivativeOrder=0, taylorCoefs=False):
        """B-spline (derivative) basis values of one segment; reference spline.py:207-252."""
        return _ev.bspline_values(knot, knots, splineOrder, u, derivativeOrder, taylorCoefs)

    def _batched_masked(self, with_respect_to, uvw, where, out, device):
        """The ufunc keyword arguments the reference's np.frompyfunc call honours (bspy/spline.py:943-947):
        ``where`` - only the selected points are evaluated (an out-of-domain value under a False entry does not
        raise); unselected results are NaN, or what ``out`` held; ``out`` - nDep arrays of the broadcast shape
        (one array when nDep == 1) that receive the results (object arrays as np.frompyfunc needs, or float
        arrays).  Returns new arrays in the coefficients' dtype, as the reference does."""
        arrays = [np.asarray(a) for a in uvw]
        mask = np.asarray(True if where is None else where, dtype=bool)
        shape = np.broadcast_shapes(mask.shape, *[a.shape for a in arrays])
        outs = None
        if out is not None:
            outs = (out,) if not isinstance(out, (tuple, list)) else tuple(out)
            if len(outs) != self.nDep:
                raise ValueError(f"out must hold {self.nDep} arrays")
            shape = np.broadcast_shapes(shape, *[o.shape for o in outs])
            for o in outs:
                if o.shape != shape:
                    raise ValueError("non-broadcastable output operand")
        sel = np.broadcast_to(mask, shape)
        res = np.full((self.nDep, *shape), np.nan, _ev.compute_dtype(self))
        if outs is not None:
            for d, o in enumerate(outs):
                res[d] = np.asarray(o, dtype=res.dtype)
        pts = [np.broadcast_to(a, shape)[sel] for a in arrays]
        if pts[0].size:
            try:
                res[:, sel] = _ev.evaluate_batch(self, with_respect_to, pts, device=device)
            except ValueError as e:
                raise ValueError(str(e)) from None
        if outs is not None:
            for d, o in enumerate(outs):
                o[...] = res[d]
        res = res.astype(self.coefs.dtype, copy=False)
        return tuple(res[d] for d in range(self.nDep)) if self.nDep > 1 else res[0]

    def _batched(self, with_respect_to, uvw, kwargs):
        device = kwargs.pop("device", None)
        check = kwargs.pop("check", True)
        where = kwargs.pop("where", None)
        out = kwargs.pop("out", None)
        if kwargs:
            raise TypeError("ufunc keyword arguments other than where= / out= are not supported by bspy_amd: " + ", ".join(sorted(kwargs)))
        if where is not None or out is not None:
            if any(_ev._is_torch(a) for a in uvw):
                raise TypeError("where= / out= take NumPy arrays")
            return self._batched_masked(with_respect_to, uvw, where, out, device)
        out = _ev.evaluate_batch(self, with_respect_to, uvw, device=device, check=check)
        if _ev._is_torch(out):
            import torch
            tdt = torch.float32 if self.coefs.dtype == np.float32 else torch.float64
            out = out.to(tdt)
            return tuple(out[d] for d in range(self.nDep)) if self.nDep > 1 else out[0]
        out = out.astype(self.coefs.dtype, copy=False)
        return tuple(out[d] for d in range(self.nDep)) if self.nDep > 1 else out[0]

    def evaluate(self, *uvw, **kwargs):
        """Value of the spline; dispatch of the reference (spline.py:935-949):
        s(u, v) with scalars or s([u, v]) -> ndarray (nDep,);
        s(uArray, vArray) (broadcast, ufunc style) -> tuple of nDep arrays (one array when nDep == 1)."""
        if len(uvw) == 0 and self.nInd == 0:
            return self.coefs
        elif _isscalar(uvw[0]):
            return _ev.evaluate(self, uvw)
        elif len(uvw) > 1 or len(uvw[0]) > self.nInd:
            return self._batched(None, uvw, kwargs)
        else:
            return _ev.evaluate(self, *uvw)

    def derivative(self, with_respect_to, *uvw, **kwargs):
        """Derivative of the spline, with_respect_to[i] = derivative order in variable i;
        same dispatch as evaluate (reference spline.py:757-770)."""
        if len(uvw) == 0 and self.nInd == 0:
            return np.zeros(self.nDep, self.coefs.dtype)
        elif _isscalar(uvw[0]):
            return _ev.derivative(self, with_respect_to, uvw)
        elif len(uvw) > 1 or len(uvw[0]) > self.nInd:
            return self._batched(with_respect_to, uvw, kwargs)
        else:
            return _ev.derivative(self, with_respect_to, *uvw)

    def jacobian(self, uvw, **kwargs):
        """(nDep, nInd) matrix of first partial derivatives at one point (reference
        spline.py:1354-1377).  Extension: nInd arrays of points -> (nDep, nInd, *shape)."""
        if len(uvw) == self.nInd and self.nInd > 0 and not _isscalar(uvw[0]) and _ndim(uvw[0]) > 0:
            out = _ev.jacobian_batch(self, uvw, device=kwargs.pop("device", None), check=kwargs.pop("check", True))
            if _ev._is_torch(out):
                return out
            return out.astype(self.coefs.dtype, copy=False)
        return _ev.jacobian(self, uvw)

    def normal(self, uvw, normalize=True, indices=None):
        """Normal of the spline (|nInd - nDep| must be 1; reference spline.py:1648-1682): unit
        length by default, else the area-scaled cofactor vector; `metadata["negateNormal"]`
        flips it.  Extension: nInd arrays of points -> (len(normal), *shape)."""
        return _ev.normal(self, uvw, normalize, indices)

    def curvature(self, uv):
        """Curvature of a curve (nDep >= 2) or Gaussian curvature of a surface in 3-D
        (reference spline.py `curvature` -> _spline_evaluation.py:80-107); arrays of points
        give an array of curvatures."""
        return _ev.curvature(self, uv)

    def tangent_space(self, uvw):
        """Same as jacobian (reference spline.py:2238-2252)."""
        return _ev.jacobian(self, uvw)

    def domain(self):
        """nInd x 2 array of parameter bounds (reference spline.py:794-808)."""
        return _ev.domain(self)

    # ------------------------------------------------------------------ persistence (input format)
    def to_dict(self):
        """`dict` with the spline's data (reference spline.py:2254-2267)."""
        return {"type": "Spline", "nInd": self.nInd, "nDep": self.nDep, "order": self.order, "nCoef": self.nCoef,
                "knots": self.knots, "coefs": self.coefs, "metadata": self.metadata}

    @staticmethod
    def from_dict(dictionary):
        """Spline from a `dict` as written by `to_dict` / the reference's files, including the
        legacy "flipNormal" metadata key (reference spline.py:1099-1125)."""
        d = dictionary
        meta = dict(d.get("metadata", {}))
        if meta.pop("flipNormal", False):              # legacy name of negateNormal
            meta["negateNormal"] = True
        return Spline(d["nInd"], d["nDep"], d["order"], d["nCoef"], [np.asarray(k) for k in d["knots"]],
                      np.asarray(d["coefs"]), meta)

    @staticmethod
    def load(fileName):
        """List of splines from a JSON file in the reference's format: one object or a list of
        objects, nested `coefs` of shape (nDep, *nCoef) (reference spline.py:1542-1583; the
        legacy .npz branch is not supported)."""
        import json
        with open(fileName, "r", encoding="utf-8") as stream:
            content = json.load(stream)
        records = [content] if isinstance(content, dict) else content
        return [Spline.from_dict(r) for r in records if r.get("type", "Spline") == "Spline"]

    def save(self, fileName, *additional_splines):
        """Write this spline (and more) as JSON in the reference's format (spline.py:1998-2026):
        one object, or a list when more splines are given; arrays as nested lists, indent 4."""
        import json

        def plain(value):
            if isinstance(value, Spline):
                return plain(value.to_dict())
            if isinstance(value, np.ndarray):
                return value.tolist()
            if isinstance(value, dict):
                return {k: plain(v) for k, v in value.items()}
            if isinstance(value, (list, tuple)):
                return [plain(v) for v in value]
            if isinstance(value, np.generic):
                return value.item()
            return value

        document = plain(self) if not additional_splines else [plain(s) for s in (self, *additional_splines)]
        with open(fileName, "w", encoding="utf-8") as stream:
            json.dump(document, stream, indent=4)

    # ------------------------------------------------------------------ device tables
    def device_tables(self, device=None):
        """The cached DeviceSpline of this spline (explicit pinning: hold on to it and call
        its *_device methods to skip the per-call checksum of knots/coefs)."""
        return _ev.device_tables(self, device)


def _isscalar(x):
    return np.isscalar(x)


def _ndim(x):
    return x.ndim if hasattr(x, "ndim") else np.ndim(x)
