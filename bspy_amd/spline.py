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
  * ufunc keyword arguments (where=, out=) of the reference's np.frompyfunc wrapper are
    not supported;
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
        if not (nInd >= 0): raise ValueError("nInd < 0")
        self.nInd = int(nInd)
        if not (nDep >= 0): raise ValueError("nDep < 0")
        self.nDep = int(nDep)
        if not (len(order) == self.nInd): raise ValueError("len(order) != nInd")
        self.order = tuple(int(x) for x in order)
        if not (len(nCoef) == self.nInd): raise ValueError("len(nCoef) != nInd")
        self.nCoef = tuple(int(x) for x in nCoef)
        if not (len(knots) == nInd): raise ValueError("len(knots) != nInd")
        for i in range(len(knots)):
            nKnots = self.order[i] + self.nCoef[i]
            if not (len(knots[i]) == nKnots):
                raise ValueError(f"Knots array for variable {i} should have length {nKnots}")
        self.knots = tuple(_as_float_array(kk) for kk in knots)
        for kk, o, n in zip(self.knots, self.order, self.nCoef):
            # knots[i] <= knots[i+1] and knots[i+order] - knots[i] > 0 for i < nCoef (spline.py:61-64)
            if n > 0 and not (np.all(kk[:n] <= kk[1:n + 1]) and np.all(kk[o:o + n] - kk[:n] > 0)):
                raise ValueError("Improper knot order or multiplicity")
        totalCoefs = 1
        for n in self.nCoef:
            totalCoefs *= n
        if not (len(coefs) == totalCoefs or len(coefs) == self.nDep):
            raise ValueError(f"Length of coefs should be {totalCoefs} or {self.nDep}")
        self.coefs = _as_float_array(coefs)
        if self.coefs.shape != (self.nDep, *self.nCoef):
            if len(self.coefs) == totalCoefs:
                # flat "list of points" form: nDep fastest, then the FIRST variable (spline.py:72-73)
                self.coefs = self.coefs.reshape((*self.nCoef[::-1], self.nDep)).T
            else:
                self.coefs = np.array([c.T for c in self.coefs]).reshape((self.nDep, *self.nCoef))
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

    def _batched(self, with_respect_to, uvw, kwargs):
        device = kwargs.pop("device", None)
        check = kwargs.pop("check", True)
        if kwargs:
            raise TypeError("ufunc keyword arguments are not supported by bspy_amd: " + ", ".join(sorted(kwargs)))
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
        spline = Spline(dictionary["nInd"], dictionary["nDep"], dictionary["order"], dictionary["nCoef"],
                        [np.array(knots) for knots in dictionary["knots"]], np.array(dictionary["coefs"]),
                        dictionary.get("metadata", {}))
        if spline.metadata.get("flipNormal", False):
            spline.metadata["negateNormal"] = True
            del spline.metadata["flipNormal"]
        return spline

    @staticmethod
    def load(fileName):
        """List of splines from a JSON file in the reference's format: one object or a list of
        objects, nested `coefs` of shape (nDep, *nCoef) (reference spline.py:1542-1583; the
        legacy .npz branch is not supported)."""
        import json
        with open(fileName, "r", encoding="utf-8") as file:
            data = json.load(file)
        if isinstance(data, dict):
            data = [data]
        return [Spline.from_dict(d) for d in data if d.get("type", "Spline") == "Spline"]

    def save(self, fileName, *additional_splines):
        """Write this spline (and more) as JSON in the reference's format (spline.py:1998-2026)."""
        import json

        class SplineEncoder(json.JSONEncoder):
            def default(self, obj):
                if isinstance(obj, np.ndarray):
                    return obj.tolist()
                if isinstance(obj, Spline):
                    return obj.to_dict()
                return super().default(obj)

        with open(fileName, "w", encoding="utf-8") as file:
            if additional_splines:
                json.dump((self, *additional_splines), file, indent=4, cls=SplineEncoder)
            else:
                json.dump(self, file, indent=4, cls=SplineEncoder)

    # ------------------------------------------------------------------ device tables
    def device_tables(self, device=None):
        """The cached DeviceSpline of this spline (explicit pinning: hold on to it and call
        its *_device methods to skip the per-call checksum of knots/coefs)."""
        return _ev.device_tables(self, device)


def _isscalar(x):
    return np.isscalar(x)


def _ndim(x):
    return x.ndim if hasattr(x, "ndim") else np.ndim(x)
