"""SplineBlock - the evaluation path of the reference's system-of-equations container
(reference bspy/spline_block.py:9-113 constructor, :179-247
derivative / domain / evaluate / jacobian).

A block is a list of rows of splines; the splines of a row are summed, rows are stacked:
``[[F, G], [h]]`` represents ``(F + G, h)``.  Every entry may carry a map from the spline's
independent variables to the block's (``[([3, 4, 5], F), ([4, 0], G)], ...``).

Only the evaluation path is built here (this repository's scope): the reference's contours /
zeros / normal_spline / algebra on blocks are CAD algorithms and stay out of scope.

Single-point calls keep the reference's signatures and return types.  Passing arrays (or CUDA
torch tensors) for the parameter values evaluates the whole batch on the GPU: the points are
uploaded once, every spline of the block runs its batched kernel on its mapped rows, rows with
one spline are written in place and rows with several are accumulated on the device, and the
result comes back once.
"""
import numpy as np

from . import _spline_evaluation as _se


def _is_torch(x):
    return type(x).__module__.split(".")[0] == "torch"


def _looks_like_spline(x):
    return all(hasattr(x, a) for a in ("nInd", "nDep", "order", "nCoef", "knots", "coefs"))


def _select(index_map, values):
    """The entries of ``values`` a spline sees through its variable map."""
    if isinstance(values, np.ndarray):
        return values[index_map]
    return [values[i] for i in index_map]


class SplineBlock:
    """System of equations made of splines: rows are stacked, the splines of a row are summed.
    Same constructor semantics, attributes (``block``, ``nInd``, ``nDep``, ``knotsDtype``,
    ``coefsDtype``, ``size``) and error messages as the reference (bspy/spline_block.py:54-113)."""

    def __init__(self, block):
        # a bare spline is a 1 x 1 block, a flat list of splines one row (reference :55-58)
        if _looks_like_spline(block):
            rows = [[block]]
        elif _looks_like_spline(block[0]) or (len(block) > 1 and _looks_like_spline(block[1])):
            rows = [block]
        else:
            rows = block

        self.block = []
        self.knotsDtype = self.coefsDtype = None
        self.nDep = self.size = 0
        bounds = {}                                   # block variable -> (lower, upper), first claim wins

        def claim(variable, lower, upper, taken):
            if variable in taken:
                raise ValueError(f"Multiple splines in the same row map to independent variable {variable}")
            taken.add(variable)
            known = bounds.setdefault(variable, (lower, upper))
            if known[0] != lower or known[1] != upper:
                raise ValueError("Domains of independent variables must match")

        for row in rows:
            entries, taken, row_dep, next_default = [], set(), None, 0
            for item in row:
                if _looks_like_spline(item):            # default map: consecutive variables along the row
                    spline, index_map = item, list(range(next_default, next_default + item.nInd))
                else:
                    index_map, spline = list(item[0]), item[1]
                next_default += spline.nInd
                if not row_dep:
                    row_dep = spline.nDep
                    if self.nDep == 0:                  # the first spline fixes the block's dtypes
                        self.knotsDtype, self.coefsDtype = spline.knots[0].dtype, spline.coefs.dtype
                elif spline.nDep != row_dep:
                    raise ValueError("All splines in the same row must have the same nDep")
                for (lower, upper), variable in zip(_se.domain(spline), index_map):
                    claim(variable, lower, upper, taken)
                entries.append((index_map, spline))
            if row_dep:
                self.block.append(entries)
                self.nDep += row_dep
                self.size += len(row)

        self.nInd = len(bounds)
        missing = [v for v in range(self.nInd) if v not in bounds]
        if missing:
            raise ValueError(f"Block is missing independent variable {missing[0]}")
        self._domain = np.array([bounds[v] for v in range(self.nInd)], self.knotsDtype)

    def __repr__(self):
        return f"SplineBlock({self.block})"

    def __call__(self, uvw):
        return self.evaluate(uvw)

    def domain(self):
        """nInd x 2 array of lower / upper bounds (reference spline_block.py:199-208)."""
        return self._domain

    # ------------------------------------------------------------------ single point (reference API)
    def _sum_rows(self, point_function, *arguments):
        """Every row's splines evaluated on their mapped arguments and summed (reference :37-44)."""
        total = np.zeros(self.nDep, self.coefsDtype)
        first = 0
        for entries in self.block:
            width = entries[0][1].nDep
            for index_map, spline in entries:
                total[first:first + width] += point_function(spline, *[_select(index_map, a) for a in arguments])
            first += width
        return total

    @staticmethod
    def _is_batch(uvw):
        return any(_is_torch(a) or np.ndim(a) > 0 for a in uvw)

    def evaluate(self, uvw):
        """Value of the block at ``uvw`` (length nInd): array of size nDep (reference
        spline_block.py:210-224).  Arrays / CUDA tensors per variable -> ``(nDep, *shape)``."""
        if self._is_batch(uvw):
            return self._batch(None, uvw)
        return self._sum_rows(_se.evaluate, uvw)

    def derivative(self, with_respect_to, uvw):
        """Derivative of the block (reference spline_block.py:179-197)."""
        if self._is_batch(uvw):
            return self._batch([int(w) for w in with_respect_to], uvw)
        return self._sum_rows(_se.derivative, with_respect_to, uvw)

    def jacobian(self, uvw):
        """(nDep, nInd) jacobian at one point (reference spline_block.py:226-247); arrays /
        CUDA tensors per variable -> ``(nDep, nInd, *shape)``."""
        if self._is_batch(uvw):
            return self._batch_jacobian(uvw)
        point = np.atleast_1d(uvw)
        result = np.zeros((self.nDep, self.nInd), self.coefsDtype)
        first = 0
        for entries in self.block:
            width = entries[0][1].nDep
            for index_map, spline in entries:
                result[first:first + width, index_map] += _se.jacobian(spline, point[index_map])
            first += width
        return result

    # ------------------------------------------------------------------ batched (GPU resident)
    def _device_points(self, uvw):
        """-> (torch, list of nInd flat CUDA tensors, broadcast shape, inputs were torch?)"""
        import torch
        from .device_spline import get_device
        if len(uvw) != self.nInd:
            raise ValueError(f"Incorrect number of parameter values: {len(uvw)}")
        was_torch = any(_is_torch(a) for a in uvw)
        if was_torch:
            dev = next(a for a in uvw if _is_torch(a) and a.is_cuda).device
        else:
            dev = torch.device("cuda", get_device())
        ts = [a if _is_torch(a) else torch.as_tensor(np.asarray(a)) for a in uvw]
        ts = torch.broadcast_tensors(*[t.to(dev) for t in ts])
        shape = tuple(ts[0].shape)
        return torch, [t.reshape(-1) for t in ts], shape, was_torch, dev

    def _torch_dtype(self, torch):
        return torch.float32 if np.dtype(self.coefsDtype) == np.float32 else torch.float64

    def _raise_domain(self, flat, imap, index):
        pt = np.array([float(flat[i][index].item()) for i in imap])
        raise ValueError(f"Spline evaluation outside domain: {pt}") from None

    def _batch(self, wrt, uvw):
        from . import _native as nv
        torch, flat, shape, was_torch, dev = self._device_points(uvw)
        if wrt is not None and len(wrt) != self.nInd:
            raise ValueError(f"Incorrect number of derivative orders: {len(wrt)}")
        n = flat[0].numel()
        odt = self._torch_dtype(torch)
        out = torch.empty((self.nDep, n), dtype=odt, device=dev)
        n_dep = 0
        for row in self.block:
            k = row[0][1].nDep
            target = out[n_dep:n_dep + k]
            for j, (imap, spline) in enumerate(row):
                tables = _se.device_tables(spline, dev.index)
                pts = [flat[i] for i in imap]
                w = None if wrt is None else [wrt[i] for i in imap]
                same = (tables.dtype == np.dtype(self.coefsDtype))
                try:
                    if j == 0 and same:
                        tables.evaluate_device(pts, w, out=target)          # written in place
                    else:
                        part = tables.evaluate_device(pts, w).to(odt)
                        if j == 0:
                            target.copy_(part)
                        else:
                            target.add_(part)
                except nv.DomainError as e:
                    self._raise_domain(flat, imap, e.index)
            n_dep += k
        out = out.view((self.nDep, *shape))
        return out if was_torch else out.cpu().numpy()

    def _batch_jacobian(self, uvw):
        from . import _native as nv
        torch, flat, shape, was_torch, dev = self._device_points(uvw)
        n = flat[0].numel()
        odt = self._torch_dtype(torch)
        out = torch.zeros((self.nDep, self.nInd, n), dtype=odt, device=dev)
        n_dep = 0
        for row in self.block:
            k = row[0][1].nDep
            for imap, spline in row:
                tables = _se.device_tables(spline, dev.index)
                try:
                    part = tables.jacobian_device([flat[i] for i in imap]).to(odt)     # (k, nInd_s, n)
                except nv.DomainError as e:
                    self._raise_domain(flat, imap, e.index)
                out[n_dep:n_dep + k, imap] += part
            n_dep += k
        out = out.view((self.nDep, self.nInd, *shape))
        return out if was_torch else out.cpu().numpy()
