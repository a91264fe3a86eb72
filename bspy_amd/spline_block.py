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


class SplineBlock:
    """Same constructor semantics as the reference (bspy/spline_block.py:54-113)."""

    @staticmethod
    def _map_args(map, args):
        # reference spline_block.py:33-35
        return [arg[map] if isinstance(arg, np.ndarray) else [arg[index] for index in map] for arg in args]

    def __init__(self, block):
        from .spline import Spline
        is_spline = lambda s: isinstance(s, Spline) or all(hasattr(s, a) for a in ("nInd", "nDep", "order", "nCoef", "knots", "coefs"))
        if is_spline(block):
            block = [[block]]
        elif is_spline(block[0]) or (len(block) > 1 and is_spline(block[1])):
            block = [block]

        self.block = []
        self.nInd = 0
        self.nDep = 0
        self.knotsDtype = None
        self.coefsDtype = None
        self.size = 0
        domain = {}
        for row in block:
            row_dep = 0
            row_ind = 0
            used = set()
            new_row = []
            for entry in row:
                if is_spline(entry):
                    spline = entry
                    imap = list(range(row_ind, row_ind + spline.nInd))
                else:
                    imap, spline = entry
                    imap = list(imap)
                row_ind += spline.nInd
                if row_dep == 0:
                    row_dep = spline.nDep
                    if self.nDep == 0:
                        self.knotsDtype = spline.knots[0].dtype
                        self.coefsDtype = spline.coefs.dtype
                elif row_dep != spline.nDep:
                    raise ValueError("All splines in the same row must have the same nDep")
                d = _se.domain(spline)
                for ind, i in enumerate(imap):
                    if i in used:
                        raise ValueError(f"Multiple splines in the same row map to independent variable {i}")
                    used.add(i)
                    if i in domain:
                        if domain[i][0] != d[ind, 0] or domain[i][1] != d[ind, 1]:
                            raise ValueError("Domains of independent variables must match")
                    else:
                        domain[i] = d[ind]
                new_row.append((imap, spline))
            if row_dep > 0:
                self.nDep += row_dep
                self.size += len(row)
                self.block.append(new_row)

        self.nInd = len(domain)
        dom = []
        for i in range(self.nInd):
            if i not in domain:
                raise ValueError(f"Block is missing independent variable {i}")
            dom.append(domain[i])
        self._domain = np.array(dom, self.knotsDtype)

    def __call__(self, uvw):
        return self.evaluate(uvw)

    def __repr__(self):
        return f"SplineBlock({self.block})"

    def domain(self):
        """nInd x 2 array of lower / upper bounds (reference spline_block.py:199-208)."""
        return self._domain

    # ------------------------------------------------------------------ single point (reference API)
    def _block_evaluation(self, function, args):
        # reference spline_block.py:37-44
        value = np.zeros(self.nDep, self.coefsDtype)
        n_dep = 0
        for row in self.block:
            for imap, spline in row:
                value[n_dep:n_dep + spline.nDep] += function(spline, *SplineBlock._map_args(imap, args))
            n_dep += spline.nDep
        return value

    @staticmethod
    def _is_batch(uvw):
        return any(_is_torch(a) or np.ndim(a) > 0 for a in uvw)

    def evaluate(self, uvw):
        """Value of the block at ``uvw`` (length nInd): array of size nDep (reference
        spline_block.py:210-224).  Arrays / CUDA tensors per variable -> ``(nDep, *shape)``."""
        if self._is_batch(uvw):
            return self._batch(None, uvw)
        return self._block_evaluation(_se.evaluate, (uvw,))

    def derivative(self, with_respect_to, uvw):
        """Derivative of the block (reference spline_block.py:179-197)."""
        if self._is_batch(uvw):
            return self._batch([int(w) for w in with_respect_to], uvw)
        return self._block_evaluation(_se.derivative, (with_respect_to, uvw))

    def jacobian(self, uvw):
        """(nDep, nInd) jacobian at one point (reference spline_block.py:226-247); arrays /
        CUDA tensors per variable -> ``(nDep, nInd, *shape)``."""
        if self._is_batch(uvw):
            return self._batch_jacobian(uvw)
        jac = np.zeros((self.nDep, self.nInd), self.coefsDtype)
        uvw = np.atleast_1d(uvw)
        n_dep = 0
        for row in self.block:
            for imap, spline in row:
                jac[n_dep:n_dep + spline.nDep, imap] += _se.jacobian(spline, uvw[imap])
            n_dep += spline.nDep
        return jac

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
