"""
Multi-GPU evaluation: the point batch is sharded across ranks (one process per GPU,
``torch.distributed``; backend "nccl" is RCCL over xGMI on ROCm), every rank evaluates its
contiguous shard with its own replica of the spline tables (<= 1 MB), and - only when the
caller asks for the full result everywhere - the shards are all-gathered.

The reference has no distributed code at all; points are independent, so the
path shards with NO data-path collective.  ``gather=True`` adds the one exchange step
BASELINE.json names (all-gather of results): ONE ``all_gather_into_tensor`` of every rank's
``(rows, ceil(N / world))`` block followed by one device copy into the ``(rows, N)`` SoA layout.  Over 7 xGMI links (~153 GB/s each) that gather costs several times the
kernel, so leave results sharded (``gather=False``) when the consumer is
sharded too.
"""
import numpy as np


def shard_bounds(n, world, rank):
    """Contiguous shard [start, stop) of rank ``rank``: ceil(n / world) points per rank,
    the tail ranks may be short or empty."""
    chunk = -(-int(n) // int(world)) if n > 0 else 0
    start = min(rank * chunk, n)
    return start, min(start + chunk, n)


def shard_chunk(n, world):
    return -(-int(n) // int(world)) if n > 0 else 0


class ShardedEvaluator:
    """Sharded ``evaluate`` / ``derivative`` / ``jacobian`` of one spline.

    ``spline``      a ``bspy_amd.Spline`` (or anything with its attributes)
    ``group``       torch.distributed process group (default: WORLD)
    ``collectives_at_world1``  run the collectives in a one-rank group too (tests)
    ``local_eval``  TEST SEAM, not a product feature: optional ``f(op, points, wrt) -> array (rows, m)``
                    replacing the local GPU evaluation, so that the world-size-2 ``gloo`` tests can run
                    the sharding and collective logic on a machine without a GPU (they inject the CPU
                    checker there).  ``op`` is "evaluate" or "jacobian".  Nothing in the package passes
                    it; the default - and the only product path - is this rank's DeviceSpline.
    """

    def __init__(self, spline, group=None, local_eval=None, device=None, collectives_at_world1=False):
        import torch.distributed as dist
        self.dist = dist
        self.spline = spline
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.local_eval = local_eval
        self.device = device
        # run the all-reduce / all-gather even in a one-rank group (exercises the RCCL path on one GPU)
        self.coll = dist.is_initialized() and (self.world > 1 or bool(collectives_at_world1))

    # -- local compute -------------------------------------------------------------------
    def _local(self, op, pts, wrt):
        """Returns (result or None, local index of the first out-of-domain point or -1)."""
        if self.local_eval is not None:
            return self.local_eval(op, pts, wrt)
        from . import _native as nv
        from . import _spline_evaluation as ev
        tables = ev.device_tables(self.spline, self.device)
        try:
            if ev._is_torch(pts[0]):
                if op == "jacobian":
                    res = tables.jacobian_device(pts, check=False).view(-1, pts[0].numel())
                else:
                    res = tables.evaluate_device(pts, wrt, check=False)
                tables.domain_status()          # the kernel's own inclusive-domain check
            elif op == "jacobian":
                res = tables.jacobian(pts).reshape(-1, len(pts[0]))
            else:
                res = tables.evaluate(pts, wrt)
        except nv.DomainError as e:
            return None, e.index
        return res, -1

    # -- public API ----------------------------------------------------------------------
    def evaluate(self, points, wrt=None, gather=True, sharded_input=False, total=None, check=True):
        return self._run("evaluate", points, wrt, gather, sharded_input, total, check)

    def derivative(self, with_respect_to, points, gather=True, sharded_input=False, total=None, check=True):
        return self._run("evaluate", points, [int(w) for w in with_respect_to], gather, sharded_input, total, check)

    def jacobian(self, points, gather=True, sharded_input=False, total=None, check=True):
        return self._run("jacobian", points, None, gather, sharded_input, total, check)

    def _run(self, op, points, wrt, gather, sharded_input, total, check=True):
        """points: nInd 1-D arrays/tensors.  With ``sharded_input`` each rank passes only its
        own shard (``total`` = global point count, shards as ``shard_bounds``); otherwise
        every rank passes the full batch and takes its slice.
        Returns (rows, N) if gather else this rank's (rows, n_local); rows = nDep
        (evaluate/derivative) or nDep * nInd (jacobian, row = d * nInd + j).
        ``check=False`` skips the blocking agreement on the first out-of-domain point (a MIN
        all-reduce + host read per call).  A rank whose own shard holds an offender still raises,
        but only AFTER it has taken part in the all-gather (with NaN rows for its shard), so the
        other ranks are never left waiting in the collective; they receive those NaN rows."""
        import torch
        nind = self.spline.nInd
        if len(points) != nind:
            raise ValueError(f"Incorrect number of parameter values: {len(points)}")
        if sharded_input:
            if total is None:
                raise ValueError("total (global point count) is required with sharded_input")
            n = int(total)
            start, stop = shard_bounds(n, self.world, self.rank)
            local = list(points)
            if len(local[0]) != stop - start:
                raise ValueError(f"rank {self.rank} must pass {stop - start} points, got {len(local[0])}")
        else:
            n = len(points[0])
            start, stop = shard_bounds(n, self.world, self.rank)
            local = [p[start:stop] for p in points]
        res, bad_local = self._local(op, local, wrt)

        # Collectives run where the backend needs them: RCCL ("nccl") only moves device memory, so NumPy
        # results / the offender index are staged on this rank's GPU; gloo takes them as they are.
        is_t = hasattr(res, "is_cuda") or (res is None and hasattr(local[0], "is_cuda"))
        cdev = None
        if self.coll and self.dist.get_backend(self.group) == "nccl":
            if is_t and hasattr(local[0], "is_cuda"):
                cdev = local[0].device
            else:
                from .device_spline import get_device
                cdev = torch.device("cuda", get_device() if self.device is None else int(self.device))

        # the reference raises for the first offending point; make every rank agree on it
        # (one MIN all-reduce; ``check=False`` skips it - the kernels' own record stays queryable)
        import_max = np.iinfo(np.int64).max
        bad_here = start + bad_local if bad_local >= 0 else import_max
        if self.coll and check:
            bad = torch.tensor([bad_here], dtype=torch.int64, device=cdev)
            self.dist.all_reduce(bad, op=self.dist.ReduceOp.MIN, group=self.group)
            bad_here = int(bad.item())
        def raise_bad():
            where = ""
            if not sharded_input:
                pt = [float(p[bad_here]) for p in points]
                where = f" {np.atleast_1d(pt)}"
            raise ValueError(f"Spline evaluation outside domain:{where} (flat index {bad_here})")
        # Without the agreement (check=False) only this rank knows about its offender: it must not
        # leave the others alone in the all-gather below, so it raises after the collective.
        deferred = bad_here != import_max and self.coll and gather and not check
        if bad_here != import_max and not deferred:
            raise_bad()
        if not gather or not self.coll:
            return res
        if deferred and res is None:
            rows = self.spline.nDep * (nind if op == "jacobian" else 1)
            dt = np.result_type(self.spline.coefs.dtype, *[k.dtype for k in self.spline.knots])
            if hasattr(local[0], "is_cuda"):
                is_t = True
                res = torch.full((rows, stop - start), float("nan"), dtype=getattr(torch, np.dtype(dt).name), device=local[0].device)
            else:
                res = np.full((rows, stop - start), np.nan, dtype=dt)

        # all-gather: ONE collective into (world, rows, chunk), then one device copy into the (rows, N) layout
        chunk = shard_chunk(n, self.world)
        rows = res.shape[0]
        t = res if is_t else torch.from_numpy(np.ascontiguousarray(res))
        if cdev is not None and t.device != cdev:
            t = t.to(cdev)
        if t.shape[1] != chunk:          # short tail shard: pad to the common chunk
            pad = torch.zeros((rows, chunk), dtype=t.dtype, device=t.device)
            pad[:, :t.shape[1]] = t
            t = pad
        gathered = torch.empty((self.world * rows, chunk), dtype=t.dtype, device=t.device)   # rank-major concatenation
        self.dist.all_gather_into_tensor(gathered, t.contiguous(), group=self.group)
        full = gathered.view(self.world, rows, chunk).permute(1, 0, 2).reshape(rows, self.world * chunk)[:, :n]
        if deferred:
            raise_bad()
        if is_t:
            return full
        return full.cpu().numpy() if full.device.type != "cpu" else full.numpy()
