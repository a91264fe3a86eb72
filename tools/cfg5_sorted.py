"""Experiment: how much does cell-sorted input help the L2-gathered cfg5 evaluation? (diagnostic)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, cases, bspy_amd
nind, ndep, order, ncoef, knots, coefs, dt = cases.bench_spline(5)
n = 10_000_000
t5 = bspy_amd.DeviceSpline(order, ncoef, knots, coefs, dt)
p = [torch.rand(n, dtype=torch.float32, device="cuda") for _ in range(3)]
o5 = torch.empty((4, n), dtype=torch.float32, device="cuda")
def timed(f, steps=5):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps
print("unsorted eval ms", timed(lambda: t5.evaluate_device(p, out=o5, check=False)) * 1e3)
# cell id on device with torch (uniform knots: 36 spans per axis)
def cell(): return ((p[0] * 36).long().clamp_(max=35) * 36 + (p[1] * 36).long().clamp_(max=35)) * 36 + (p[2] * 36).long().clamp_(max=35)
c = cell()
print("cell id ms", timed(cell) * 1e3)
print("sort ms", timed(lambda: torch.sort(c)) * 1e3)
order_idx = torch.sort(c).indices
ps = [x[order_idx].contiguous() for x in p]
print("gather params ms", timed(lambda: [x[order_idx] for x in p]) * 1e3)
print("sorted eval ms", timed(lambda: t5.evaluate_device(ps, out=o5, check=False)) * 1e3)
inv = torch.empty_like(order_idx); inv[order_idx] = torch.arange(n, device="cuda")
print("unsort results ms", timed(lambda: o5[:, inv]) * 1e3)
