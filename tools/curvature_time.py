"""Batched Gaussian curvature on the cfg2 surface, 10 M points, device-resident (diagnostic)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, cases, bspy_amd
nind, ndep, order, ncoef, knots, coefs, dt = cases.bench_spline(2)
n = 10_000_000
t = bspy_amd.DeviceSpline(order, ncoef, knots, coefs, dt)
uv = np.random.default_rng(2).random((2, n))
u, v = torch.as_tensor(uv[0], device="cuda"), torch.as_tensor(uv[1], device="cuda")
o = torch.empty(n, dtype=torch.float64, device="cuda")
for _ in range(20): t.curvature_device([u, v], out=o, check=False)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): t.curvature_device([u, v], out=o, check=False)
e1.record(); torch.cuda.synchronize()
print(f"curvature: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per 10 M points")
