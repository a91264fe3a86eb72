"""cfg5 shape, fused-by-passes jacobian of 10 M points (nInd derivative passes through the cell-order pipeline, one sort)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, cases, bspy_amd
nind, ndep, order, ncoef, knots, coefs, dt = cases.bench_spline(5)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
t = bspy_amd.DeviceSpline(order, ncoef, knots, coefs, dt)
p = [torch.rand(n, dtype=torch.float32, device="cuda") for _ in range(3)]
jo = torch.empty((4, 3, n), dtype=torch.float32, device="cuda")
for _ in range(10): t.jacobian_device(p, out=jo, check=False)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): t.jacobian_device(p, out=jo, check=False)
e1.record(); torch.cuda.synchronize()
print(f"cfg5 jacobian {n} points: {e0.elapsed_time(e1) / 10:.4f} ms")
t.domain_status()
