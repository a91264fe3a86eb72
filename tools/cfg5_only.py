"""cfg5 alone (trivariate order 5, 40^3 x 4 fp32, 10 M random points): for rocprofv3 --kernel-trace --stats."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, cases, bspy_amd
nind, ndep, order, ncoef, knots, coefs, dt = cases.bench_spline(5)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
t = bspy_amd.DeviceSpline(order, ncoef, knots, coefs, dt)
p = [torch.rand(n, dtype=torch.float32, device="cuda") for _ in range(3)]
o = torch.empty((4, n), dtype=torch.float32, device="cuda")
W, K = (int(x) for x in os.environ.get("CFG5_ITERS", "30,20").split(","))     # fewer launches for PMC passes
for _ in range(W): t.evaluate_device(p, out=o, check=False)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(K): t.evaluate_device(p, out=o, check=False)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / K
print(f"cfg5 {n} points: {ms:.4f} ms = {n / ms / 1e3:.1f} M evals/s, {28 * n / ms / 1e6:.1f} GB/s algorithmic")
t.domain_status()
