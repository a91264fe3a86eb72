"""Per-launch fixed cost of the cfg2 kernel: steady-state time per launch for batches of k x (256 CUs x 1024
lanes) points (k iterations per wave), least-squares line T(N) = fixed + slope * N."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.cases import bench_spline
from bspy_amd import DeviceSpline
nind, ndep, order, ncoef, knots, coefs, dt = bench_spline(2)
ds = DeviceSpline(order, ncoef, knots, coefs)
g = torch.Generator(device="cuda").manual_seed(5)
NMAX = 10_000_000
rnd = torch.rand((2, NMAX), dtype=torch.float64, device="cuda", generator=g)
out = torch.empty((3, NMAX), dtype=torch.float64, device="cuda")
full = [rnd[0].contiguous(), rnd[1].contiguous()]
for _ in range(300): ds.evaluate_device(full, out=out, check=False)
xs, ys = [], []
for k in (1, 2, 4, 8, 16, 24, 38):
    n = min(k * 256 * 1024, NMAX)
    pts = [rnd[0][:n].contiguous(), rnd[1][:n].contiguous()]
    o = out[:, :n]
    o = torch.empty((3, n), dtype=torch.float64, device="cuda")
    for _ in range(200): ds.evaluate_device(pts, out=o, check=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(400): ds.evaluate_device(pts, out=o, check=False)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 2.5
    xs.append(n); ys.append(t)
    print(f"{ds.last_kernel()} n {n:9d} ({k:2d} iterations/wave) {t:7.2f} us/launch  {40 * n / t / 1e6:.2f} TB/s", flush=True)
a, b = np.polyfit(np.array(xs, float), np.array(ys), 1)
print(f"fit: fixed {b:.2f} us + {a * 1e7:.2f} us per 10 M points")
