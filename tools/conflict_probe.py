"""How much of the cfg2 kernel time is LDS bank conflicts?  Times eval_rowrot on 10 M points that are
(a) uniform random (the benchmark), (b) a sweep along v: consecutive lanes sit in consecutive spans, so
every half-wave reads 32 distinct banks (no conflicts in coefficient or table reads), (c) all equal
(every read a broadcast)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.cases import bench_spline
from bspy_amd import DeviceSpline

N = 10_000_000
nind, ndep, order, ncoef, knots, coefs, dt = bench_spline(2)
ds = DeviceSpline(order, ncoef, knots, coefs)
g = torch.Generator(device="cuda").manual_seed(5)
rnd = torch.rand((2, N), dtype=torch.float64, device="cuda", generator=g)
i = torch.arange(N, device="cuda")
sweep = torch.stack([((i // 61) % 61).double() / 61 + 0.3 / 61, (i % 61).double() / 61 + 0.3 / 61])
same = torch.full((2, N), 0.4321, dtype=torch.float64, device="cuda")
out = torch.empty((3, N), dtype=torch.float64, device="cuda")
for name, p in (("random", rnd), ("sweep", sweep), ("same", same), ("random", rnd), ("sweep", sweep)):
    pts = [p[0].contiguous(), p[1].contiguous()]
    for _ in range(300): ds.evaluate_device(pts, out=out, check=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): ds.evaluate_device(pts, out=out, check=False)
    e1.record(); torch.cuda.synchronize()
    print(f"{name:8s} {e0.elapsed_time(e1) * 10:.1f} us per launch")
