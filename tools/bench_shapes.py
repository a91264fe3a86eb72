"""Throughput of the LDS-resident kernels on other shapes, 10 M points, device-resident I/O (diagnostic)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, cases, bspy_amd

def timed(f, steps=30):
    import time
    t_end = time.perf_counter() + 0.1
    while time.perf_counter() < t_end: f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e-3

n = 10_000_000
rng = np.random.default_rng(3)
shapes = [("curve o4 n32 d1 f64", (4,), (32,), 1, np.float64), ("curve o4 n32 d3 f64", (4,), (32,), 3, np.float64),
          ("curve o6 n100 d2 f64", (6,), (100,), 2, np.float64), ("surface o3 20x20 d3 f64", (3, 3), (20, 20), 3, np.float64),
          ("surface o4 64x64 d3 f32", (4, 4), (64, 64), 3, np.float32), ("surface o5 30x30 d3 f64", (5, 5), (30, 30), 3, np.float64),
          ("volume o4 8x9x10 d1 f64", (4, 4, 4), (8, 9, 10), 1, np.float64), ("volume o3 12^3 d3 f64", (3, 3, 3), (12, 12, 12), 3, np.float64)]
for name, order, ncoef, ndep, dt in shapes:
    knots = [cases.clamped_uniform_knots(o, c, dt) for o, c in zip(order, ncoef)]
    coefs = rng.standard_normal((ndep, *ncoef)).astype(dt)
    t = bspy_amd.DeviceSpline(order, ncoef, knots, coefs, dt)
    tdt = torch.float32 if dt == np.float32 else torch.float64
    p = [torch.as_tensor(rng.random(n).astype(dt), device="cuda") for _ in order]
    o = torch.empty((ndep, n), dtype=tdt, device="cuda")
    s = timed(lambda: t.evaluate_device(p, out=o, check=False))
    bpe = (len(order) + ndep) * np.dtype(dt).itemsize
    print(f"{name:28s} {s * 1e6:8.1f} us  {n / s / 1e9:7.1f} G evals/s  {bpe * n / s / 1e12:5.2f} TB/s = {bpe * n / s / 8e12:5.1%} of HBM roofline", flush=True)
    jo = torch.empty((ndep, len(order), n), dtype=tdt, device="cuda")
    s = timed(lambda: t.jacobian_device(p, out=jo, check=False))
    bpe = (len(order) + ndep * len(order)) * np.dtype(dt).itemsize
    print(f"{'  jacobian':28s} {s * 1e6:8.1f} us  {n / s / 1e9:7.1f} G evals/s  {bpe * n / s / 1e12:5.2f} TB/s = {bpe * n / s / 8e12:5.1%}", flush=True)
    t.domain_status(); del p, o, jo
