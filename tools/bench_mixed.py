import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch, cases, bspy_amd, time
def timed(f, steps=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e-3
n = 10_000_000
rng = np.random.default_rng(3)
for name, order, ncoef, ndep in (("surface 3x4 20x20", (3, 4), (20, 20), 3), ("surface 4x5 900x11 (TomsNasty shape)", (4, 5), (900, 11), 3),
                                 ("surface 4x4 20x20 (same order)", (4, 4), (20, 20), 3), ("curve o9", (9,), (30,), 2), ("curve o12", (12,), (30,), 2), ("surface 7x3 11x20", (7, 3), (11, 20), 3), ("surface 8x8 20x20", (8, 8), (20, 20), 3), ("4 variables o3", (3, 3, 3, 3), (6, 6, 6, 6), 2)):
    knots = [cases.clamped_uniform_knots(o, c) for o, c in zip(order, ncoef)]
    coefs = rng.standard_normal((ndep, *ncoef))
    t = bspy_amd.DeviceSpline(order, ncoef, knots, coefs)
    p = [torch.as_tensor(rng.random(n), device="cuda") for _ in order]
    o = torch.empty((ndep, n), dtype=torch.float64, device="cuda")
    s = timed(lambda: t.evaluate_device(p, out=o, check=False))
    print(f"{name:40s} {s*1e6:9.1f} us  {n/s/1e9:6.1f} G evals/s", flush=True)
    if len(order) == 2:
        jo = torch.empty((ndep, 2, n), dtype=torch.float64, device="cuda")
        s = timed(lambda: t.jacobian_device(p, out=jo, check=False))
        print(f"{'  jacobian':40s} {s*1e6:9.1f} us  {n/s/1e9:6.1f} G evals/s", flush=True)
        del jo
# grids: s(u[:, None], v[None, :]) on 2048 x 2048
gg = torch.linspace(0, 1, 2048, dtype=torch.float64, device="cuda")
for name, order, ncoef in (("grid 4x4 20x20", (4, 4), (20, 20)), ("grid 3x4 20x20", (3, 4), (20, 20)), ("grid 2x5 8x30", (2, 5), (8, 30))):
    knots = [cases.clamped_uniform_knots(o, c) for o, c in zip(order, ncoef)]
    t = bspy_amd.DeviceSpline(order, ncoef, knots, rng.standard_normal((3, *ncoef)))
    o = torch.empty((3, 2048, 2048), dtype=torch.float64, device="cuda")
    s = timed(lambda: t.evaluate_grid_device([gg, gg], out=o, check=False))
    print(f"{name:40s} {s*1e6:9.1f} us  {2048*2048/s/1e9:6.1f} G evals/s", flush=True)
