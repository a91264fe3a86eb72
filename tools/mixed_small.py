"""Order-(3,4) LDS-resident surface: eval_slab2 in one slab against eval_mixed (BSK_VARIANT=7) over batch sizes.
usage (GPU box): python tools/mixed_small.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, cases, bspy_amd
rng = np.random.default_rng(1)
order, ncoef = (3, 4), (20, 20)
knots = [cases.clamped_uniform_knots(o, c) for o, c in zip(order, ncoef)]
coefs = rng.standard_normal((3, *ncoef))
t = bspy_amd.DeviceSpline(order, ncoef, knots, coefs)
os.environ["BSK_VARIANT"] = "7"
m = bspy_amd.DeviceSpline(order, ncoef, knots, coefs)
del os.environ["BSK_VARIANT"]
def timed(f, steps=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3
for n in (70_000, 150_000, 300_000, 600_000, 1_200_000, 2_500_000, 5_000_000, 10_000_000):
    p = [torch.rand(n, dtype=torch.float64, device="cuda") for _ in range(2)]
    o = torch.empty((3, n), dtype=torch.float64, device="cuda")
    a = timed(lambda: t.evaluate_device(p, out=o, check=False)); ka = t.last_kernel()
    b = timed(lambda: m.evaluate_device(p, out=o, check=False)); kb = m.last_kernel()
    print(f"n {n:9d}: {ka} {a:7.1f} us   {kb} {b:7.1f} us")
