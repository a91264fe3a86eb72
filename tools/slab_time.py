"""eval_slab2 on the TomsNasty shape (order (4, 5), nCoef (900, 11), nDep 3, fp64), 10 M points; BSK_SLAB_DBG switches
phases off for timing (results are then wrong).  usage: python tools/slab_time.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, cases, bspy_amd
n = 10_000_000
rng = np.random.default_rng(3)
order, ncoef, ndep = (4, 5), (900, 11), 3
knots = [cases.clamped_uniform_knots(o, c) for o, c in zip(order, ncoef)]
t = bspy_amd.DeviceSpline(order, ncoef, knots, rng.standard_normal((ndep, *ncoef)))
p = [torch.as_tensor(rng.random(n), device="cuda") for _ in order]
o = torch.empty((ndep, n), dtype=torch.float64, device="cuda")
for _ in range(5): t.evaluate_device(p, out=o, check=False)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): t.evaluate_device(p, out=o, check=False)
e1.record(); torch.cuda.synchronize()
print(f"BSK_SLAB_DBG={os.environ.get('BSK_SLAB_DBG', '0')}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us  [{t.last_kernel()}]")
