"""fp32 bicubic 64 x 64 x 3 on 10 M random points (the 'surface o4 64x64 d3 f32' row of tools/bench_shapes.py) alone:
for rocprofv3 kernel-trace / PMC passes.  BSK_VARIANT=9 pins eval_rowrot<float> instead of eval_rec32."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, cases, bspy_amd
n = 10_000_000
rng = np.random.default_rng(3)
dt = np.float32
knots = [cases.clamped_uniform_knots(4, 64, dt)] * 2
t = bspy_amd.DeviceSpline((4, 4), (64, 64), knots, rng.standard_normal((3, 64, 64)).astype(dt), dt)
p = [torch.as_tensor(rng.random(n).astype(dt), device="cuda") for _ in range(2)]
o = torch.empty((3, n), dtype=torch.float32, device="cuda")
W, K = (int(x) for x in os.environ.get("ITERS", "30,30").split(","))
for _ in range(W): t.evaluate_device(p, out=o, check=False)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(K): t.evaluate_device(p, out=o, check=False)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / K
print(f"[{t.last_kernel()}] {ms * 1e3:.1f} us per 10 M points = {20 * n / ms / 1e6 / 8000:.3f} of the HBM roofline")
