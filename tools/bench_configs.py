"""Timings of the other BASELINE configs on one GPU (diagnostic; GPU box only):
cfg3 derivative + jacobian (10 M points), cfg4 teapot tessellation grid, cfg5 trivariate fp32."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import cases, bspy_amd

def timed(f, steps=20, warm=3, spin_s=0.15):
    import time as _t
    t_end = _t.perf_counter() + spin_s          # clock spin-up (see bench.py)
    while _t.perf_counter() < t_end:
        f()
    torch.cuda.synchronize()
    for _ in range(warm): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e-3

res = {}
# cfg2 / cfg3
nind, ndep, order, ncoef, knots, coefs, dt = cases.bench_spline(2)
n = 10_000_000
t = bspy_amd.DeviceSpline(order, ncoef, knots, coefs, dt)
u = torch.rand(n, dtype=torch.float64, device="cuda"); v = torch.rand(n, dtype=torch.float64, device="cuda")
out = torch.empty((3, n), dtype=torch.float64, device="cuda"); jout = torch.empty((3, 2, n), dtype=torch.float64, device="cuda")
for name, f, bpe in (("cfg2_evaluate", lambda: t.evaluate_device([u, v], out=out, check=False), 40),
                     ("cfg3_derivative_1_1", lambda: t.evaluate_device([u, v], [1, 1], out=out, check=False), 40),
                     ("cfg3_jacobian", lambda: t.jacobian_device([u, v], out=jout, check=False), 64)):
    s = timed(f)
    res[name] = {"ms": round(s * 1e3, 4), "Mevals_s": round(n / s / 1e6, 1), "GBs": round(bpe * n / s / 1e9, 1)}
    print(name, res[name], flush=True)
t.domain_status()
del u, v, out, jout
# cfg4: teapot patch, 2048 x 2048 grid, fp32 (all 32 patches share the shape; time one and report x32)
g = np.load(os.path.join(ROOT, "tests", "golden", "reference_tables.npz"))
knots4 = np.array((0, 0, 0, 0, 1, 1, 1, 1), np.float32)
V = g["teapot_vertices"]; patch = g["teapot_patch_index"][0]
c = np.empty((3, 4, 4), np.float32)
for i in range(4):
    for j in range(4):
        vv = V[patch[4 * i + j] - 1]; c[0, i, j], c[1, i, j], c[2, i, j] = vv[0], vv[2], vv[1]
t4 = bspy_amd.DeviceSpline((4, 4), (4, 4), (knots4, knots4), c, np.float32)
gg = torch.linspace(0, 1, 2048, dtype=torch.float32, device="cuda")
o4 = torch.empty((3, 2048, 2048), dtype=torch.float32, device="cuda")
s = timed(lambda: t4.evaluate_grid_device([gg, gg], out=o4, check=False))
npt = 2048 * 2048
res["cfg4_teapot_patch_grid2048"] = {"ms": round(s * 1e3, 4), "Mevals_s": round(npt / s / 1e6, 1), "GBs": round(12 * npt / s / 1e9, 1),
                                      "all_32_patches_ms": round(32 * s * 1e3, 3)}
print("cfg4", res["cfg4_teapot_patch_grid2048"], flush=True)
t4.domain_status()
# cfg4, whole job: all 32 patches from one call (bsk_tessellate), positions only and positions + unit normals
pt = [bspy_amd.DeviceSpline(o, c_, k, cf, np.float32) for (o, c_, k, cf) in cases.teapot_patches(g)]
pos = torch.empty((32, 3, 2048, 2048), dtype=torch.float32, device="cuda")
nrm = torch.empty_like(pos)
s = timed(lambda: bspy_amd.tessellate_tables(pt, (gg, gg), normals=False, out=(pos, None), check=False), steps=10, warm=2)
res["cfg4_teapot_32_patches_positions"] = {"ms": round(s * 1e3, 4), "Mevals_s": round(32 * npt / s / 1e6, 1), "GBs": round(32 * 12 * npt / s / 1e9, 1)}
print("cfg4 batch", res["cfg4_teapot_32_patches_positions"], flush=True)
s = timed(lambda: bspy_amd.tessellate_tables(pt, (gg, gg), out=(pos, nrm), check=False), steps=10, warm=2)
res["cfg4_teapot_32_patches_positions_normals"] = {"ms": round(s * 1e3, 4), "Mevals_s": round(32 * npt / s / 1e6, 1), "GBs": round(32 * 24 * npt / s / 1e9, 1)}
print("cfg4 batch + normals", res["cfg4_teapot_32_patches_positions_normals"], flush=True)
pt[0].domain_status()
del pos, nrm
# cfg5: trivariate order 5, 40^3, nDep 4, fp32, 50 M points total over 8 GPUs -> 6.25 M per GPU; time 10 M here
nind, ndep, order, ncoef, knots, coefs, dt = cases.bench_spline(5)
n5 = 10_000_000
t5 = bspy_amd.DeviceSpline(order, ncoef, knots, coefs, dt)
p5 = [torch.rand(n5, dtype=torch.float32, device="cuda") for _ in range(3)]
o5 = torch.empty((4, n5), dtype=torch.float32, device="cuda")
s = timed(lambda: t5.evaluate_device(p5, out=o5, check=False), steps=5, warm=2)
res["cfg5_trivariate_f32"] = {"ms": round(s * 1e3, 4), "Mevals_s": round(n5 / s / 1e6, 1), "GBs": round(28 * n5 / s / 1e9, 1)}
print("cfg5", res["cfg5_trivariate_f32"], flush=True)
t5.domain_status()
print(json.dumps(res))
