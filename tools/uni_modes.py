"""The cfg2 workload on the uniform-knot kernel (argument 0) and on the general kernels (v9 = BSK_VARIANT=9):
error against the C oracle on 200 k points, then steady-state time per 10 M-point launch on random points
and on a conflict-free sweep (consecutive lanes in consecutive spans).  During round 2 the arguments selected
build variants of eval_uni (BSK_UNI_MODE bits, since folded into the one kernel: DESIGN.md section 5.1).
usage: python tools/uni_modes.py 0 v9 0"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np, torch
    from cases import bench_spline
    import oracle
    from bspy_amd import DeviceSpline
    N = 10_000_000
    nind, ndep, order, ncoef, knots, coefs, dt = bench_spline(2)
    ds = DeviceSpline(order, ncoef, knots, coefs)
    g = torch.Generator(device="cuda").manual_seed(5)
    rnd = torch.rand((2, N), dtype=torch.float64, device="cuda", generator=g)
    small = [rnd[0][:200_000].contiguous(), rnd[1][:200_000].contiguous()]
    got = ds.evaluate_device(small).cpu().numpy()
    ref, _ = oracle.c_evaluate(order, ncoef, knots, coefs, [0, 0], [p.cpu().numpy() for p in small])
    err = float(np.abs(got - ref).max())
    i = torch.arange(N, device="cuda")
    sweep = torch.stack([((i // 61) % 61).double() / 61 + 0.3 / 61, (i % 61).double() / 61 + 0.3 / 61])
    out = torch.empty((3, N), dtype=torch.float64, device="cuda")
    res = []
    for name, p in (("random", rnd), ("sweep", sweep), ("random", rnd), ("sweep", sweep)):
        pts = [p[0].contiguous(), p[1].contiguous()]
        for _ in range(300): ds.evaluate_device(pts, out=out, check=False)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): ds.evaluate_device(pts, out=out, check=False)
        e1.record(); torch.cuda.synchronize()
        res.append(f"{name} {e0.elapsed_time(e1) * 5:.1f}")
    print(f"{sys.argv[2]:>4s} kernel {ds.last_kernel():12s} err {err:.2e}  us/launch: " + "  ".join(res), flush=True)
    sys.exit(0)
for mode in sys.argv[1:]:
    env = dict(os.environ)
    if mode.startswith("v"):
        env["BSK_VARIANT"] = mode[1:]
    else:
        env["BSK_UNI_MODE"] = mode
    subprocess.run([sys.executable, __file__, "--child", mode], env=env, check=False)
