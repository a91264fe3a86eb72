"""cfg5 (trivariate order 5, 40^3 x 4 fp32, 10 M random points) on the variants of the cell-order pipeline:
BSK_VARIANT 0 = eval_cellsort MFMA, 12 = eval_cellsort VALU, 13 = eval_binned_lds (round 1), 7 = gather."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np, torch, cases, bspy_amd, oracle
    nind, ndep, order, ncoef, knots, coefs, dt = cases.bench_spline(5)
    n = 10_000_000
    t = bspy_amd.DeviceSpline(order, ncoef, knots, coefs, dt)
    g = torch.Generator(device="cuda").manual_seed(9)
    p = [torch.rand(n, dtype=torch.float32, device="cuda", generator=g) for _ in range(3)]
    o = torch.empty((4, n), dtype=torch.float32, device="cuda")
    t.evaluate_device(p, out=o, check=False)
    sm = [x[:100_000].cpu().numpy() for x in p]
    ref, _ = oracle.c_evaluate(order, ncoef, knots, coefs, [0, 0, 0], sm)
    err = float(np.abs(o[:, :100_000].cpu().numpy() - ref).max() / max(1.0, np.abs(ref).max()))
    for _ in range(20): t.evaluate_device(p, out=o, check=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): t.evaluate_device(p, out=o, check=False)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"variant {sys.argv[2]:>3s} {t.last_kernel():48s} rel err {err:.1e}  {ms:.4f} ms per 10 M points", flush=True)
    sys.exit(0)
for v in sys.argv[1:]:
    subprocess.run([sys.executable, __file__, "--child", v], env=dict(os.environ, BSK_VARIANT=v), check=False)
