"""A/B two BSK_VARIANT settings on the cfg2 workload: bitwise comparison of the results, then timing.
usage: python tools/ab_variant.py 9 10 [op]"""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import numpy as np, torch
    from tests.cases import bench_spline
    from bspy_amd import DeviceSpline
    nind, ndep, order, ncoef, knots, coefs, dt = bench_spline(2)
    ds = DeviceSpline(order, ncoef, knots, coefs)
    g = torch.Generator(device="cuda").manual_seed(5)
    uv = torch.rand((2, 2_000_003), dtype=torch.float64, device="cuda", generator=g)
    out = ds.evaluate_device([uv[0], uv[1]])
    d = ds.evaluate_device([uv[0], uv[1]], wrt=(1, 0))
    torch.cuda.synchronize()
    np.save(sys.argv[2], torch.cat([out, d]).cpu().numpy())
    sys.exit(0)
va, vb = sys.argv[1], sys.argv[2]
op = sys.argv[3] if len(sys.argv) > 3 else "evaluate"
import numpy as np
outs = []
for v in (va, vb):
    f = f"/tmp/ab_{v}.npy"
    subprocess.run([sys.executable, __file__, "--child", f], env=dict(os.environ, BSK_VARIANT=v), check=True)
    outs.append(np.load(f))
print("bitwise equal:", np.array_equal(outs[0], outs[1]), "max abs diff:", float(np.abs(outs[0] - outs[1]).max()))
for v in (va, vb, va, vb):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu", "--op", op, "--steps", "100", "--warmup", "5"],
                       env=dict(os.environ, BSK_VARIANT=v), capture_output=True, text=True)
    j = json.loads(r.stdout.strip().splitlines()[-1])
    print("variant", v, op, "ms_per_step", j["ms_per_step"], "frac", j["roofline"]["frac"])
