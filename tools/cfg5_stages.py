"""cfg5 (trivariate order 5, 40^3 x 4 fp32): per-kernel times of the cell-order pipeline from HIP events between its
kernels (bsk_debug_stage_times, the BSK_INTERNAL measurement hook), and the whole call.
usage: python tools/cfg5_stages.py [points] [reps] [--jac]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, cases, bspy_amd
from bspy_amd import _native as nv


def stage_times(tables, fn, reps=10):
    """median per-stage ms over `reps` recorded calls: {name: ms}"""
    lib = nv.lib()
    ms = (ctypes.c_float * 16)()
    names = (ctypes.c_char_p * 16)()
    cnt = ctypes.c_int(0)
    nv.check(lib.bsk_debug_stage_times(tables._handle, 1, None, None, 0, None))
    acc = {}
    for _ in range(reps):
        fn()
        nv.check(lib.bsk_debug_stage_times(tables._handle, 1, ms, names, 16, ctypes.byref(cnt)))
        for i in range(cnt.value):
            acc.setdefault(names[i].decode(), []).append(ms[i])
    nv.check(lib.bsk_debug_stage_times(tables._handle, 0, None, None, 0, None))
    return {k: float(np.median(v)) for k, v in acc.items()}


if __name__ == "__main__":
    jac = "--jac" in sys.argv
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    n = int(args[0]) if len(args) > 0 else 10_000_000
    reps = int(args[1]) if len(args) > 1 else 20
    nind, ndep, order, ncoef, knots, coefs, dt = cases.bench_spline(5)
    t = bspy_amd.DeviceSpline(order, ncoef, knots, coefs, dt)
    p = [torch.rand(n, dtype=torch.float32, device="cuda") for _ in range(3)]
    o = torch.empty((4, 3, n) if jac else (4, n), dtype=torch.float32, device="cuda")
    f = (lambda: t.jacobian_device(p, out=o, check=False)) if jac else (lambda: t.evaluate_device(p, out=o, check=False))
    for _ in range(30): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"cfg5 {n} points [{t.last_kernel()}]: {ms:.4f} ms per call = {n / ms / 1e3:.1f} M evals/s = {28 * n / ms / 1e6 / 8000:.4f} of the HBM roofline")
    st = stage_times(t, f, reps)
    print("  " + "  ".join(f"{k} {v * 1e3:.1f} us" for k, v in st.items()) + f"  (sum {sum(st.values()) * 1e3:.1f} us)")
    t.domain_status()
