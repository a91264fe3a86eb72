"""PCIe-inclusive rate of the BSK_HOST (NumPy in / NumPy out) path on cfg2 (diagnostic)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, cases, bspy_amd
nind, ndep, order, ncoef, knots, coefs, dt = cases.bench_spline(2)
n = 10_000_000
uv = np.random.default_rng(0).random((2, n))
t = bspy_amd.DeviceSpline(order, ncoef, knots, coefs, dt)
t.evaluate([uv[0], uv[1]])
best = 1e9
for _ in range(5):
    t0 = time.perf_counter(); t.evaluate([uv[0], uv[1]]); best = min(best, time.perf_counter() - t0)
print(f"host path: {best * 1e3:.1f} ms per 10 M points = {n / best / 1e6:.0f} M evals/s, {400e6 / best / 1e9:.1f} GB/s of host traffic")
# the same call writing into a caller-owned, already touched result array (no first-touch page faults)
import ctypes
from bspy_amd import _native as nv
out = np.empty((3, n)); out[:] = 0.0
ptrs = nv.ptr_array([uv[0].ctypes.data, uv[1].ctypes.data]); bad = ctypes.c_int64(-1)
best = 1e9
for _ in range(5):
    t0 = time.perf_counter()
    nv.lib().bsk_evaluate(t._handle, None, ptrs, n, nv.BSK_HOST, out.ctypes.data, None, ctypes.byref(bad))
    best = min(best, time.perf_counter() - t0)
print(f"host path, result array reused: {best * 1e3:.1f} ms per 10 M points = {n / best / 1e6:.0f} M evals/s, {400e6 / best / 1e9:.1f} GB/s")
t0 = time.perf_counter(); z = np.empty((3, n)); z[:] = 0.0; print(f"(first touch of a fresh 240 MB array alone: {(time.perf_counter() - t0) * 1e3:.1f} ms)")
