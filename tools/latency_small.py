"""Latency of small host-memory calls (the reference's single-point API goes through these)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, cases, bspy_amd
from bspy_amd import Spline
nind, ndep, order, ncoef, knots, coefs, dt = cases.bench_spline(2)
s = Spline(nind, ndep, order, ncoef, knots, coefs)
t = s.device_tables()
rng = np.random.default_rng(0)
for n in (1, 16, 256, 2048, 4096, 16384, 65536, 262144):
    u, v = rng.random(n), rng.random(n)
    for _ in range(50): t.evaluate([u, v])
    t0 = time.perf_counter()
    reps = 500 if n <= 4096 else 50
    for _ in range(reps): t.evaluate([u, v])
    dt_dev = (time.perf_counter() - t0) / reps
    print(f"DeviceSpline.evaluate host arrays n={n:5d}: {dt_dev * 1e6:8.1f} us per call")
for name, f in (("Spline.evaluate([u, v]) single point", lambda: s.evaluate([0.3, 0.7])),
                ("Spline.jacobian([u, v]) single point", lambda: s.jacobian([0.3, 0.7])),
                ("Spline.normal([u, v]) single point", lambda: s.normal([0.3, 0.7]))):
    for _ in range(50): f()
    t0 = time.perf_counter()
    for _ in range(500): f()
    print(f"{name}: {(time.perf_counter() - t0) / 500 * 1e6:.1f} us per call")
k = np.concatenate((np.zeros(4), np.linspace(0, 1, 30)[1:-1], np.ones(4)))
for _ in range(50): Spline.bspline_values(None, k, 4, 0.37)
t0 = time.perf_counter()
for _ in range(500): Spline.bspline_values(None, k, 4, 0.37)
print(f"Spline.bspline_values(None, knots, 4, u) single point: {(time.perf_counter() - t0) / 500 * 1e6:.1f} us per call")
