"""Does the depth of the launch queue change the kernel time?  cfg2 evaluate, 200 launches each:
(a) direct C-ABI calls back to back (bench.py), (b) through the Python wrapper, (c) direct calls with
a synchronize every 20 launches, (d) direct calls, each followed by ~20 us of host spinning."""
import os, sys, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import cases, bspy_amd
from bspy_amd import _native as nv

nind, ndep, order, ncoef, knots, coefs, dt = cases.bench_spline(2)
n = 10_000_000
t = bspy_amd.DeviceSpline(order, ncoef, knots, coefs, dt)
uv = np.random.default_rng(1000).random((2, n))
u = torch.as_tensor(uv[0], device="cuda"); v = torch.as_tensor(uv[1], device="cuda")
out = torch.empty((3, n), dtype=torch.float64, device="cuda")
lib = nv.lib()
ptrs = nv.ptr_array([u.data_ptr(), v.data_ptr()])
sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
op = ctypes.c_void_p(out.data_ptr())

def direct(): lib.bsk_evaluate(t._handle, None, ptrs, n, nv.BSK_DEVICE, op, sp, None)
def wrapper(): t.evaluate_device([u, v], out=out, check=False)
def spin():
    direct()
    e = time.perf_counter() + 20e-6
    while time.perf_counter() < e: pass

def run(name, f, steps=200, sync_every=0):
    for _ in range(300): direct()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(steps):
        f()
        if sync_every and (i + 1) % sync_every == 0: torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:28s} {e0.elapsed_time(e1) / steps * 1e3:.1f} us per step", flush=True)

for _ in range(2):
    run("direct back to back", direct)
    run("python wrapper", wrapper)
    run("direct, sync every 20", direct, sync_every=20)
    run("direct + 20 us host spin", spin)
    run("direct, 20 steps", direct, steps=20)
