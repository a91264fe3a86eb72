"""Where does a bench step's time go: kernel duration (event pair around every launch) vs
wall time per step of a back-to-back launch loop (diagnostic; GPU box only)."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import cases, bspy_amd
from bspy_amd import _native as nv
nind, ndep, order, ncoef, knots, coefs, dt = cases.bench_spline(2)
n = 10_000_000
t = bspy_amd.DeviceSpline(order, ncoef, knots, coefs, dt)
u = torch.rand(n, dtype=torch.float64, device="cuda"); v = torch.rand(n, dtype=torch.float64, device="cuda")
out = torch.empty((3, n), dtype=torch.float64, device="cuda")
lib = nv.lib(); ptrs = nv.ptr_array([u.data_ptr(), v.data_ptr()])
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream); outp = ctypes.c_void_p(out.data_ptr())
def step(): lib.bsk_evaluate(t._handle, None, ptrs, n, 1, outp, stream, None)
for _ in range(5): step()
torch.cuda.synchronize()
K = 50
t0 = time.perf_counter()
for _ in range(K): step()
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"host issue time per launch {t_issue / K * 1e6:.1f} us, wall per step {t_all / K * 1e6:.1f} us")
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
for a, b in evs:
    a.record(); step(); b.record()
torch.cuda.synchronize()
d = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
print(f"per-launch event duration: min {d[0]:.1f} median {d[K // 2]:.1f} max {d[-1]:.1f} us")
# same loop with a different random batch each step (no L2/MALL reuse of the inputs)
us = [torch.rand(n, dtype=torch.float64, device="cuda") for _ in range(4)]
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(K):
    p = nv.ptr_array([us[i % 4].data_ptr(), us[(i + 1) % 4].data_ptr()])
    lib.bsk_evaluate(t._handle, None, p, n, 1, outp, stream, None)
torch.cuda.synchronize()
print(f"rotating inputs: wall per step {(time.perf_counter() - t0) / K * 1e6:.1f} us")

outs = [torch.empty((3, n), dtype=torch.float64, device="cuda") for _ in range(3)]
torch.cuda.synchronize()
for name, nin, nout in (("same in, same out", 1, 1), ("4 in, same out", 4, 1), ("same in, 3 out", 1, 3), ("4 in, 3 out", 4, 3), ("2 in, same out", 2, 1)):
    args = []
    for i in range(12):
        args.append((nv.ptr_array([us[i % nin].data_ptr(), us[(i + 1) % nin if nin > 1 else 0].data_ptr()]) if nin > 1 else ptrs,
                     ctypes.c_void_p(outs[i % nout].data_ptr())))
    for i in range(6): lib.bsk_evaluate(t._handle, None, args[i][0], n, 1, args[i][1], stream, None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(60):
        a = args[i % 12]
        lib.bsk_evaluate(t._handle, None, a[0], n, 1, a[1], stream, None)
    torch.cuda.synchronize()
    print(f"{name}: wall per step {(time.perf_counter() - t0) / 60 * 1e6:.1f} us", flush=True)
