"""Memory-side floor of the evaluation kernels' launch geometry (diagnostic; GPU box only).
Streams 2 x 8 B in + 3 x 8 B out per point through bsk_debug_probe for several geometries."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import cases, bspy_amd
from bspy_amd import _native as nv

L = nv.lib()
L.bsk_debug_probe.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p,
                              ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
L.bsk_debug_probe.restype = ctypes.c_int
nind, ndep, order, ncoef, knots, coefs, dt = cases.bench_spline(2)
t = bspy_amd.DeviceSpline(order, ncoef, knots, coefs, dt)
n = 10_000_000
u = torch.rand(n, dtype=torch.float64, device="cuda"); v = torch.rand(n, dtype=torch.float64, device="cuda")
out = torch.empty((3, n), dtype=torch.float64, device="cuda")
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for mode in (0, 1):
    for bpc, threads, lds in ((1, 1024, 110000), (1, 1024, 0), (2, 1024, 0), (8, 256, 0), (1, 512, 110000)):
        def run():
            st = L.bsk_debug_probe(t._handle, mode, bpc, threads, lds, u.data_ptr(), v.data_ptr(), n, out.data_ptr(), stream)
            assert st == 0, L.bsk_last_error()
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"mode {mode} ({8 * (mode + 1)} B/lane)  blocks/CU {bpc} x {threads} thr  lds {lds:6d}:  {ms * 1e3:7.1f} us  {40 * n / ms / 1e6:7.1f} GB/s", flush=True)
ref = (u + v)
assert torch.equal(out[0], ref)
