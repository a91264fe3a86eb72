"""cfg4 positions job under the measurement knobs of bsk_tessellate (BSK_TESS_T / _W / _R, read at every call), on ONE
set of buffers in ONE process: the run-to-run spread of tools/cfg4_tess.py (0.25 - 0.33 ms for the same build) comes with the
placement of the 1.6 GB result, so variants are compared here on the same placement, interleaved, median of 5.
usage (GPU box): python tools/tess_sweep.py"""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, cases, bspy_amd
g = np.load(os.path.join(ROOT, "tests", "golden", "reference_tables.npz"))
patches = [bspy_amd.DeviceSpline(o, c, k, cf, np.float32) for (o, c, k, cf) in cases.teapot_patches(g)]
gg = torch.linspace(0, 1, 2048, dtype=torch.float32, device="cuda")
pos = torch.empty((32, 3, 2048, 2048), dtype=torch.float32, device="cuda")
f = lambda: bspy_amd.tessellate_tables(patches, (gg, gg), normals=False, out=(pos, None), check=False)
def timed(steps=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps
variants = [dict(T=256), dict(T=256, W=8), dict(T=512, W=8), dict(T=512, W=4), dict(T=512, W=2), dict(T=512, W=4, R=8), dict(T=512, W=4, R=2), dict()]
t_end = time.perf_counter() + 0.3
while time.perf_counter() < t_end: f()
res = {i: [] for i in range(len(variants))}
for rep in range(5):
    for i, v in enumerate(variants):
        for k in ("T", "W", "R"):
            os.environ.pop("BSK_TESS_" + k, None)
        for k, x in v.items():
            os.environ["BSK_TESS_" + k] = str(x)
        res[i].append(timed())
n = 32 * 2048 * 2048
for i, v in enumerate(variants):
    ms = statistics.median(res[i])
    print(f"{str(v):44s} median {ms:.4f} ms  (min {min(res[i]):.4f}, max {max(res[i]):.4f})  {12 * n / ms / 1e9:.2f} TB/s")
