"""cfg4 whole job: the 32 teapot patches on a 2048 x 2048 grid from one bsk_tessellate call (positions, positions + normals)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, cases, bspy_amd
g = np.load(os.path.join(ROOT, "tests", "golden", "reference_tables.npz"))
patches = [bspy_amd.DeviceSpline(o, c, k, cf, np.float32) for (o, c, k, cf) in cases.teapot_patches(g)]
gg = torch.linspace(0, 1, 2048, dtype=torch.float32, device="cuda")
pos = torch.empty((32, 3, 2048, 2048), dtype=torch.float32, device="cuda")
nrm = torch.empty((32, 3, 2048, 2048), dtype=torch.float32, device="cuda")
def timed(f, steps=20):
    t_end = time.perf_counter() + 0.1
    while time.perf_counter() < t_end: f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps
n = 32 * 2048 * 2048
ms = timed(lambda: bspy_amd.tessellate_tables(patches, (gg, gg), normals=False, out=(pos, None), check=False))
print(f"positions           {ms:.4f} ms  {12 * n / ms / 1e9:.2f} TB/s = {12 * n / ms / 1e9 / 8:.1%} of the HBM roofline")
ms = timed(lambda: bspy_amd.tessellate_tables(patches, (gg, gg), normals=True, out=(pos, nrm), check=False))
print(f"positions + normals {ms:.4f} ms  {24 * n / ms / 1e9:.2f} TB/s = {24 * n / ms / 1e9 / 8:.1%}")
