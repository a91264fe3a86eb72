import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, bspy_amd
g = np.load(os.path.join(ROOT, "tests", "golden", "reference_tables.npz"))
knots4 = np.array((0, 0, 0, 0, 1, 1, 1, 1), np.float32)
V = g["teapot_vertices"]; patch = g["teapot_patch_index"][0]
c = np.empty((3, 4, 4), np.float32)
for i in range(4):
    for j in range(4):
        vv = V[patch[4 * i + j] - 1]; c[0, i, j], c[1, i, j], c[2, i, j] = vv[0], vv[2], vv[1]
t4 = bspy_amd.DeviceSpline((4, 4), (4, 4), (knots4, knots4), c, np.float32)
gg = torch.linspace(0, 1, 2048, dtype=torch.float32, device="cuda")
o4 = torch.empty((3, 2048, 2048), dtype=torch.float32, device="cuda")
for _ in range(300): t4.evaluate_grid_device([gg, gg], out=o4, check=False)
torch.cuda.synchronize()
