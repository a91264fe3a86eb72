"""Write-only memory floor: torch's fill kernel and hipMemsetAsync on 1.6 GB (the bytes the cfg4 tessellation writes).
usage (GPU box): python tools/write_floor.py"""
import torch
n = 32 * 3 * 2048 * 2048
x = torch.empty(n, dtype=torch.float32, device="cuda")
for name, f in (("fill_", lambda: x.fill_(1.5)), ("zero_ (memset)", lambda: x.zero_())):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{name:16s} {4 * n / 1e9:.2f} GB in {ms * 1e3:.1f} us = {4 * n / ms / 1e9:.2f} TB/s")
