"""Streaming-copy rate of capdec_probe_copy at several sizes (read + write bytes / time): the large sizes give the
chip's achievable HBM rate, the small ones what ANY kernel moving that few bytes can reach (launch + ramp)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagecaption_amd._capi import check, lib, ptr, stream
for mb in (21, 36, 72, 145, 512, 1024):
    n = mb << 20
    src = torch.zeros(n, dtype=torch.uint8, device="cuda"); dst = torch.empty_like(src)
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): check(lib().capdec_probe_copy(ptr(src), ptr(dst), n, stream()), "copy")
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10 * 1e-3)
    print(f"copy {mb:5d} MiB (x2 moved): {best*1e6:8.1f} us  {2*n/best/1e12:5.2f} TB/s", flush=True)
