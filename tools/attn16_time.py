"""Times the all-bf16 attention kernels beside the exact-fp32 ones at the training shapes (events on the launch stream).
usage: python tools/attn16_time.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagecaption_amd import ops

SHAPES = [("C3 cross", 256, 8, 20, 196, 128, False), ("C3 self", 256, 8, 20, 20, 128, True),
          ("C2 cross", 256, 8, 20, 49, 64, False), ("C2 self", 256, 8, 20, 20, 64, True),
          ("C1 self", 32, 8, 299, 299, 64, True)]


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for name, B, H, Tq, Tk, dh, causal in SHAPES:
    D = H * dh
    g = torch.Generator().manual_seed(1)
    q32 = torch.randn(B * Tq, 3 * D, generator=g).cuda()
    kv32 = torch.randn(B * Tk, 2 * D, generator=g).cuda()
    do32 = torch.randn(B * Tq, D, generator=g).cuda()
    q16, kv16, do16 = q32.bfloat16(), kv32.bfloat16(), do32.bfloat16()
    p, seed = float(os.environ.get("ATTN_P", "0.1")), 7
    o32, lse = ops.attn_fwd(q32[:, :D], kv32[:, :D], kv32[:, D:], B, H, Tq, Tk, causal, p, seed)
    o16, lse16 = ops.attn_fwd(q16[:, :D], kv16[:, :D], kv16[:, D:], B, H, Tq, Tk, causal, p, seed)
    dq32, dkv32 = torch.empty_like(do32), torch.empty_like(kv32)
    dq16, dkv16 = torch.empty_like(do16), torch.empty_like(kv16)
    f32 = timed(lambda: ops.attn_fwd(q32[:, :D], kv32[:, :D], kv32[:, D:], B, H, Tq, Tk, causal, p, seed, out=o32))
    f16 = timed(lambda: ops.attn_fwd(q16[:, :D], kv16[:, :D], kv16[:, D:], B, H, Tq, Tk, causal, p, seed, out=o16))
    b32 = timed(lambda: ops.attn_bwd(q32[:, :D], kv32[:, :D], kv32[:, D:], o32, do32, lse, B, H, Tq, Tk, causal, p, seed,
                                     dq=dq32, dk=dkv32[:, :D], dv=dkv32[:, D:]))
    b16 = timed(lambda: ops.attn_bwd(q16[:, :D], kv16[:, :D], kv16[:, D:], o16, do16, lse16, B, H, Tq, Tk, causal, p, seed,
                                     dq=dq16, dk=dkv16[:, :D], dv=dkv16[:, D:]))
    by_f = 2.0 * B * D * (2 * Tq + 2 * Tk)   # bf16 algorithmic bytes
    by_b = 2.0 * B * D * (4 * Tq + 4 * Tk)
    print(f"{name:9s} fwd fp32 {f32:7.1f} us  bf16 {f16:7.1f} us ({by_f / f16 / 1e6:5.2f} TB/s of bf16 bytes) | "
          f"bwd fp32 {b32:7.1f} us  bf16 {b16:7.1f} us ({by_b / b16 / 1e6:5.2f} TB/s)", flush=True)
