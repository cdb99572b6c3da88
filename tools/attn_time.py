"""Per-launch time of the attention kernels at the C2 / C1 / C3 shapes (HIP events, 20 launches each).
ATTN_P = dropout probability (default 0.1, the training value)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagecaption_amd import ops

def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

P = float(os.environ.get("ATTN_P", "0.1"))
g = torch.Generator().manual_seed(0)
for name, B, H, T, S, d, causal in (("c2 cross", 256, 8, 20, 49, 512, False), ("c2 self", 256, 8, 20, 20, 512, True),
                                    ("c1 self", 32, 8, 299, 299, 512, True), ("c3 cross", 256, 8, 20, 196, 1024, False),
                                    ("c3 self", 256, 8, 20, 20, 1024, True)):
    q = torch.randn(B * T, d, generator=g).cuda(); kv = torch.randn(B * S, 2 * d, generator=g).cuda(); do = torch.randn(B * T, d, generator=g).cuda()
    o, lse = ops.attn_fwd(q, kv[:, :d], kv[:, d:], B, H, T, S, causal, P, 7)
    tf = t(lambda: ops.attn_fwd(q, kv[:, :d], kv[:, d:], B, H, T, S, causal, P, 7))
    tb = t(lambda: ops.attn_bwd(q, kv[:, :d], kv[:, d:], o, do, lse, B, H, T, S, causal, P, 7))
    fb = 4.0 * B * d * (2 * T + 2 * S); bb = 4.0 * B * d * (4 * T + 4 * S)
    print(f"p {P} {name:9s} fwd {tf:7.1f} us {fb/tf/1e6:6.2f} TB/s | bwd {tb:7.1f} us {bb/tb/1e6:6.2f} TB/s", flush=True)
