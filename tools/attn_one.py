"""Run the C2 cross-attention forward + backward a few times (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagecaption_amd import ops
B, H, T, S, d = 256, 8, 20, 49, 512
g = torch.Generator().manual_seed(0)
q = torch.randn(B * T, d, generator=g).cuda(); kv = torch.randn(B * S, 2 * d, generator=g).cuda(); do = torch.randn(B * T, d, generator=g).cuda()
for _ in range(5):
    o, lse = ops.attn_fwd(q, kv[:, :d], kv[:, d:], B, H, T, S, False, 0.1, 7)
    ops.attn_bwd(q, kv[:, :d], kv[:, d:], o, do, lse, B, H, T, S, False, 0.1, 7)
torch.cuda.synchronize()
