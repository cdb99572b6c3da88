"""One shape of the all-bf16 attention, a few forward + backward calls (for rocprofv3 --pmc passes).
usage: python tools/attn16_one.py [B H Tq Tk dh causal reps]   (default: the C3 cross-attention shape)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagecaption_amd import ops

a = [int(x) for x in sys.argv[1:]]
B, H, Tq, Tk, dh, causal, reps = (a + [256, 8, 20, 196, 128, 0, 5][len(a):])[:7]
D = H * dh
g = torch.Generator().manual_seed(1)
q = torch.randn(B * Tq, D, generator=g).cuda().bfloat16()
kv = torch.randn(B * Tk, 2 * D, generator=g).cuda().bfloat16()
do = torch.randn(B * Tq, D, generator=g).cuda().bfloat16()
dq, dkv = torch.empty_like(q), torch.empty_like(kv)
for _ in range(reps):
    o, lse = ops.attn_fwd(q, kv[:, :D], kv[:, D:], B, H, Tq, Tk, bool(causal), 0.1, 7)
    ops.attn_bwd(q, kv[:, :D], kv[:, D:], o, do, lse, B, H, Tq, Tk, bool(causal), 0.1, 7, dq=dq, dk=dkv[:, :D], dv=dkv[:, D:])
torch.cuda.synchronize()
print("done")
