"""The vocabulary forward product (bf16 storage, fused cross-entropy partials) in its result forms, timed alone:
fp32 logits / fp32 logits + target logits in the partials / bf16 logits.  usage: python tools/vocab_fwd_time.py [M V K]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagecaption_amd import ops, _capi

M, V, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (5120, 30000, 2048)
ops.COMPUTE_DTYPE = _capi.BF16S
g = torch.Generator().manual_seed(1)
h = ops.cast_bf16(torch.randn(M, K, generator=g).cuda())
w = ops.cast_bf16((torch.randn(V, K, generator=g) / 8).cuda())
b = torch.randn(V, generator=g).cuda()
y = torch.randint(0, V, (M,), generator=g).cuda()
part = torch.empty((M, ops.ce_tiles(M, V, K), 4), device="cuda")
z32 = torch.empty((M, V), device="cuda")
z16 = ops.bf16_matrix(M, V, "cuda")


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


forms = {"fp32 logits, no partials": lambda: ops.linear_fwd(h, w, b, out=z32),
         "bf16 logits, no partials": lambda: ops.linear_fwd(h, w, b, out=z16),
         "fp32 logits + partials": lambda: ops.linear_fwd(h, w, b, out=z32, ce_partials=part),
         "fp32 logits + partials + target logits": lambda: ops.linear_fwd(h, w, b, out=z32, ce_partials=part, ce_targets=y),
         "bf16 logits + partials + target logits": lambda: ops.linear_fwd(h, w, b, out=z16, ce_partials=part, ce_targets=y)}
for rep in range(2):
    for name, fn in forms.items():
        us = t(fn)
        print(f"{name:42s} {us:8.1f} us  {2.0 * M * V * K / us / 1e6:7.1f} TF/s", flush=True)
