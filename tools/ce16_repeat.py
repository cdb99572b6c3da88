"""Repeat screen of the vocabulary forward with bf16 logits, cross-entropy partials and target logits (eight-phase kernel and
the 128 x 128 forms): every launch must reproduce the first one's logits and partials bit for bit.  usage: python tools/ce16_repeat.py [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagecaption_amd import ops, _capi

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
ops.COMPUTE_DTYPE = _capi.BF16S
lib = _capi.lib()
bad = 0
for M, V, K in ((5120, 30000, 2048), (5120, 10000, 1024), (777, 1000, 520), (300, 2052, 192)):
    g = torch.Generator().manual_seed(M + V)
    h = ops.cast_bf16(torch.randn(M, K, generator=g).cuda())
    w = ops.cast_bf16((torch.randn(V, K, generator=g) / 8).cuda())
    b = torch.randn(V, generator=g).cuda()
    y = torch.randint(0, V, (M,), generator=g).cuda()
    for plan in ((256, 256, 1, 32), (128, 128, 1, 32), (128, 128, 1, 0), (256, 256, 1, 0)):
        lib.capdec_gemm_tune(*plan)
        ops._WS_NEED.clear()
        part = torch.empty((M, ops.ce_tiles(M, V, K), 4), device="cuda")
        z = ops.bf16_matrix(M, V, "cuda")
        ops.linear_fwd(h, w, b, out=z, ce_partials=part, ce_targets=y)
        z0, p0 = z.clone(), part.clone()
        same = True
        for _ in range(reps):
            part.fill_(float("nan"))
            ops.linear_fwd(h, w, b, out=z, ce_partials=part, ce_targets=y)
            same &= bool(torch.equal(z, z0)) and bool(torch.equal(part, p0))
        bad += not same
        print(f"{M}x{V}x{K} plan {plan}: {reps} repeats identical: {same}", flush=True)
lib.capdec_gemm_tune(0, 0, 0, 0)
print("FAILED" if bad else "ok")
sys.exit(1 if bad else 0)
