"""Which epilogue variants of the eight-phase kernel differ from the 128 x 128 kernel's element-form epilogue (GPU only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagecaption_amd import ops, _capi
from imagecaption_amd._capi import ACT_GELU, ACT_GELU_BWD
l = _capi.lib()
DEV = "cuda"
M, N, K = 700, 520, 64
g = torch.Generator().manual_seed(1)
x, w = torch.randn(M, K, generator=g).to(DEV), (torch.randn(N, K, generator=g) / 8).to(DEV)
b, r = torch.randn(N, generator=g).to(DEV), torch.randn(M, N, generator=g).to(DEV)
with ops.compute_dtype("bf16s"):
    x16, w16 = ops.cast_bf16(x), ops.cast_bf16(w)
    def variants():
        pre = torch.empty(M, N, device=DEV)
        o16 = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        return [ops.linear_fwd(x16, w16, b, act=ACT_GELU, addend=r), ops.gemm(x16, w16, 0, 1),
                ops.linear_fwd(x16, w16, b, act=ACT_GELU), ops.linear_fwd(x16, w16, b, addend=r),
                ops.linear_fwd(x16, w16, b, drop_p=0.1, seed=77, addend=r), ops.linear_fwd(x16, w16, b, drop_p=0.3, seed=5),
                ops.linear_fwd(x16, w16, b, act=ACT_GELU, pre=pre, out=o16).float(), pre,
                ops.gemm(x16, w16, 0, 1, bias=b, aux=r, act=ACT_GELU_BWD, addend=pre),
                ops.gemm(x16, w16, 0, 1, aux=r, act=ACT_GELU_BWD, drop_p=0.1, seed=3)]
    l.capdec_gemm_tune(128, 128, 1, 0)
    want = variants()
    l.capdec_gemm_tune(256, 256, 1, 32)
    got = variants()
    l.capdec_gemm_tune(0, 0, 0, 0)
for i, (a, c) in enumerate(zip(got, want)):
    d = (a - c).abs()
    print(i, bool(torch.equal(a, c)), float(d.max()), int((d > 0).sum()), "of", d.numel())
