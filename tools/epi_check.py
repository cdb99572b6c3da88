"""Vector epilogue vs element epilogue, and run-to-run determinism, over layouts / epilogue options / compute types (GPU only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagecaption_amd import ops, _capi
from imagecaption_amd._capi import ACT_GELU, ACT_GELU_BWD
l = _capi.lib()
DEV = "cuda"
def variants(M, N, K, ta, tb, dt):
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn((K, M) if ta else (M, K), generator=g).to(DEV)
    b_ = (torch.randn((N, K) if tb else (K, N), generator=g) / 8).to(DEV)
    bias, r = torch.randn(N, generator=g).to(DEV), torch.randn(M, N, generator=g).to(DEV)
    with ops.compute_dtype(dt):
        if dt == "bf16s":
            a, b_ = ops.cast_bf16(a), ops.cast_bf16(b_)
        pre = torch.empty(M, N, device=DEV)
        o16 = torch.empty(M, (N + 7) // 8 * 8, device=DEV, dtype=torch.bfloat16)[:, :N]
        res = [ops.gemm(a, b_, ta, tb),
               ops.gemm(a, b_, ta, tb, bias=bias),
               ops.gemm(a, b_, ta, tb, bias=bias, drop_p=0.1, seed=77, addend=r),
               ops.gemm(a, b_, ta, tb, bias=bias, act=ACT_GELU, pre=pre), pre.clone(),
               ops.gemm(a, b_, ta, tb, aux=r, act=ACT_GELU_BWD, drop_p=0.1, seed=3)]
        inpl = r.clone()
        ops.gemm(a, b_, ta, tb, bias=bias, drop_p=0.1, seed=9, addend=inpl, out=inpl)   # in-place residual
        res.append(inpl)
        if dt == "bf16s":
            res.append(ops.gemm(a, b_, ta, tb, bias=bias, out=o16).float())
            res.append(ops.gemm(a, b_, ta, tb, bias=bias, act=ACT_GELU, pre=pre, out=o16).float())
            res.append(ops.gemm(a, b_, ta, tb, aux=r, act=ACT_GELU_BWD, out=o16).float())
        if ta:
            cs = torch.empty(M, device=DEV)
            res.append(ops.gemm(a, b_, ta, tb, colsum=cs)); res.append(cs)
    return res
shapes = [(5120, 512, 512), (5120, 2048, 512), (5120, 512, 2048), (512, 512, 5120), (12544, 1024, 512), (300, 260, 200), (5120, 1536, 512)]
bad = 0
for dt in sys.argv[1:] or ["bf16s", "fp32"]:
    for (M, N, K) in shapes:
        for ta, tb in ((0, 1), (0, 0), (1, 0)):
            os.environ.pop("CAPDEC_NO_VEC_EPI", None)
            v1 = variants(M, N, K, ta, tb, dt)
            v2 = variants(M, N, K, ta, tb, dt)
            os.environ["CAPDEC_NO_VEC_EPI"] = "1"
            e1 = variants(M, N, K, ta, tb, dt)
            os.environ.pop("CAPDEC_NO_VEC_EPI", None)
            for i, (x, y, z) in enumerate(zip(v1, v2, e1)):
                rep, same = bool(torch.equal(x, y)), bool(torch.equal(x, z))
                if not (rep and same):
                    bad += 1
                    print(dt, (M, N, K), "ta/tb", ta, tb, "variant", i, "repeatable", rep, "equals element form", same,
                          "maxdiff", float((x - z).abs().max()), "n", int(((x - z).abs() > 0).sum()), flush=True)
print("bad:", bad)
