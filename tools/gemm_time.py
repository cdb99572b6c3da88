"""Time one GEMM shape (HIP events, 20 launches): python3 tools/gemm_time.py ta tb M N K   (env GEMM_DTYPE, GEMM_TILE=bm,bn,split)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagecaption_amd import ops, _capi
ops.COMPUTE_DTYPE = {"fp32": _capi.F32, "bf16": _capi.BF16, "f32x3": _capi.F32X3}[os.environ.get("GEMM_DTYPE", "fp32")]
if os.environ.get("GEMM_TILE"):
    bm, bn, sp, bk = ([int(x) for x in os.environ["GEMM_TILE"].split(",")] + [0])[:4]   # "bm,bn,split[,bk]"
    _capi.lib().capdec_gemm_tune(bm, bn, sp, bk)
ta, tb, M, N, K = [int(x) for x in sys.argv[1:6]]
a = torch.randn((K, M) if ta else (M, K), device="cuda")
b = torch.randn((N, K) if tb else (K, N), device="cuda")
out = torch.empty(M, N, device="cuda")
for _ in range(3): ops.gemm(a, b, ta, tb, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): ops.gemm(a, b, ta, tb, out=out)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 20 * 1e-3
print(f"{M}x{N}x{K} {t*1e6:.1f}us {2.0*M*N*K/t/1e12:.1f}TF", end="")
