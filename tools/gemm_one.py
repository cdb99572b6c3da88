"""Run one GEMM shape a few times (for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagecaption_amd import ops, _capi
ops.COMPUTE_DTYPE = {"fp32": _capi.F32, "bf16": _capi.BF16, "f32x3": _capi.F32X3,
                     "bf16s": _capi.BF16S}[os.environ.get("GEMM_DTYPE", "fp32")]
if os.environ.get("GEMM_TILE"):  # "bm,bn,split"
    bm, bn, sp, bk = ([int(x) for x in os.environ["GEMM_TILE"].split(",")] + [0])[:4]   # "bm,bn,split[,bk]"
    _capi.lib().capdec_gemm_tune(bm, bn, sp, bk)
ta, tb, M, N, K = [int(x) for x in sys.argv[1:6]]
n = int(sys.argv[6]) if len(sys.argv) > 6 else 5
a = torch.randn((K, M) if ta else (M, K), device="cuda")
b = torch.randn((N, K) if tb else (K, N), device="cuda")
if ops.COMPUTE_DTYPE == _capi.BF16S:  # operands as a training step holds them: bf16 copies made once
    a, b = ops.cast_bf16(a), ops.cast_bf16(b)
out = torch.empty(M, N, device="cuda")
kw = {}
if os.environ.get("GEMM_CE"):  # the vocabulary projection as the trainer launches it: bias + cross-entropy partials
    kw = dict(bias=torch.randn(N, device="cuda"), ce_partials=torch.empty(M, ops.ce_tiles(M, N, K), 4, device="cuda"))
if os.environ.get("GEMM_CE") == "16":  # ... with the logits block kept in bf16 (the bf16 storage mode's default)
    out = ops.bf16_matrix(M, N, "cuda")
    kw["ce_targets"] = torch.randint(0, N, (M,), device="cuda")
ops.gemm(a, b, ta, tb, out=out, **kw)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    ops.gemm(a, b, ta, tb, out=out, **kw)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / n * 1e3
print(f"{'nt'[ta]}{'nt'[tb]} {M}x{N}x{K} {os.environ.get('GEMM_DTYPE', 'fp32')} tile {os.environ.get('GEMM_TILE', 'auto')} "
      f"{'+CE' if kw else ''}: {us:.1f} us  {2.0 * M * N * K / us / 1e6:.1f} TF", flush=True)
