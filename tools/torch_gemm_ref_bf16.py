"""Reference point: torch.matmul on bf16 matrices (hipBLASLt) vs capdec_gemm in bf16-storage mode on the C3 step's
largest products (bf16 operands resident, no epilogue on either side; the result is fp32 for capdec, bf16 for torch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagecaption_amd import ops


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
    return e0.elapsed_time(e1) / n * 1e-3


shapes = [(0, 1, 5120, 30000, 1024), (0, 0, 5120, 1024, 30000), (1, 0, 30000, 1024, 5120), (0, 1, 5120, 4096, 1024),
          (0, 1, 5120, 1024, 4096), (0, 1, 5120, 1024, 1024), (0, 1, 5120, 3072, 1024), (0, 1, 50176, 12288, 1024),
          (0, 0, 50176, 1024, 12288), (1, 0, 1024, 1024, 5120), (1, 0, 4096, 1024, 5120)]
with ops.compute_dtype("bf16s"):
    for ta, tb, M, N, K in shapes:
        a = torch.randn((K, M) if ta else (M, K), device="cuda")
        b = torch.randn((N, K) if tb else (K, N), device="cuda")
        a16, b16 = ops.cast_bf16(a), ops.cast_bf16(b)
        at, bt = a.to(torch.bfloat16), b.to(torch.bfloat16)
        del a, b
        out = torch.empty(M, N, device="cuda")
        out16 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        A = at.t() if ta else at
        B = bt.t() if tb else bt
        tt = t(lambda: torch.matmul(A, B, out=out16))
        tc = t(lambda: ops.gemm(a16, b16, ta, tb, out=out))
        o16 = torch.empty(M, (N + 7) // 8 * 8, device="cuda", dtype=torch.bfloat16)[:, :N]
        tc16 = t(lambda: ops.gemm(a16, b16, ta, tb, out=o16))
        fl = 2.0 * M * N * K
        print(f"{'nt'[ta]}{'nt'[tb]} {M}x{N}x{K}: torch(bf16 out) {tt*1e6:8.1f} us {fl/tt/1e12:7.1f} TF | capdec(fp32 out) {tc*1e6:8.1f} us "
              f"{fl/tc/1e12:7.1f} TF | capdec(bf16 out) {tc16*1e6:8.1f} us {fl/tc16/1e12:7.1f} TF", flush=True)
        del out, out16, o16, a16, b16, at, bt
