"""Reference point: torch.matmul (rocBLAS / hipBLASLt, fp32) on the step's GEMM shapes vs capdec_gemm (fp32 exact)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagecaption_amd import ops

def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3

torch.backends.cuda.matmul.allow_tf32 = False
for ta, tb, M, N, K in [(0, 1, 5120, 10000, 1024), (0, 0, 5120, 1024, 10000), (1, 0, 10000, 1024, 5120), (0, 1, 5120, 512, 512),
                        (1, 0, 512, 512, 5120), (0, 1, 5120, 2048, 512), (0, 1, 12544, 1024, 512), (0, 1, 12544, 1024, 2048)]:
    a = torch.randn((K, M) if ta else (M, K), device="cuda")
    b = torch.randn((N, K) if tb else (K, N), device="cuda")
    out = torch.empty(M, N, device="cuda")
    A = a.t() if ta else a
    B = b.t() if tb else b
    tt = t(lambda: torch.matmul(A, B, out=out))
    tc = t(lambda: ops.gemm(a, b, ta, tb, out=out))
    fl = 2.0 * M * N * K
    print(f"{'tn'[ta]}{'nt'[tb]} {M}x{N}x{K}: torch {tt*1e6:8.1f} us {fl/tt/1e12:6.1f} TF | capdec {tc*1e6:8.1f} us {fl/tc/1e12:6.1f} TF", flush=True)
