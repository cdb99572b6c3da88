"""Run one GEMM shape a few times (for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagecaption_amd import ops
ta, tb, M, N, K = [int(x) for x in sys.argv[1:6]]
n = int(sys.argv[6]) if len(sys.argv) > 6 else 5
a = torch.randn((K, M) if ta else (M, K), device="cuda")
b = torch.randn((N, K) if tb else (K, N), device="cuda")
out = torch.empty(M, N, device="cuda")
for _ in range(n):
    ops.gemm(a, b, ta, tb, out=out)
torch.cuda.synchronize()
