"""Time forced bf16-storage GEMM plans on a few nt shapes (GPU only): python tools/p8_time.py [M N K ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagecaption_amd import ops, _capi

def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3

l = _capi.lib()
ops.COMPUTE_DTYPE = _capi.BF16S
args = [int(v) for v in sys.argv[1:]]
shapes = [tuple(args[i:i + 3]) for i in range(0, len(args), 3)] or [(4096, 4096, 4096), (8192, 8192, 8192), (5120, 4096, 1024), (5120, 30000, 2048)]
plans = [(128, 128, 1, 0), (128, 128, 1, 32), (256, 256, 1, 0), (256, 256, 1, 32)]
for M, N, K in shapes:
    a = ops.cast_bf16(torch.randn(M, K, device="cuda"))
    b = ops.cast_bf16(torch.randn(N, K, device="cuda"))
    out = torch.empty(M, N, device="cuda")
    out16 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    row = []
    for pl in plans:
        l.capdec_gemm_tune(*pl)
        ops._WS_NEED.clear()
        t = timeit(lambda: ops.gemm(a, b, 0, 1, out=out))
        row.append(f"{pl[0]}x{pl[1]}x{pl[3]} {t*1e6:8.1f}us {2.0*M*N*K/t/1e12:7.1f}TF")
    l.capdec_gemm_tune(0, 0, 0, 0)
    t = timeit(lambda: torch.matmul(a[:, :K], b[:, :K].T, out=out16))
    row.append(f"torch(bf16 out) {t*1e6:8.1f}us {2.0*M*N*K/t/1e12:7.1f}TF")
    print(f"nt {M}x{N}x{K}: " + " | ".join(row), flush=True)
