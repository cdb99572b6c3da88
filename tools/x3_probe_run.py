"""Time one f32x3 GEMM shape under the current CAPDEC_LIBRARY (probe builds)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagecaption_amd import ops, _capi
ops.COMPUTE_DTYPE = _capi.F32X3
l = _capi.lib()
for ta, tb, M, N, K, tile in [(0, 1, 5120, 10000, 1024, (128, 128, 1)), (0, 1, 5120, 10000, 1024, (64, 64, 1)),
                              (0, 1, 5120, 2048, 512, (64, 64, 1)), (1, 0, 10000, 1024, 5120, (128, 128, 2))]:
    a = torch.randn((K, M) if ta else (M, K), device="cuda")
    b = torch.randn((N, K) if tb else (K, N), device="cuda")
    out = torch.empty(M, N, device="cuda")
    l.capdec_gemm_tune(tile[0], tile[1], tile[2], 0)
    for _ in range(3): ops.gemm(a, b, ta, tb, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.gemm(a, b, ta, tb, out=out)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10 * 1e3
    print(f"{os.path.basename(_capi.LIB_PATH):28s} {'nt'[ta]}{'nt'[tb]} {M}x{N}x{K} tile {tile}: {t:8.1f} us  {2.0*M*N*K/t/1e6:6.1f} TF", flush=True)
