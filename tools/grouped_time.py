"""The weight gradients of one C2 decoder layer: one grouped launch (capdec_gemm_grouped, by block tile) against the seven
single launches under the library's plans.  HIP events behind a long kernel (GPU time only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagecaption_amd import ops

big = torch.randn(6144, 6144, device="cuda")
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    big @ big
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

d, R, RS = 512, 5120, 12544
shapes = [(1024, d, RS), (3 * d, d, R), (d, d, R), (d, d, R), (d, d, R), (4 * d, d, R), (d, 4 * d, R)]  # (N_out, K_in, rows)
g = torch.Generator().manual_seed(0)
prods = []
for n_out, k_in, rows in shapes:
    dy = torch.randn(rows, n_out, generator=g).cuda(); x = torch.randn(rows, k_in, generator=g).cuda()
    prods.append((dy, x, True, False, torch.empty(n_out, k_in, device="cuda"), torch.empty(n_out, device="cuda")))
flops = sum(2.0 * a.shape[1] * b.shape[1] * a.shape[0] for a, b, *_ in prods)
ts = t(lambda: [ops.gemm(a, b, True, False, out=o, colsum=c) for a, b, _, _, o, c in prods])
print(f"seven single launches (library plans): {ts:7.1f} us  {flops/ts/1e6:6.1f} TFLOP/s")
for tile in ((64, 64), (128, 64), (64, 128), (128, 128)):
    tg = t(lambda: ops.gemm_grouped(prods, tile))
    print(f"grouped {tile}: {tg:7.1f} us  {flops/tg/1e6:6.1f} TFLOP/s")
