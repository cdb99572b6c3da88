"""Tile / split-K sweep of capdec_gemm over the shapes of the C2 training step (tuning aid, GPU only)."""
import sys, os, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagecaption_amd import ops, _capi

SHAPES = [  # (ta, tb, M, N, K)
    (0, 1, 5120, 10000, 1024), (0, 0, 5120, 1024, 10000), (1, 0, 10000, 1024, 5120),
    (0, 1, 5120, 512, 512), (0, 0, 5120, 512, 512), (1, 0, 512, 512, 5120),
    (0, 1, 5120, 2048, 512), (0, 1, 5120, 512, 2048), (1, 0, 2048, 512, 5120), (1, 0, 512, 2048, 5120),
    (0, 1, 5120, 1536, 512), (1, 0, 1536, 512, 5120), (0, 1, 12544, 1024, 512), (1, 0, 1024, 512, 12544),
    (0, 0, 12544, 512, 1024), (0, 1, 12544, 6144, 512), (0, 0, 12544, 512, 6144), (1, 0, 1024, 2048, 12544),
    (0, 1, 5120, 1024, 512), (0, 0, 5120, 512, 1024), (1, 0, 1024, 512, 5120), (0, 0, 5120, 2048, 512),
    (0, 0, 5120, 512, 2048), (0, 0, 5120, 512, 1536),
]
C3_SHAPES = [  # the C3 training step (rows 5120, memory rows 50176, d 1024, ffn 4096, head 2048, vocab 30000)
    (0, 1, 5120, 30000, 2048), (0, 0, 5120, 2048, 30000), (1, 0, 30000, 2048, 5120),
    (0, 1, 5120, 3072, 1024), (0, 1, 5120, 1024, 1024), (0, 1, 5120, 4096, 1024), (0, 1, 5120, 1024, 4096),
    (0, 0, 5120, 1024, 4096), (0, 0, 5120, 4096, 1024), (1, 0, 4096, 1024, 5120), (1, 0, 1024, 4096, 5120),
    (1, 0, 1024, 1024, 5120), (1, 0, 3072, 1024, 5120),
    (0, 1, 50176, 12288, 1024), (0, 0, 50176, 1024, 12288), (1, 0, 2048, 1024, 50176),
    (0, 1, 50176, 1024, 512), (0, 1, 50176, 2048, 512), (1, 0, 1024, 2048, 50176), (0, 0, 50176, 2048, 1024),
]
if os.environ.get("SWEEP_SHAPES") == "c3":
    SHAPES = C3_SHAPES
if os.environ.get("SWEEP_ONLY"):
    SHAPES = [s for i, s in enumerate(SHAPES) if str(i) in os.environ["SWEEP_ONLY"].split(",")]

def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3

l = _capi.lib()
DT = os.environ.get("SWEEP_DTYPE", "fp32")
ops.COMPUTE_DTYPE = {"fp32": _capi.F32, "bf16": _capi.BF16, "f32x3": _capi.F32X3, "bf16s": _capi.BF16S}[DT]
CANDS = {"fp32": [(128, 128), (128, 64), (64, 128), (64, 64)], "bf16": [(128, 128), (64, 64)],
         "f32x3": [(128, 128), (128, 64), (64, 64)], "bf16s": [(256, 256), (256, 128), (128, 128), (64, 64)]}[DT]
SPLITS = [1, 2, 3, 4, 5, 6, 8, 10, 12, 16] if not os.environ.get("SWEEP_FEW") else [1, 2, 4, 8]
BKS = [16, 32] if DT == "fp32" else [0]
for ta, tb, M, N, K in SHAPES:
    a = torch.randn((K, M) if ta else (M, K), device="cuda")
    b = torch.randn((N, K) if tb else (K, N), device="cuda")
    if DT == "bf16s":  # operands as the step holds them: bf16 copies made once
        a, b = ops.cast_bf16(a), ops.cast_bf16(b)
    out = torch.empty(M, N, device="cuda")
    res = []
    l.capdec_gemm_tune(0, 0, 0, 0)
    ops._WS_NEED.clear()
    t = timeit(lambda: ops.gemm(a, b, ta, tb, out=out))
    res.append(("auto", t))
    for (bm, bn), sp, bk in itertools.product(CANDS, SPLITS, BKS if DT != "bf16s" else [0, 32]):
        if DT == "bf16s" and bk == 32 and not (bm == 128 or (bm == 256 and bn == 256)): continue  # 32 = the two-image form of the 128 x 128 LDS-DMA kernel / the eight-phase 256 x 256 kernel
        if sp > 1 and K // sp < 128: continue
        if sp > 1 and M * N * sp * 4 > (1 << 30): continue
        l.capdec_gemm_tune(bm, bn, sp, bk)
        ops._WS_NEED.clear()  # the workspace need depends on the forced plan
        try:
            t = timeit(lambda: ops.gemm(a, b, ta, tb, out=out))
        except Exception as e:
            continue
        res.append((f"{bm}x{bn}x{bk}/s{sp}", t))
    l.capdec_gemm_tune(0, 0, 0, 0)
    ops._WS_NEED.clear()
    fl = 2.0 * M * N * K
    res.sort(key=lambda r: r[1])
    auto = [r for r in res if r[0] == "auto"][0]
    print(f"{'nt'[ta]}{'nt'[tb]} {M}x{N}x{K}: auto {auto[1]*1e6:7.1f}us {fl/auto[1]/1e12:6.1f}TF | best " +
          "  ".join(f"{n} {t*1e6:.1f}us {fl/t/1e12:.1f}TF" for n, t in res[:4]), flush=True)
