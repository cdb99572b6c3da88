"""Race screen for the 256 x 256 bf16-storage GEMMs (RACE_BK=0: gemm_bf16s_pair_kernel, RACE_BK=32 (default): the eight-phase
kernel of gemm_bf16s_p8.hip): every layout, with and without split-K,
ragged edges; the forced 256 x 256 plan is run many times and every run must equal the first one bit for bit and the
128 x 128 plan's result to fp32 accumulation tolerance.  usage: python tools/pair_race_screen.py [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagecaption_amd import ops, _capi

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
BK = int(os.environ.get("RACE_BK", "32"))
ops.COMPUTE_DTYPE = _capi.BF16S
lib = _capi.lib()
SHAPES = [(0, 1, 5120, 3072, 1024), (0, 0, 5120, 1024, 4096), (1, 0, 4096, 1024, 5120), (0, 1, 777, 1000, 520),
          (1, 0, 520, 304, 7000), (0, 0, 2048, 2048, 2048), (0, 1, 50176, 1024, 512), (1, 1, 600, 900, 1088),
          (0, 1, 5120, 30000, 2048), (0, 0, 5120, 2048, 30000), (1, 0, 30000, 2048, 5120), (0, 1, 4096, 4096, 64), (0, 0, 512, 512, 128)]
bad = 0
for ta, tb, M, N, K in SHAPES:
    g = torch.Generator().manual_seed(M + N + K)
    a = ops.cast_bf16(torch.randn((K, M) if ta else (M, K), generator=g).cuda())
    b = ops.cast_bf16(torch.randn((N, K) if tb else (K, N), generator=g).cuda())
    bias = torch.randn(N, generator=g).cuda()
    lib.capdec_gemm_tune(128, 128, 1, 0)
    ops._WS_NEED.clear()
    ref = ops.gemm(a, b, ta, tb, bias=bias)
    for split in (1, 2, 4):
        lib.capdec_gemm_tune(256, 256, split, BK)
        ops._WS_NEED.clear()
        first = ops.gemm(a, b, ta, tb, bias=bias).clone()
        err = float((first - ref).abs().max() / ref.abs().max())
        same = True
        for _ in range(reps):
            same &= bool(torch.equal(ops.gemm(a, b, ta, tb, bias=bias), first))
        ok = same and err < 1e-4
        bad += not ok
        print(f"{'nt'[ta]}{'nt'[tb]} {M}x{N}x{K} split {split}: rel err vs 128 plan {err:.2e}, {reps} repeats identical: {same}",
              flush=True)
lib.capdec_gemm_tune(0, 0, 0, 0)
print("FAILED" if bad else "ok")
sys.exit(1 if bad else 0)
