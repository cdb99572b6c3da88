"""Diagnostic (library built with -DCAPDEC_P8_DIAG, CAPDEC_LIBRARY=...): where a workgroup of the eight-phase kernel spends its
time on a one-round launch -- s_memtime (100 MHz) at entry, first k-tile landed, main loop done, stores drained.
usage: python tools/p8_stamps.py M N K"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from imagecaption_amd import ops, _capi

M, N, K = (int(x) for x in sys.argv[1:4])
ops.COMPUTE_DTYPE = _capi.BF16S
lib = _capi.lib()
raw = ctypes.CDLL(os.environ["CAPDEC_LIBRARY"])
g = torch.Generator().manual_seed(0)
a = ops.cast_bf16(torch.randn(M, K, generator=g).cuda())
b = ops.cast_bf16(torch.randn(N, K, generator=g).cuda())
bias = torch.randn(N, generator=g).cuda()
out = torch.empty(M, N, device="cuda")
lib.capdec_gemm_tune(256, 256, 1, 32)
for _ in range(3):
    ops.gemm(a, b, False, True, bias=bias, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
ops.gemm(a, b, False, True, bias=bias, out=out)
e1.record()
torch.cuda.synchronize()
n = -(-M // 256) * -(-N // 256)
buf = np.zeros((n, 4), dtype=np.uint64)
rc = raw.capdec_p8_stamps(buf.ctypes.data_as(ctypes.c_void_p), n)
assert rc == 0, rc
t = buf.astype(np.int64) * 10  # ns
t0 = t[:, 0].min()
print(f"nt {M}x{N}x{K}: {n} workgroups, launch (events) {e0.elapsed_time(e1) * 1e3:.1f} us; first entry -> last end {(t[:, 3].max() - t0) / 1e3:.1f} us")
for name, d in (("entry after the first workgroup's", t[:, 0] - t0), ("entry -> k-tile 0 landed", t[:, 1] - t[:, 0]),
                ("main loop", t[:, 2] - t[:, 1]), ("epilogue (stores drained)", t[:, 3] - t[:, 2]), ("lifetime", t[:, 3] - t[:, 0])):
    print(f"  {name:36s} median {np.median(d) / 1e3:7.2f} us   p10 {np.percentile(d, 10) / 1e3:7.2f}   p90 {np.percentile(d, 90) / 1e3:7.2f}   max {d.max() / 1e3:7.2f}")
print(f"  main loop per 64-deep k-tile: {np.median(t[:, 2] - t[:, 1]) / 1e3 / (-(-K // 64)):.2f} us")
