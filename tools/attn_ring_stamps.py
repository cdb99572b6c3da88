"""Diagnostic (needs a library built with CAPDEC_EXTRA_FLAGS=-DCAPDEC_RING_STAMPS, e.g. into CAPDEC_LIBRARY=/tmp/x.so):
where a ring-kernel wave spends its cycles at the C2 shapes."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from imagecaption_amd import ops
from imagecaption_amd._capi import lib
g = torch.Generator().manual_seed(0)
for name, B, H, T, S, d, causal in (("c2 cross", 256, 8, 20, 49, 512, False), ("c2 self", 256, 8, 20, 20, 512, True)):
    q = torch.randn(B * T, d, generator=g).cuda(); kv = torch.randn(B * S, 2 * d, generator=g).cuda()
    for _ in range(3):
        ops.attn_fwd(q, kv[:, :d], kv[:, d:], B, H, T, S, causal, float(os.environ.get("ATTN_P", "0.1")), 7)
    torch.cuda.synchronize()
    n = 1024
    buf = (C.c_ulonglong * (4 * n))()
    fn = C.CDLL(os.environ.get("CAPDEC_LIBRARY")).capdec_debug_ring_stamps
    assert fn(buf, n) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(n, 4).astype(np.float64)
    for i, lab in enumerate(("lifetime", "vm_wait", "request", "lgkm_wait")):
        print(f"{name}: {lab:10s} median {np.median(a[:, i]):9.0f}  p10 {np.percentile(a[:, i], 10):9.0f}  p90 {np.percentile(a[:, i], 90):9.0f} cycles")

    if os.environ.get("STAMP_BWD"):
        do = torch.randn(B * T, d, generator=g).cuda()
        o, lse = ops.attn_fwd(q, kv[:, :d], kv[:, d:], B, H, T, S, causal, float(os.environ.get("ATTN_P", "0.1")), 7)
        for _ in range(3):
            ops.attn_bwd(q, kv[:, :d], kv[:, d:], o, do, lse, B, H, T, S, causal, float(os.environ.get("ATTN_P", "0.1")), 7)
        torch.cuda.synchronize()
        n = 1024
        assert fn(buf, n) == 0
        a = np.frombuffer(buf, dtype=np.uint64).reshape(n * 4 // 8, 8).astype(np.float64)
        labs = ("wait Q/dO", "prologue", "wait K/V", "S mfma", "dP+elementwise", "T + dQ", "dV, dK + stores", "epilogue")
        for i, lab in enumerate(labs):
            print(f"{name} bwd: {lab:16s} median {np.median(a[:, i]):9.0f}  p10 {np.percentile(a[:, i], 10):9.0f}  p90 {np.percentile(a[:, i], 90):9.0f} cycles")
        print(f"{name} bwd: total median {np.median(a.sum(1)):9.0f}")
