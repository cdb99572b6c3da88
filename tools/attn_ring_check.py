"""Ring (persistent, LDS-DMA) fp32 attention kernels against the legacy one-item-per-wave kernels: bit equality and
per-launch time (HIP events).  ATTN_P = dropout probability (default 0.1).  ATTN_BWD=0 skips the backward."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagecaption_amd import ops
from imagecaption_amd._capi import lib, check

_big = None
def t(fn, n=20):
    # the launches are enqueued behind a long kernel, so the host runs ahead and the events see GPU time only
    global _big
    if _big is None:
        _big = torch.randn(6144, 6144, device="cuda")
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    _big @ _big
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

P = float(os.environ.get("ATTN_P", "0.1"))
BWD = os.environ.get("ATTN_BWD", "1") == "1"  # noqa
g = torch.Generator().manual_seed(0)
shapes = [("tiny", 3, 8, 20, 49, 512, False), ("tiny causal", 3, 8, 20, 20, 512, True), ("odd", 2, 2, 70, 33, 128, False),
          ("long causal", 1, 4, 299, 299, 256, True), ("dh128", 2, 3, 40, 40, 384, True), ("q1", 2, 2, 1, 37, 128, True),
          ("c2 cross", 256, 8, 20, 49, 512, False), ("c2 self", 256, 8, 20, 20, 512, True),
          ("c1 self", 32, 8, 299, 299, 512, True), ("c3 cross", 256, 8, 20, 196, 1024, False),
          ("c3 self", 256, 8, 20, 20, 1024, True)]
only = os.environ.get("ATTN_ONLY")
bad = 0
for name, B, H, T, S, d, causal in shapes:
    if only and only not in name:
        continue
    q = torch.randn(B * T, d, generator=g).cuda(); kv = torch.randn(B * S, 2 * d, generator=g).cuda(); do = torch.randn(B * T, d, generator=g).cuda()
    res = {}
    for mode in (1, 0):
        check(lib().capdec_attn_tune(mode))
        o, lse = ops.attn_fwd(q, kv[:, :d], kv[:, d:], B, H, T, S, causal, P, 7)
        torch.cuda.synchronize()
        tf = t(lambda: ops.attn_fwd(q, kv[:, :d], kv[:, d:], B, H, T, S, causal, P, 7))
        tb, grads = float("nan"), None
        if BWD:
            grads = ops.attn_bwd(q, kv[:, :d], kv[:, d:], o, do, lse, B, H, T, S, causal, P, 7)
            torch.cuda.synchronize()
            tb = t(lambda: ops.attn_bwd(q, kv[:, :d], kv[:, d:], o, do, lse, B, H, T, S, causal, P, 7))
        res[mode] = (o, lse, grads, tf, tb)
    check(lib().capdec_attn_tune(0))
    same = torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    if BWD:
        same = same and all(torch.equal(a, b) for a, b in zip(res[0][2], res[1][2]))
    bad += not same
    fb = 4.0 * B * d * (2 * T + 2 * S); bb = 4.0 * B * d * (4 * T + 4 * S)
    print(f"p {P} {name:12s} bit-equal {same} | fwd legacy {res[1][3]:7.1f} ring {res[0][3]:7.1f} us ({fb/res[0][3]/1e6:5.2f} TB/s)"
          f" | bwd legacy {res[1][4]:7.1f} ring {res[0][4]:7.1f} us ({bb/res[0][4]/1e6:5.2f} TB/s)", flush=True)
    if not same:
        d0 = (res[0][0] - res[1][0]).abs().max().item()
        print(f"    max |o diff| {d0:.3e}  nan {torch.isnan(res[0][0]).any().item()}", flush=True)
sys.exit(1 if bad else 0)
