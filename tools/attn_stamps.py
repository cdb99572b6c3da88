"""Diagnostic: where one wave of attn_fwd_kernel<64> spends its time (s_memtime stamps, CAPDEC_ATTN_STAMPS build).
Run with CAPDEC_LIBRARY=tools/_diag/libcapdec_diag.so."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from imagecaption_amd._capi import check, lib, ptr, stream

g = torch.Generator().manual_seed(0)
for name, B, H, T, S, d, causal in (("c2 cross", 256, 8, 20, 49, 512, 0), ("c2 self", 256, 8, 20, 20, 512, 1)):
    dh = d // H
    q = torch.randn(B * T, d, generator=g).cuda(); kv = torch.randn(B * S, 2 * d, generator=g).cuda()
    o = torch.empty_like(q)
    nblk = B * H * ((T + 31) // 32)
    lse = torch.zeros(B * H * T + 2 + nblk * 20, dtype=torch.float32, device="cuda")
    k, v = kv[:, :d], kv[:, d:]
    for it in range(3):
        check(lib().capdec_attn_fwd(ptr(q), q.stride(0), ptr(k), k.stride(0), ptr(v), v.stride(0), ptr(o), o.stride(0), ptr(lse),
                                    B, H, T, S, dh, causal, 1.0 / math.sqrt(dh), 0.1, 7, stream()), "attn")
        torch.cuda.synchronize()
    off = (B * H * T + 1) & ~1
    dbg = lse[off:off + nblk * 20].cpu().numpy().view(np.uint64).reshape(nblk, 10).astype(np.int64)
    np.save(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "stamps_" + name.replace(" ", "_") + ".npy"), dbg)
    names = ["start", "Q in", "K0 stored", "S done", "softmax", "V0 stored", "PV0 done", "loop done", "end"]
    print(f"== {name}: phase lengths per wave in shader cycles (s_memtime)")
    for i in range(1, 9):
        ok = (dbg[:, i] != 0) & (dbg[:, i - 1] != 0)
        d_ = (dbg[ok, i] - dbg[ok, i - 1])
        print(f"  {names[i-1]:10s} -> {names[i]:10s} median {np.median(d_):8.0f}  p10 {np.percentile(d_,10):8.0f}  p90 {np.percentile(d_,90):8.0f}")
    life = dbg[:, 8] - dbg[:, 0]
    print(f"  wave lifetime median {np.median(life):.0f} p90 {np.percentile(life,90):.0f} max {life.max()} cycles")
    hw = dbg[:, 9]
    simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
    xcc = (hw >> 32) & 15
    key = xcc * 100000 + se * 10000 + sh * 1000 + cu * 10 + simd
    cukey = key // 10
    uc, cc = np.unique(cukey, return_counts=True)
    print(f"  XCCs {np.unique(xcc).tolist()}; CUs seen {len(uc)}; waves per CU: " + ", ".join(f"{c}x{n}" for c, n in zip(*np.unique(cc, return_counts=True))))
    u, cnt = np.unique(key, return_counts=True)
    print(f"  distinct SIMD slots seen {len(u)}; waves per SIMD slot: " + ", ".join(f"{c}x{n}" for c, n in zip(*np.unique(cnt, return_counts=True))))
    for c in np.unique(cnt):
        sel = np.isin(key, u[cnt == c])
        print(f"    slots holding {c} wave(s): lifetime median {np.median(life[sel]):.0f} cycles")
