"""capdec_beam_select at a 30 000-word vocabulary: LDS-staged (rows read once) vs re-reading kernel (ldl = V + 4 keeps float4 but... no: V + 1 breaks staging)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagecaption_amd._capi import lib, check, ptr, stream
N, W, V, Tmax, t = 2048, 5, 30000, 20, 7
g = torch.Generator().manual_seed(0)
base = (torch.randn(N * W, V + 1, generator=g) * 3).cuda()
score = (torch.randn(N, W, generator=g) * 2).cuda()
fin = torch.zeros(N * W, dtype=torch.int32, device="cuda")
anc = torch.randint(0, N * W, (N * W, Tmax), generator=g).to(torch.int32).cuda()
seq = torch.randint(0, V, (N * W, Tmax), generator=g).to(torch.int32).cuda()
so, fo, ao, qo = torch.empty_like(score), torch.empty_like(fin), torch.zeros_like(anc), torch.zeros_like(seq)
tok = torch.empty(N * W, dtype=torch.int64, device="cuda")
for name, lg in (("staged (one read)", base[:, :V].contiguous()), ("unstaged (three reads)", base[:, :V])):
    def run():
        check(lib().capdec_beam_select(ptr(lg), lg.stride(0), ptr(score), ptr(so), ptr(fin), ptr(fo), ptr(anc), ptr(ao),
                                       ptr(seq), ptr(qo), ptr(tok), None, N, W, V, Tmax, t, stream()), "beam_select")
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"beam_select N={N} W={W} V={V} {name}: {us:.1f} us, {N * W * V * 4 / us / 1e3:.0f} GB/s of logits", flush=True)
