"""Kernel-only durations of the few-row decode products (run under rocprofv3 --kernel-trace, or read the event times)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagecaption_amd import ops
M = int(sys.argv[1]) if len(sys.argv) > 1 else 32
shapes = [(512, 512), (1536, 512), (2048, 512), (512, 2048), (1024, 512), (108, 1024)]
# cold weights, as in a decode step: many distinct matrices so that nothing stays in L2 between uses
ws = {s: [torch.randn(s[0], s[1], device="cuda") / 8 for _ in range(24)] for s in shapes}
for (N, K) in shapes:
    x = torch.randn(M, K, device="cuda")
    out = torch.empty(M, N, device="cuda")
    for w in ws[(N, K)][:4]:
        ops.linear_fwd(x, w, out=out)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(25)]
    e[0].record()
    for i, w in enumerate(ws[(N, K)]):
        ops.linear_fwd(x, w, out=out)
        e[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(e[i].elapsed_time(e[i + 1]) * 1e3 for i in range(24))
    print(f"M={M} N={N} K={K}: median {ts[12]:.1f} us, min {ts[0]:.1f} us (event to event, includes launch gaps)", flush=True)
