"""LayerNorm forward / backward kernel time and effective bandwidth at the step's shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagecaption_amd import ops

_big = torch.randn(6144, 6144, device="cuda")
def t(fn, n=50):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    _big @ _big  # the host runs ahead behind a long kernel: the events see GPU time only
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3

for M, d in ((5120, 512), (12544, 512), (9568, 512), (5120, 1024)):
    x = torch.randn(M, d, device="cuda"); dy = torch.randn(M, d, device="cuda")
    g = torch.randn(d, device="cuda"); b = torch.randn(d, device="cuda")
    y, mean, rstd = ops.layernorm_fwd(x, g, b)
    dg = torch.empty(d, device="cuda"); db = torch.empty(d, device="cuda")
    tf = t(lambda: ops.layernorm_fwd(x, g, b))
    tb = t(lambda: ops.layernorm_bwd(dy, x, g, b, mean, rstd, dgamma=dg, dbeta=db))
    tb2 = t(lambda: ops.layernorm_bwd(dy, x, g, b, mean, rstd, drop2_p=0.1, seed2=5, want_dx_drop=True, dgamma=dg, dbeta=db))
    by = M * d * 4
    print(f"{M}x{d}: fwd {tf*1e6:6.1f} us {2*by/tf/1e12:5.2f} TB/s | bwd {tb*1e6:6.1f} us {3*by/tb/1e12:5.2f} TB/s | bwd+masked copy {tb2*1e6:6.1f} us {4*by/tb2/1e12:5.2f} TB/s", flush=True)
