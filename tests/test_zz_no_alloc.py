"""capdec_layernorm_bwd_noalloc (the one entry point that allocates nothing) against capdec_layernorm_bwd,
through the raw C ABI: two calls in a row, the second on a smaller input, bit-exact; over-size shapes rejected.

Collected LAST (file name) so a failure here cannot hide the other GPU tests behind pytest -x."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ln_bwd_pair(M, d, seed, act=0, p=0.0, p2=0.0):
    """Run both entry points on the same seeded inputs; returns ((dx, dx_drop, dgamma, dbeta) x 2)."""
    from imagecaption_amd import _capi
    from imagecaption_amd._capi import check, lib, ptr, stream
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(M, d, generator=g).to(DEV)
    dy = torch.randn(M, d, generator=g).to(DEV)
    gamma = (1 + 0.1 * torch.randn(d, generator=g)).to(DEV)
    beta = (0.1 * torch.randn(d, generator=g)).to(DEV)
    mean = x.mean(1).contiguous()
    rstd = (x.var(1, unbiased=False) + 1e-5).rsqrt().contiguous()
    l = lib()
    outs = []
    for noalloc in (False, True):
        dx, dxd = torch.full_like(x, float("nan")), torch.full_like(x, float("nan"))
        dg, db = torch.full((d,), float("nan"), device=DEV), torch.full((d,), float("nan"), device=DEV)
        if noalloc:
            check(l.capdec_layernorm_bwd_noalloc(ptr(dy), ptr(x), ptr(gamma), ptr(beta), ptr(mean), ptr(rstd), ptr(dx),
                                                 ptr(dxd), ptr(dg), ptr(db), M, d, act, p, 11, p2, 12, stream()),
                  "capdec_layernorm_bwd_noalloc")
        else:
            ws = _capi.workspace(l.capdec_layernorm_bwd_workspace_bytes(M, d), x.device)
            check(l.capdec_layernorm_bwd(ptr(dy), ptr(x), ptr(gamma), ptr(beta), ptr(mean), ptr(rstd), ptr(dx), ptr(dxd),
                                         ptr(dg), ptr(db), M, d, act, p, 11, p2, 12, ptr(ws), ws.numel(), stream()),
                  "capdec_layernorm_bwd")
        torch.cuda.synchronize()
        outs.append((dx, dxd, dg, db))
    return outs


def test_noalloc_layernorm_bwd_matches_workspace_entry_point_twice_second_smaller():
    # call 1: the bench shape (B*T = 5120 rows, d = 512, 512 workgroups of partials = the scratch's full 2 MiB use
    # at this d); call 2: fewer rows AND a narrower row, so stale partials of call 1 lie beyond what call 2 writes
    for M, d, seed, act, p, p2 in ((5120, 512, 1, 0, 0.0, 0.1), (300, 128, 2, 1, 0.1, 0.0)):
        ref, got = _ln_bwd_pair(M, d, seed, act, p, p2)
        for name, a, b in zip(("dx", "dx_drop", "dgamma", "dbeta"), ref, got):
            assert torch.isfinite(b).all(), name
            assert torch.equal(a, b), (M, d, name)  # same kernels, same partial layout: bit-exact


def test_noalloc_allocates_nothing_and_rejects_oversize_shapes():
    from imagecaption_amd._capi import check, lib, ptr, stream
    l = lib()
    assert l.capdec_layernorm_bwd_noalloc_max_bytes() == 4 << 20
    _ln_bwd_pair(1024, 1024, 3)  # first use on this device done (symbol address looked up, code loaded)
    # 512 workgroups x 2 x 1024 floats = exactly 4 MiB: the largest shape that fits.  All tensors exist before the
    # measurement, so any change in the device's free memory across the call would be the library's own.
    M, d = 4096, 1024
    x, dy = torch.randn(M, d, device=DEV), torch.randn(M, d, device=DEV)
    gamma, beta = torch.ones(d, device=DEV), torch.zeros(d, device=DEV)
    mean, rstd = x.mean(1).contiguous(), (x.var(1, unbiased=False) + 1e-5).rsqrt().contiguous()
    dx, dg, db = torch.empty_like(x), torch.empty(d, device=DEV), torch.empty(d, device=DEV)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(3):
        check(l.capdec_layernorm_bwd_noalloc(ptr(dy), ptr(x), ptr(gamma), ptr(beta), ptr(mean), ptr(rstd), ptr(dx), None,
                                             ptr(dg), ptr(db), M, d, 0, 0.0, 0, 0.0, 0, stream()), "noalloc")
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free1 == free0 and torch.isfinite(dx).all() and torch.isfinite(dg).all()
    # 512 workgroups x 2 x 2048 floats = 8 MiB > 4 MiB: a clear error, nothing launched
    z = torch.zeros(8, device=DEV)
    p = z.data_ptr()
    rc = l.capdec_layernorm_bwd_noalloc(p, p, p, p, p, p, p, None, p, p, 4096, 2048, 0, 0.0, 0, 0.0, 0, None)
    assert rc == -4 and b"built-in scratch holds" in l.capdec_last_error()
