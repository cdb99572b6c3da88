"""All-bf16 attention (capdec_attn_fwd_bf16 / capdec_attn_bwd_bf16, the attention of the CAPDEC_BF16S storage path).

Oracles:
* no dropout: float64 torch attention ON THE bf16-ROUNDED inputs.  What is left is the arithmetic the kernel is
  documented to use: fp32 accumulation, probabilities / score gradients rounded to bf16 (2^-9 relative each, averaged
  over the keys) and bf16 results (2^-9 relative) -- tolerance 1.2e-2 of the largest reference element and 6e-3
  relative RMS, stated below;
* with dropout: the exact-fp32 kernels (themselves checked against float64 torch in test_ops_gpu.py) run on the
  widened inputs with the SAME seed -- the masks must be the same ones (same element index), so the same tolerance
  applies; a wrong mask would be an O(1) error.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"
TOL_MAX, TOL_RMS = 1.2e-2, 6e-3

SHAPES = [
    # B, H, Tq, Tk, dh, causal
    (2, 8, 20, 196, 128, False),   # C3 cross-attention: 7 key tiles, the last with 4 keys
    (3, 8, 20, 49, 64, False),     # C2 cross-attention
    (2, 4, 20, 20, 128, True),     # C3 self-attention
    (2, 2, 70, 70, 64, True),      # three query tiles, causal
    (1, 2, 33, 100, 64, False),    # a second query tile with one row
    (1, 1, 40, 75, 128, True),     # causal with more keys than queries (offset 35)
    (2, 2, 5, 1, 64, False),       # one key
    (1, 2, 299, 299, 64, True),    # C1 caption length: ten tiles, ring of four images wraps
    (2, 2, 20, 50, 64, True),      # one query tile (the one-walk backward), causal with an offset of 30
    (1, 2, 7, 100, 128, True),     # one query tile, more key tiles than the ring holds
]


def _ops():
    from imagecaption_amd import ops
    return ops


def _err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    # floors: with one key dq and dk are exactly zero in the reference and rounding noise (1e-6) here
    scale = max(float(b.abs().max()), 1e-2)
    return float((a - b).abs().max()) / scale, float((a - b).norm() / (b.norm() + 1e-2 * math.sqrt(b.numel())))


def _inputs(B, H, Tq, Tk, dh, seed):
    """q, k, v, do as bf16 views of packed, over-wide blocks (row strides the engine's packed projections have)."""
    ops = _ops()
    g = torch.Generator().manual_seed(seed)
    D = H * dh
    qblk = torch.randn(B * Tq, ops._bf16_pitch(3 * D), generator=g).to(DEV).to(torch.bfloat16)
    kvblk = torch.randn(B * Tk, ops._bf16_pitch(2 * D), generator=g).to(DEV).to(torch.bfloat16)
    do = torch.randn(B * Tq, D, generator=g).to(DEV).to(torch.bfloat16)
    return qblk[:, D:2 * D], kvblk[:, :D], kvblk[:, D:2 * D], do


def _ref64(q, k, v, do, B, H, Tq, Tk, dh, causal):
    q4 = q.double().view(B, Tq, H, dh).transpose(1, 2).detach().requires_grad_(True)
    k4 = k.double().view(B, Tk, H, dh).transpose(1, 2).detach().requires_grad_(True)
    v4 = v.double().view(B, Tk, H, dh).transpose(1, 2).detach().requires_grad_(True)
    s = q4 @ k4.transpose(-1, -2) / math.sqrt(dh)
    if causal:
        tq = torch.arange(Tq, device=s.device)[:, None]
        tk = torch.arange(Tk, device=s.device)[None, :]
        s = s.masked_fill(tk > tq + (Tk - Tq), float("-inf"))
    lse = torch.logsumexp(s, dim=-1)
    o4 = torch.softmax(s, dim=-1) @ v4
    o = o4.transpose(1, 2).reshape(B * Tq, H * dh)
    o.backward(do.double())
    back = lambda t, T: t.grad.transpose(1, 2).reshape(B * T, H * dh)
    return o.detach(), lse.detach(), back(q4, Tq), back(k4, Tk), back(v4, Tk)


@pytest.mark.parametrize("B,H,Tq,Tk,dh,causal", SHAPES)
def test_bf16_attention_matches_float64_on_rounded_inputs(B, H, Tq, Tk, dh, causal):
    ops = _ops()
    q, k, v, do = _inputs(B, H, Tq, Tk, dh, 11 * Tq + Tk)
    o_ref, lse_ref, dq_ref, dk_ref, dv_ref = _ref64(q, k, v, do, B, H, Tq, Tk, dh, causal)
    o, lse = ops.attn_fwd(q, k, v, B, H, Tq, Tk, causal)
    assert o.dtype == torch.bfloat16 and tuple(o.shape) == (B * Tq, H * dh)
    emax, erms = _err(o, o_ref)
    assert emax < TOL_MAX and erms < TOL_RMS, (emax, erms)
    assert float((lse.double() - lse_ref).abs().max()) < 2e-5 * (1 + float(lse_ref.abs().max()))
    # backward from the reference's own (rounded) output, so the three gradients are compared on equal terms
    o16 = o_ref.to(torch.bfloat16)
    dq, dk, dv = ops.attn_bwd(q, k, v, o16, do, lse, B, H, Tq, Tk, causal)
    for name, got, ref in (("dq", dq, dq_ref), ("dk", dk, dk_ref), ("dv", dv, dv_ref)):
        assert got.dtype == torch.bfloat16
        emax, erms = _err(got, ref)
        # delta = rowsum(dO * O) is taken from the bf16 O: one more 2^-9 term in dq / dk
        assert emax < 2 * TOL_MAX and erms < 2 * TOL_RMS, (name, emax, erms)


@pytest.mark.parametrize("B,H,Tq,Tk,dh,causal", SHAPES[:6] + SHAPES[8:])
def test_bf16_attention_uses_the_fp32_kernels_dropout_masks(B, H, Tq, Tk, dh, causal):
    ops = _ops()
    q, k, v, do = _inputs(B, H, Tq, Tk, dh, 5 * Tq + Tk)
    p, seed = 0.25, 0x1234567890ABCDEF
    o_ref, lse_ref = ops.attn_fwd(q.float(), k.float(), v.float(), B, H, Tq, Tk, causal, p, seed)
    o, lse = ops.attn_fwd(q, k, v, B, H, Tq, Tk, causal, p, seed)
    emax, erms = _err(o, o_ref)
    assert emax < TOL_MAX and erms < TOL_RMS, (emax, erms)
    assert float((lse - lse_ref).abs().max()) < 2e-5 * (1 + float(lse_ref.abs().max()))
    o16 = o_ref.to(torch.bfloat16)
    ref = ops.attn_bwd(q.float(), k.float(), v.float(), o16.float(), do.float(), lse_ref, B, H, Tq, Tk, causal, p, seed)
    got = ops.attn_bwd(q, k, v, o16, do, lse_ref, B, H, Tq, Tk, causal, p, seed)
    for name, a, b in zip(("dq", "dk", "dv"), got, ref):
        emax, erms = _err(a, b)
        assert emax < 2 * TOL_MAX and erms < 2 * TOL_RMS, (name, emax, erms)


def test_bf16_attention_is_bitwise_reproducible_and_writes_into_given_views():
    ops = _ops()
    B, H, Tq, Tk, dh = 2, 8, 20, 196, 128
    D = H * dh
    q, k, v, do = _inputs(B, H, Tq, Tk, dh, 3)
    o1, lse1 = ops.attn_fwd(q, k, v, B, H, Tq, Tk, False, 0.1, 9)
    blk = torch.full((B * Tq, ops._bf16_pitch(2 * D)), 7.0, dtype=torch.bfloat16, device=DEV)
    o2, lse2 = ops.attn_fwd(q, k, v, B, H, Tq, Tk, False, 0.1, 9, out=blk[:, D:2 * D])
    assert torch.equal(o1, o2) and torch.equal(lse1, lse2)
    assert torch.all(blk[:, :D] == 7.0)  # nothing outside the view was written
    g1 = ops.attn_bwd(q, k, v, o1, do, lse1, B, H, Tq, Tk, False, 0.1, 9)
    dkv = torch.full((B * Tk, ops._bf16_pitch(2 * D)), 7.0, dtype=torch.bfloat16, device=DEV)
    g2 = ops.attn_bwd(q, k, v, o1, do, lse1, B, H, Tq, Tk, False, 0.1, 9, dk=dkv[:, :D], dv=dkv[:, D:2 * D])
    for a, b in zip(g1, g2):
        assert torch.equal(a, b)
    whole = torch.as_strided(dkv, (B * Tk, dkv.stride(0)), (dkv.stride(0), 1))
    assert torch.all(whole[:, 2 * D:] == 7.0)


def test_bf16_attention_contract_errors():
    ops = _ops()
    from imagecaption_amd._capi import CapdecError
    q, k, v, do = _inputs(1, 2, 4, 6, 32, 1)  # head dim 32 is not covered
    with pytest.raises(CapdecError):
        ops.attn_fwd(q, k, v, 1, 2, 4, 6, False)
    q, k, v, do = _inputs(1, 2, 4, 6, 64, 1)
    with pytest.raises(CapdecError):  # mixed types: bf16 q with fp32 k
        ops.attn_fwd(q, k.float(), v.float(), 1, 2, 4, 6, False)
    o, lse = ops.attn_fwd(q, k, v, 1, 2, 4, 6, False)
    with pytest.raises(CapdecError):  # fp32 dO with bf16 q
        ops.attn_bwd(q, k, v, o, do.float(), lse, 1, 2, 4, 6, False)
