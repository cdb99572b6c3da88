"""Launch plans of the caption-decoder hot path: which kernel runs on what, forward and backward.

This is the host-side "graph" of the build.  It is a fixed sequence of C-ABI launches per block
(no tracing compiler): the same functions serve the autograd wrappers of ``functional.py`` (drop-in
nn.Module path) and the allocation-light trainer / decoder of ``trainer.py`` / ``decode.py``.

The arithmetic mirrors, block for block, what the reference executes through torch.nn:
* decoder layer, post-norm:       torch/nn/modules/transformer.py:1143-1199 (as built at ref models.py:53-65)
* packed in-projection / SDPA:    torch/nn/functional.py:5819-5852, 6386-6390, 6613-6642
* feature projection, head:       ref models.py:37-44, 68-73
Parameter names are the reference's state_dict keys.
"""
import torch

from . import ops
from ._capi import ACT_GELU

LAYER_KEYS = ("self_attn.in_proj_weight", "self_attn.in_proj_bias", "self_attn.out_proj.weight",
              "self_attn.out_proj.bias", "multihead_attn.in_proj_weight", "multihead_attn.in_proj_bias",
              "multihead_attn.out_proj.weight", "multihead_attn.out_proj.bias", "linear1.weight", "linear1.bias",
              "linear2.weight", "linear2.bias", "norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias",
              "norm3.weight", "norm3.bias")
PROJ_KEYS = ("0.weight", "0.bias", "1.weight", "1.bias", "4.weight", "4.bias", "5.weight", "5.bias")
HEAD_KEYS = ("0.weight", "0.bias", "3.weight", "3.bias")

_M64 = (1 << 64) - 1


def _splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & _M64
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


class DropoutPlan:
    """Per-call dropout seeds: seed(site) is a 64-bit hash of (base seed, call counter, site id).

    ``p == 0`` (eval mode) disables every site.  Backward re-derives the
    masks from the same (p, seed) pairs, so nothing but this small object is kept.
    """

    def __init__(self, p=0.0, base_seed=0, call=0):
        self.p = float(p)
        self.key = _splitmix64((int(base_seed) & _M64) ^ _splitmix64(int(call)))

    def seed(self, site):
        return _splitmix64(self.key ^ ((int(site) * 0xD6E8FEB86659FD93) & _M64))


# site ids: layer l uses 16*l + k; blocks outside the stack use the 1000 range
S_SA_P, S_SA_OUT, S_CA_P, S_CA_OUT, S_FFN_IN, S_FFN_OUT = range(6)
S_EMBED, S_PROJ, S_HEAD = 1000, 1001, 1002


# ---------------------------------------------------------------------------------------------
# decoder layer
# ---------------------------------------------------------------------------------------------
def _single_key_fast(S, d, H):
    """The streaming single-key cross-attention branch (capdec_cross1_*) covers d % 64 == 0 and power-of-two heads."""
    return S == 1 and d % 64 == 0 and H in (1, 2, 4, 8, 16)


def _attn16(H, d):
    return ops.COMPUTE_DTYPE == ops.BF16S and ops.attn16_ok(H, d)


def _hidden16(rows, cols, device):
    """bf16 storage with a 64-multiple width: a bf16-only result buffer (its padding is then never read); else None."""
    return ops.bf16_matrix(rows, cols, device) if ops.COMPUTE_DTYPE == ops.BF16S and cols % 64 == 0 else None


def _like16(t, rows, cols):
    """bf16 result buffer when ``t`` (what forward saved) is bf16, else None (the product allocates fp32)."""
    return ops.bf16_matrix(rows, cols, t.device) if t.dtype == torch.bfloat16 else None


def layer_fwd(P, x, mem, B, T, S, H, causal, dp, l, save, kv=None):
    """x [B*T,d], mem [B*S,d] -> [B*T,d].  ``save`` (dict or None) receives what backward needs.
    ``kv``: this layer's cross-attention K/V projection of ``mem`` when the caller has already computed it
    (stack_fwd batches all layers' projections into one GEMM); a [B*S, 2d] row-strided view."""
    d = x.shape[1]
    p, sid = dp.p, 16 * l
    # bf16 storage: the projections around attention write bf16 and the all-bf16 attention kernels read / write it
    # (q, k, v, the context and, in backward, dO, dq, dk, dv never exist as fp32; no cast passes around attention)
    a16 = _attn16(H, d)
    qkv = ops.linear_fwd(x, P["self_attn.in_proj_weight"], P["self_attn.in_proj_bias"],
                         out=ops.bf16_matrix(x.shape[0], 3 * d, x.device) if a16 else None)
    a, lse1 = ops.attn_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], B, H, T, T, causal, p, dp.seed(sid + S_SA_P),
                           save_lse=save is not None)
    s1 = ops.linear_fwd(a, P["self_attn.out_proj.weight"], P["self_attn.out_proj.bias"], drop_p=p,
                        seed=dp.seed(sid + S_SA_OUT), addend=x)
    x1, mean1, rstd1 = ops.layernorm_fwd(s1, P["norm1.weight"], P["norm1.bias"], save_stats=save is not None)
    wc, bc = P["multihead_attn.in_proj_weight"], P["multihead_attn.in_proj_bias"]
    if kv is None:
        kv = ops.linear_fwd(mem, wc[d:], bc[d:],
                            out=ops.bf16_matrix(mem.shape[0], 2 * d, x.device) if a16 and S > 1 else None)
    q = c = lse2 = vblk = None
    if _single_key_fast(S, d, H):
        # ONE memory position (ref models.py:104): softmax over one key is 1 whatever the query (the query
        # projection cannot reach the output, its gradient is exactly zero) and head h's context is v[b, h-slice]
        # for every token, kept or dropped by the probability dropout.  The out-projection therefore runs on
        # B*H rows (v masked to one head per row) and a streaming kernel assembles the [B*T, d] branch output.
        vblk = ops.cross1_spread(kv[:, d:], B, H)
        U = ops.linear_fwd(vblk, P["multihead_attn.out_proj.weight"])
        s2 = ops.cross1_fwd(U, P["multihead_attn.out_proj.bias"], x1, B, T, H, p, dp.seed(sid + S_CA_P), p,
                            dp.seed(sid + S_CA_OUT))
    else:
        # S == 1 with an odd geometry: generic kernels
        q = x1 if S == 1 else ops.linear_fwd(x1, wc[:d], bc[:d],
                                             out=ops.bf16_matrix(x.shape[0], d, x.device) if a16 else None)
        c, lse2 = ops.attn_fwd(q, kv[:, :d], kv[:, d:], B, H, T, S, False, p, dp.seed(sid + S_CA_P),
                               save_lse=save is not None and S != 1)
        s2 = ops.linear_fwd(c, P["multihead_attn.out_proj.weight"], P["multihead_attn.out_proj.bias"], drop_p=p,
                            seed=dp.seed(sid + S_CA_OUT), addend=x1)
    x2, mean2, rstd2 = ops.layernorm_fwd(s2, P["norm2.weight"], P["norm2.bias"], save_stats=save is not None)
    u = torch.empty((x.shape[0], P["linear1.weight"].shape[0]), dtype=torch.float32, device=x.device) \
        if save is not None else None
    # bf16 storage: the hidden activation is read by matrix products only -- it exists as bf16 alone
    h = ops.linear_fwd(x2, P["linear1.weight"], P["linear1.bias"], act=ACT_GELU, pre=u, drop_p=p,
                       seed=dp.seed(sid + S_FFN_IN), out=_hidden16(x.shape[0], P["linear1.weight"].shape[0], x.device))
    s3 = ops.linear_fwd(h, P["linear2.weight"], P["linear2.bias"], drop_p=p, seed=dp.seed(sid + S_FFN_OUT), addend=x2)
    x3, mean3, rstd3 = ops.layernorm_fwd(s3, P["norm3.weight"], P["norm3.bias"], save_stats=save is not None)
    if save is not None:
        save.update(x=x, qkv=qkv, a=a, lse1=lse1, s1=s1, mean1=mean1, rstd1=rstd1, x1=x1, q=q, kv=kv, c=c, lse2=lse2, vblk=vblk,
                    s2=s2, mean2=mean2, rstd2=rstd2, x2=x2, u=u, h=h, s3=s3, mean3=mean3, rstd3=rstd3)
    return x3


def _zero_query_grads(G, gw, gb, d):
    """The query rows of the cross-attention in-projection gradient are exactly zero over a one-position memory.  A
    gradient dict whose rows are known to hold zeros already (a trainer's views of its flat block, which nothing else
    writes: ``G["_cross_q_grads_zero"]``) is left alone -- no fill launches in the steady state."""
    if not G.get("_cross_q_grads_zero", False):
        gw[:d].zero_()
        gb[:d].zero_()
        if "_cross_q_grads_zero" in G:  # only dicts that carry the flag promise that nobody else writes these rows
            G["_cross_q_grads_zero"] = True


def layer_bwd(P, G, sv, dx3, mem, dmem, B, T, S, H, causal, dp, l, dkv=None):
    """Backward of layer_fwd.  Writes parameter grads into G[...] (overwrite); returns (dx, dmem).
    ``dkv``: a [B*S, 2d] row-strided view to receive dK | dV when the caller batches the d(memory) product
    over all layers (stack_bwd); then ``dmem`` is passed through untouched."""
    d = dx3.shape[1]
    p, sid = dp.p, 16 * l
    tr = p > 0.0
    # bf16 storage: the gradients entering the branch dropouts are read by matrix products only -- the LayerNorm
    # backward writes them as bf16 (with p == 0 that is the bf16 copy of dx) and nothing casts them
    d16 = ops.COMPUTE_DTYPE == ops.BF16S
    d16c = d16 and not _single_key_fast(S, d, H)  # the streaming single-key branch reads this one as fp32
    g3, g3m, _, _ = ops.layernorm_bwd(dx3, sv["s3"], P["norm3.weight"], P["norm3.bias"], sv["mean3"], sv["rstd3"],
                                      drop2_p=p, seed2=dp.seed(sid + S_FFN_OUT), want_dx_drop=tr or d16,
                                      dgamma=G["norm3.weight"], dbeta=G["norm3.bias"], dx_drop16=d16)
    g3m = g3m if tr or d16 else g3
    ops.linear_wgrad(g3m, sv["h"], dw=G["linear2.weight"], db=G["linear2.bias"])
    du = ops.linear_dgrad(g3m, P["linear2.weight"], gelu_pre=sv["u"], drop_p=p, seed=dp.seed(sid + S_FFN_IN),
                          out=_like16(sv["h"], *sv["h"].shape))
    ops.linear_wgrad(du, sv["x2"], dw=G["linear1.weight"], db=G["linear1.bias"])
    ops.wgrad_flush(stage=True)
    dx2 = ops.linear_dgrad(du, P["linear1.weight"], addend=g3)
    g2, g2m, _, _ = ops.layernorm_bwd(dx2, sv["s2"], P["norm2.weight"], P["norm2.bias"], sv["mean2"], sv["rstd2"],
                                      drop2_p=p, seed2=dp.seed(sid + S_CA_OUT), want_dx_drop=tr or d16c,
                                      dgamma=G["norm2.weight"], dbeta=G["norm2.bias"], dx_drop16=d16c)
    g2m = g2m if tr or d16c else g2
    kv = sv["kv"]
    batched = dkv is not None
    wc = P["multihead_attn.in_proj_weight"]
    gw, gb = G["multihead_attn.in_proj_weight"], G["multihead_attn.in_proj_bias"]
    if _single_key_fast(S, d, H):  # see layer_fwd: everything of the branch lives on B*H rows
        dU, dbp = ops.cross1_bwd(g2m, B, T, H, p, dp.seed(sid + S_CA_P))
        ops.linear_wgrad(dU, sv["vblk"], dw=G["multihead_attn.out_proj.weight"])
        dvblk = ops.linear_dgrad(dU, P["multihead_attn.out_proj.weight"])  # [B*H, d]; head h's slice of row (b, h) is dv
        if not batched:
            dkv = torch.empty_like(kv)
        # dk == 0 exactly (p = 1 for the only key), as torch computes it; dv = the head-diagonal of dvblk; bias gradient
        ops.cross1_gather(dvblk, dbp, dkv[:, :d], dkv[:, d:], G["multihead_attn.out_proj.bias"], B, H)
        _zero_query_grads(G, gw, gb, d)  # dq == 0 exactly: zero query-projection gradients, nothing flows to x1 through q
    else:
        ops.linear_wgrad(g2m, sv["c"], dw=G["multihead_attn.out_proj.weight"], db=G["multihead_attn.out_proj.bias"])
        dc = ops.linear_dgrad(g2m, P["multihead_attn.out_proj.weight"], out=_like16(sv["c"], g2m.shape[0], d))
        if not batched:
            dkv = torch.empty_like(kv)
        dq, _, _ = ops.attn_bwd(sv["q"], kv[:, :d], kv[:, d:], sv["c"], dc, sv["lse2"], B, H, T, S, False, p,
                                dp.seed(sid + S_CA_P), dq=_like16(sv["q"], g2m.shape[0], d) if S != 1 else None,
                                dk=dkv[:, :d], dv=dkv[:, d:], need_dq=S != 1)
        if S == 1:
            _zero_query_grads(G, gw, gb, d)
        else:
            G["_cross_q_grads_zero"] = False
            ops.linear_wgrad(dq, sv["x1"], dw=gw[:d], db=gb[:d])
    ops.linear_wgrad(dkv, mem, dw=gw[d:], db=gb[d:])
    ops.wgrad_flush(stage=True)
    if not batched:
        dmem = ops.linear_dgrad(dkv, wc[d:], addend=dmem)
    dx1 = g2 if S == 1 else ops.linear_dgrad(dq, wc[:d], addend=g2)
    g1, g1m, _, _ = ops.layernorm_bwd(dx1, sv["s1"], P["norm1.weight"], P["norm1.bias"], sv["mean1"], sv["rstd1"],
                                      drop2_p=p, seed2=dp.seed(sid + S_SA_OUT), want_dx_drop=tr or d16,
                                      dgamma=G["norm1.weight"], dbeta=G["norm1.bias"], dx_drop16=d16)
    g1m = g1m if tr or d16 else g1
    ops.linear_wgrad(g1m, sv["a"], dw=G["self_attn.out_proj.weight"], db=G["self_attn.out_proj.bias"])
    qkv = sv["qkv"]
    da = ops.linear_dgrad(g1m, P["self_attn.out_proj.weight"], out=_like16(qkv, g1m.shape[0], d))
    dqkv = _like16(qkv, qkv.shape[0], 3 * d)
    if dqkv is None:
        dqkv = torch.empty_like(qkv)
    ops.attn_bwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], sv["a"], da, sv["lse1"], B, H, T, T, causal, p,
                 dp.seed(sid + S_SA_P), dq=dqkv[:, :d], dk=dqkv[:, d:2 * d], dv=dqkv[:, 2 * d:])
    ops.linear_wgrad(dqkv, sv["x"], dw=G["self_attn.in_proj_weight"], db=G["self_attn.in_proj_bias"])
    ops.wgrad_flush()  # the layer's weight gradients: one grouped launch on the side lane (ops.GROUP_WGRADS)
    dx = ops.linear_dgrad(dqkv, P["self_attn.in_proj_weight"], addend=g1)
    return dx, dmem


# ---------------------------------------------------------------------------------------------
# decoder stack (+ final norm), feature projection, output head
# ---------------------------------------------------------------------------------------------
def kv_batched(n_layers, mem_rows):
    """Whether stack_fwd computes the cross-attention K | V projections of all layers as ONE product."""
    return n_layers > 1 and mem_rows >= 1024


def stack_fwd(layers, norm_w, norm_b, x, mem, B, T, S, H, causal, dp, save, kv_stack=None, plans=None):
    """``kv_stack``: (w_all [L*2d, d], b_all [L*2d]), the layers' cross-attention K | V projection weights / biases
    already stacked (the trainer keeps such a block and refreshes it with one launch per step: ops.StackedRows);
    None: concatenated here.  ``plans``: a train_plan.LayerPlans built over these layers for this geometry -- each
    layer is then ONE C call per direction (same launches, same bits) instead of the op-by-op plan below."""
    saves = [] if save is not None else None
    # The cross-attention K/V projections of all layers depend only on the memory: one GEMM with the layers'
    # weights stacked ([L*2d, d], a 6 MB gather at C2) instead of L mid-size ones; each layer reads its
    # [B*S, 2d] column block in place (the attention kernels take row strides).
    d = x.shape[1]
    kv_all = w_all = None
    if kv_batched(len(layers), mem.shape[0]):
        if kv_stack is not None:
            w_all, b_all = kv_stack
        else:
            w_all = torch.cat([P["multihead_attn.in_proj_weight"][d:] for P in layers])
            b_all = torch.cat([P["multihead_attn.in_proj_bias"][d:] for P in layers])
        # bf16 storage: the block exists only as bf16 (capdec_epilogue.c16) and the all-bf16 attention kernels stream it
        kv_all = ops.linear_fwd(mem, w_all, b_all, out=ops.bf16_matrix(mem.shape[0], w_all.shape[0], x.device)
                                if _attn16(H, d) and S > 1 else None)
    if plans is not None and (kv_all is None or save is None):
        raise ValueError("stack_fwd: one-call layer plans need the batched K | V projection and a training pass")
    if plans is not None:
        x = plans.forward(x, kv_all, dp.key)
    else:
        for l, P in enumerate(layers):
            sv = {} if save is not None else None
            x = layer_fwd(P, x, mem, B, T, S, H, causal, dp, l, sv,
                          kv=None if kv_all is None else kv_all[:, l * 2 * d:(l + 1) * 2 * d])
            if saves is not None:
                saves.append(sv)
    y, mean, rstd = ops.layernorm_fwd(x, norm_w, norm_b, save_stats=save is not None)
    if save is not None:
        save.update(layers=saves, xl=x, mean=mean, rstd=rstd, w_kv_all=w_all if kv_all is not None else None,
                    plans=plans, kv_all=kv_all if plans is not None else None)
    return y


def stack_bwd(layers, grads, norm_w, norm_b, g_norm_w, g_norm_b, sv, dy, mem, B, T, S, H, causal, dp,
              after_layer=None):
    """``after_layer(l)`` is called once layer l's parameter grads are enqueued (bucketed all-reduce hook)."""
    dx, _, _, _ = ops.layernorm_bwd(dy, sv["xl"], norm_w, norm_b, sv["mean"], sv["rstd"], dgamma=g_norm_w,
                                    dbeta=g_norm_b)
    dmem = None
    # forward batched the cross K/V projections: d(memory) = [dKV_0 | ... | dKV_L-1] @ [W_0; ...; W_L-1] is then
    # ONE product over K = L*2d after the last layer instead of L accumulating ones (no parameter gradient is
    # involved, so the bucket order of the data-parallel reducer is untouched)
    w_all = sv.get("w_kv_all")
    d = dy.shape[1]
    dkv_all = None
    if w_all is not None:
        if ops.COMPUTE_DTYPE == ops.BF16S and S > 1:
            # the only readers of dK | dV are bf16-storage products: the attention backward writes their operand
            # directly (no fp32 block, no cast pass); the padding columns of the pitch must read as zeros
            pitch = ops._bf16_pitch(w_all.shape[0])
            buf = torch.empty((mem.shape[0], pitch), dtype=torch.bfloat16, device=dy.device)
            dkv_all = buf[:, :w_all.shape[0]]
            if w_all.shape[0] % 64:  # (a width that is a multiple of 64 has no reduction tail: the padding is never read)
                buf[:, w_all.shape[0]:].zero_()
                dkv_all = ops._zero_padded(dkv_all)
        else:
            dkv_all = torch.empty((mem.shape[0], w_all.shape[0]), dtype=torch.float32, device=dy.device)
    if sv.get("plans") is not None:
        for G in grads:  # (the plans cover S > 1 only: the query rows of the cross-attention in-projection get real gradients)
            if "_cross_q_grads_zero" in G:
                G["_cross_q_grads_zero"] = False
        dx = sv["plans"].backward(dx, mem, sv["kv_all"], dkv_all, dp.key, after_layer)
    else:
        for l in range(len(layers) - 1, -1, -1):
            dx, dmem = layer_bwd(layers[l], grads[l], sv["layers"][l], dx, mem, dmem, B, T, S, H, causal, dp, l,
                                 dkv=None if dkv_all is None else dkv_all[:, l * 2 * d:(l + 1) * 2 * d])
            if after_layer is not None:
                after_layer(l)
    if dkv_all is not None:
        dmem = ops.linear_dgrad(dkv_all, w_all)
    return dx, dmem


def proj_fwd(P, f, dp, save):
    """feature_projection (ref models.py:37-44): Linear, LN, GELU, Dropout, Linear, LN on f [R,Din]."""
    h0 = ops.linear_fwd(f, P["0.weight"], P["0.bias"])
    h1, mean1, rstd1 = ops.layernorm_fwd(h0, P["1.weight"], P["1.bias"], act=ACT_GELU, drop_p=dp.p,
                                         seed=dp.seed(S_PROJ), save_stats=save is not None)
    h2 = ops.linear_fwd(h1, P["4.weight"], P["4.bias"])
    mem, mean5, rstd5 = ops.layernorm_fwd(h2, P["5.weight"], P["5.bias"], save_stats=save is not None)
    if save is not None:
        save.update(f=f, h0=h0, mean1=mean1, rstd1=rstd1, h1=h1, h2=h2, mean5=mean5, rstd5=rstd5)
    return mem


def proj_bwd(P, G, sv, dmem, dp, want_df=False):
    d16 = ops.COMPUTE_DTYPE == ops.BF16S  # both dx feed matrix products only: bf16 copies straight from the kernel
    g5, g5c, _, _ = ops.layernorm_bwd(dmem, sv["h2"], P["5.weight"], P["5.bias"], sv["mean5"], sv["rstd5"],
                                      dgamma=G["5.weight"], dbeta=G["5.bias"], want_dx_drop=d16, dx_drop16=d16)
    g5 = g5c if d16 else g5
    ops.linear_wgrad(g5, sv["h1"], dw=G["4.weight"], db=G["4.bias"])
    dh1 = ops.linear_dgrad(g5, P["4.weight"])
    g1, g1c, _, _ = ops.layernorm_bwd(dh1, sv["h0"], P["1.weight"], P["1.bias"], sv["mean1"], sv["rstd1"], act=ACT_GELU,
                                      drop_p=dp.p, seed=dp.seed(S_PROJ), dgamma=G["1.weight"], dbeta=G["1.bias"],
                                      want_dx_drop=d16, dx_drop16=d16)
    g1 = g1c if d16 else g1
    ops.linear_wgrad(g1, sv["f"], dw=G["0.weight"], db=G["0.bias"])
    ops.wgrad_flush()
    return ops.linear_dgrad(g1, P["0.weight"]) if want_df else None


def head_fwd(P, x, dp, save, logits_out=None, ce_partials=None, ce_targets=None):
    """output_layer (ref models.py:68-73): Linear, GELU, Dropout, Linear on x [M,d] -> logits [M,V].
    ``ce_partials``: the vocabulary product's epilogue also leaves the per-tile softmax partials of its logits there
    (label-smoothed cross-entropy fused into the product, ref train.py:41-50)."""
    u = torch.empty((x.shape[0], P["0.weight"].shape[0]), dtype=torch.float32, device=x.device) \
        if save is not None else None
    h = ops.linear_fwd(x, P["0.weight"], P["0.bias"], act=ACT_GELU, pre=u, drop_p=dp.p, seed=dp.seed(S_HEAD),
                       out=_hidden16(x.shape[0], P["0.weight"].shape[0], x.device))
    logits = ops.linear_fwd(h, P["3.weight"], P["3.bias"], out=logits_out, ce_partials=ce_partials, ce_targets=ce_targets)
    if save is not None:
        save.update(x=x, u=u, h=h)
    return logits


def head_bwd(P, G, sv, dlogits, dp, ce=None):
    """``ce = (lse, targets, eps, grad_scale)``: ``dlogits`` is still the LOGITS block; both products read it through
    the cross-entropy gradient transform in their loaders (nothing is written back to the block)."""
    ops.linear_wgrad(dlogits, sv["h"], dw=G["3.weight"], db=G["3.bias"], ce=ce)
    ops.wgrad_flush()  # the big one: on the lane before the input-gradient product is enqueued, so that they overlap
    du = ops.linear_dgrad(dlogits, P["3.weight"], gelu_pre=sv["u"], drop_p=dp.p, seed=dp.seed(S_HEAD), ce=ce,
                          out=_like16(sv["h"], *sv["h"].shape))
    ops.linear_wgrad(du, sv["x"], dw=G["0.weight"], db=G["0.bias"])
    ops.wgrad_flush()
    return ops.linear_dgrad(du, P["0.weight"])
