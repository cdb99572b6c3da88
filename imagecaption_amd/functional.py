"""torch.autograd wrappers that put the HIP launch plans of ``engine.py`` behind ordinary tensors.

These make the drop-in ``nn.Module`` surface differentiable, so the reference's unchanged
``train.py`` (``loss.backward()``, ``clip_grad_norm_``, ``optimizer.step()``) works on top.
"""
import torch

from . import engine, ops
from ._capi import ACT_GELU


def _flat2(x):
    return x.reshape(-1, x.shape[-1]).contiguous()


class DecoderStackFn(torch.autograd.Function):
    """All decoder layers + the final LayerNorm (nn.TransformerDecoder.forward, transformer.py:599-658)."""

    @staticmethod
    def forward(ctx, tgt, memory, meta, *params):
        H, causal, dp, nl = meta["nhead"], meta["causal"], meta["dropout"], meta["num_layers"]
        B, T, d = tgt.shape
        S = memory.shape[1]
        nk = len(engine.LAYER_KEYS)
        layers = [dict(zip(engine.LAYER_KEYS, params[i * nk:(i + 1) * nk])) for i in range(nl)]
        norm_w, norm_b = params[nl * nk], params[nl * nk + 1]
        need = any(ctx.needs_input_grad)
        save = {} if need else None
        x, mem = _flat2(tgt), _flat2(memory)
        y = engine.stack_fwd(layers, norm_w, norm_b, x, mem, B, T, S, H, causal, dp, save)
        if need:
            ctx.sv, ctx.layers, ctx.norm, ctx.mem = save, layers, (norm_w, norm_b), mem
            ctx.dims = (B, T, S, H, causal, dp, d)
        return y.view(B, T, d)

    @staticmethod
    def backward(ctx, dy):
        B, T, S, H, causal, dp, d = ctx.dims
        grads = [{k: torch.empty_like(v) for k, v in P.items()} for P in ctx.layers]
        gnw, gnb = torch.empty_like(ctx.norm[0]), torch.empty_like(ctx.norm[1])
        dx, dmem = engine.stack_bwd(ctx.layers, grads, ctx.norm[0], ctx.norm[1], gnw, gnb, ctx.sv, _flat2(dy), ctx.mem,
                                    B, T, S, H, causal, dp)
        flat = [g[k] for g in grads for k in engine.LAYER_KEYS] + [gnw, gnb]
        ctx.sv = None
        return (dx.view(B, T, d), dmem.view(B, S, d), None) + tuple(flat)


class ProjectionFn(torch.autograd.Function):
    """feature_projection as one fused plan (ref models.py:37-44)."""

    @staticmethod
    def forward(ctx, features, dp, *params):
        P = dict(zip(engine.PROJ_KEYS, params))
        need = any(ctx.needs_input_grad)
        save = {} if need else None
        out = engine.proj_fwd(P, _flat2(features), dp, save)
        if need:
            ctx.sv, ctx.P, ctx.dp, ctx.want_df = save, P, dp, features.requires_grad
            ctx.in_shape = features.shape
        return out.view(*features.shape[:-1], out.shape[-1])

    @staticmethod
    def backward(ctx, dmem):
        G = {k: torch.empty_like(v) for k, v in ctx.P.items()}
        df = engine.proj_bwd(ctx.P, G, ctx.sv, _flat2(dmem), ctx.dp, want_df=ctx.want_df)
        ctx.sv = None
        return (df.view(ctx.in_shape) if df is not None else None, None) + tuple(G[k] for k in engine.PROJ_KEYS)


class HeadFn(torch.autograd.Function):
    """output_layer as one fused plan (ref models.py:68-73)."""

    @staticmethod
    def forward(ctx, x, dp, *params):
        P = dict(zip(engine.HEAD_KEYS, params))
        need = any(ctx.needs_input_grad)
        save = {} if need else None
        logits = engine.head_fwd(P, _flat2(x), dp, save)
        if need:
            ctx.sv, ctx.P, ctx.dp, ctx.in_shape = save, P, dp, x.shape
        return logits.view(*x.shape[:-1], logits.shape[-1])

    @staticmethod
    def backward(ctx, dlogits):
        G = {k: torch.empty_like(v) for k, v in ctx.P.items()}
        dx = engine.head_bwd(ctx.P, G, ctx.sv, _flat2(dlogits), ctx.dp)
        ctx.sv = None
        return (dx.view(ctx.in_shape), None) + tuple(G[k] for k in engine.HEAD_KEYS)


class EmbedFn(torch.autograd.Function):
    """E[ids] (+ pe[t0+t]) (+ dropout): ref models.py:109-114 in one kernel."""

    @staticmethod
    def forward(ctx, ids, weight, pe, t0, p, seed):
        ctx.save_for_backward(ids)
        ctx.V, ctx.p, ctx.seed = weight.shape[0], p, seed
        return ops.embed_fwd(ids, weight, pe, t0, p, seed)

    @staticmethod
    def backward(ctx, dout):
        (ids,) = ctx.saved_tensors
        return None, ops.embed_bwd(ids, dout, ctx.V, ctx.p, ctx.seed), None, None, None, None


class PEAddFn(torch.autograd.Function):
    """PositionalEncoding.forward on a seq-first tensor (ref models.py:17-19)."""

    @staticmethod
    def forward(ctx, x, pe, p, seed):
        T, B = x.shape[0], x.shape[1]
        ctx.p, ctx.seed = p, seed
        return ops.pe_add(x, pe, div=B, mod=T, drop_p=p, seed=seed)

    @staticmethod
    def backward(ctx, dout):
        if ctx.p > 0.0:
            return ops.act_dropout_bwd(dout, None, ctx.p, ctx.seed), None, None, None
        return dout, None, None, None


class LinearFn(torch.autograd.Function):
    """One Linear (+ optional fused GELU / dropout), for sub-modules called on their own."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, p, seed):
        x2 = _flat2(x)
        pre = None
        need = any(ctx.needs_input_grad)
        if act == ACT_GELU and need:
            pre = torch.empty((x2.shape[0], weight.shape[0]), dtype=torch.float32, device=x.device)
        y = ops.linear_fwd(x2, weight, bias, act=act, pre=pre, drop_p=p, seed=seed)
        ctx.save_for_backward(x2, weight, pre)
        ctx.meta = (act, p, seed, x.shape, bias is not None)
        return y.view(*x.shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, weight, pre = ctx.saved_tensors
        act, p, seed, shape, has_bias = ctx.meta
        dy2 = _flat2(dy)
        if act == ACT_GELU or p > 0.0:
            dy2 = ops.act_dropout_bwd(dy2, pre, p, seed)
        dw = torch.empty_like(weight)
        db = torch.empty(weight.shape[0], dtype=torch.float32, device=dy.device) if has_bias else None
        ops.linear_wgrad(dy2, x2, dw=dw, db=db)
        dx = ops.linear_dgrad(dy2, weight)
        return dx.view(shape), dw, db, None, None, None


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps, act, p, seed):
        x2 = _flat2(x)
        y, mean, rstd = ops.layernorm_fwd(x2, weight, bias, eps, act, p, seed)
        ctx.save_for_backward(x2, weight, bias, mean, rstd)
        ctx.meta = (act, p, seed, x.shape)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, weight, bias, mean, rstd = ctx.saved_tensors
        act, p, seed, shape = ctx.meta
        dx, _, dg, db = ops.layernorm_bwd(_flat2(dy), x2, weight, bias, mean, rstd, act, p, seed)
        return dx.view(shape), dg, db, None, None, None, None


class SoftmaxCEFn(torch.autograd.Function):
    """Label-smoothed CE, mean over all rows (ref train.py:41-50); dlogits computed in the forward pass."""

    @staticmethod
    def forward(ctx, logits, targets, eps):
        l2 = _flat2(logits)
        rows, dl = ops.softmax_ce(l2, targets.reshape(-1), eps, want_grad=True)
        ctx.save_for_backward(dl)
        ctx.shape = logits.shape
        return ops.reduce_sum(rows, 1.0 / rows.numel()).reshape(())

    @staticmethod
    def backward(ctx, dloss):
        (dl,) = ctx.saved_tensors
        return (dl * dloss).view(ctx.shape), None, None


def label_smoothed_cross_entropy(logits, targets, smooth_factor=0.1):
    return SoftmaxCEFn.apply(logits, targets, smooth_factor)
