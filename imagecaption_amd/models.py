"""Drop-in replacement for the reference's ``models.py`` (decoder half) on MI355X.

Same class names, constructor / ``forward`` signatures, sub-module attribute names, calling
conventions and ``state_dict`` keys as /root/reference/models.py, so the reference's ``main.py``,
``train.py`` and ``test.py`` run on top unchanged.  The arithmetic is done by
hand-written gfx950 HIP kernels behind the C ABI of ``include/capdec.h``; there is no torch.nn
compute and no CPU fallback in here.

Differences, all on the out-of-scope encoder side:
* ``.vit`` is a pass-through: ``images`` are cached / synthetic features ``[B, feature_dim]`` (memory
  length 1, models.py:104) or grid features ``[B, S, feature_dim]`` (memory length S).  A reference
  checkpoint's ``vit.*`` keys are therefore absent here: load it with ``strict=False``.
* ``feature_dim`` (default 768, the literal of models.py:38) is an extra keyword.
"""
import math

import torch
import torch.nn as nn

from . import engine
from ._capi import ACT_NONE
from .engine import DropoutPlan
from .functional import DecoderStackFn, EmbedFn, HeadFn, LayerNormFn, LinearFn, PEAddFn, ProjectionFn

_GLOBAL_CALL = [0]


def _next_call():
    _GLOBAL_CALL[0] += 1
    return _GLOBAL_CALL[0]


def _plan(module, p):
    """Dropout plan for one forward call: active only in training mode."""
    if module.training and p > 0.0:
        return DropoutPlan(p, torch.initial_seed(), _next_call())
    return DropoutPlan(0.0)


class PositionalEncoding(nn.Module):
    """ref models.py:6-19.  ``pe`` is a persistent buffer [max_len, 1, d_model] (same state_dict key)."""

    def __init__(self, d_model, dropout=0.1, max_len=5000):
        super().__init__()
        self.p = float(dropout)
        position = torch.arange(max_len).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
        pe = torch.zeros(max_len, 1, d_model)
        pe[:, 0, 0::2] = torch.sin(position * div_term)
        pe[:, 0, 1::2] = torch.cos(position * div_term)
        self.register_buffer("pe", pe)

    def table(self):
        return self.pe.view(self.pe.shape[0], self.pe.shape[2])

    def forward(self, x):  # x: [T, B, d] seq-first (ref models.py:112-114)
        dp = _plan(self, self.p)
        return PEAddFn.apply(x, self.table(), dp.p, dp.seed(engine.S_EMBED))


class Embedding(nn.Module):
    """nn.Embedding stand-in (ref models.py:47): key ``weight`` [V, d], N(0,1) init like torch's."""

    def __init__(self, num_embeddings, embedding_dim):
        super().__init__()
        self.num_embeddings, self.embedding_dim = num_embeddings, embedding_dim
        self.weight = nn.Parameter(torch.empty(num_embeddings, embedding_dim))
        nn.init.normal_(self.weight)
        self.register_buffer("_zero_pe", torch.zeros(1, embedding_dim), persistent=False)

    def forward(self, ids):
        if ids.dim() == 1:
            ids = ids.unsqueeze(0)
        T = ids.shape[1]
        if self._zero_pe.shape[0] < T:
            self._zero_pe = torch.zeros(T, self.embedding_dim, device=self.weight.device)
        return EmbedFn.apply(ids, self.weight, self._zero_pe, 0, 0.0, 0)


class Linear(nn.Linear):
    """nn.Linear parameters (same keys / init), HIP GEMM forward with optional fused GELU + dropout."""

    def __init__(self, in_features, out_features, act=ACT_NONE, dropout=0.0):
        super().__init__(in_features, out_features)
        self.act, self.p = act, float(dropout)

    def forward(self, x):
        dp = _plan(self, self.p)
        return LinearFn.apply(x, self.weight, self.bias, self.act, dp.p, dp.seed(7))


class LayerNorm(nn.LayerNorm):
    def __init__(self, d, act=ACT_NONE, dropout=0.0):
        super().__init__(d)
        self.act, self.p = act, float(dropout)

    def forward(self, x):
        dp = _plan(self, self.p)
        return LayerNormFn.apply(x, self.weight, self.bias, self.eps, self.act, dp.p, dp.seed(8))


class _Passthrough(nn.Module):
    """Placeholder keeping the reference's Sequential indices (GELU / Dropout slots are fused away)."""

    def forward(self, x):
        return x


class FeatureProjection(nn.Sequential):
    """ref models.py:37-44 with the same child indices 0..5 (so keys ``feature_projection.{0,1,4,5}.*``)."""

    def __init__(self, feature_dim, d_model, dropout=0.1):
        super().__init__(Linear(feature_dim, d_model * 2), LayerNorm(d_model * 2), _Passthrough(), _Passthrough(),
                         Linear(d_model * 2, d_model), LayerNorm(d_model))
        self.p = float(dropout)

    def forward(self, features):
        dp = _plan(self, self.p)
        params = [self[0].weight, self[0].bias, self[1].weight, self[1].bias, self[4].weight, self[4].bias,
                  self[5].weight, self[5].bias]
        return ProjectionFn.apply(features, dp, *params)


class OutputLayer(nn.Sequential):
    """ref models.py:68-73 with the same child indices (keys ``output_layer.{0,3}.*``)."""

    def __init__(self, d_model, vocab_size, dropout=0.1):
        super().__init__(Linear(d_model, d_model * 2), _Passthrough(), _Passthrough(), Linear(d_model * 2, vocab_size))
        self.p = float(dropout)

    def forward(self, x):
        dp = _plan(self, self.p)
        return HeadFn.apply(x, dp, self[0].weight, self[0].bias, self[3].weight, self[3].bias)


class _OutProj(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(d, d))
        self.bias = nn.Parameter(torch.zeros(d))


class _Attention(nn.Module):
    """Parameter holder with nn.MultiheadAttention's keys and defaults (activation.py:1093-1160)."""

    def __init__(self, d):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d, d))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d))
        self.out_proj = _OutProj(d)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.kaiming_uniform_(self.out_proj.weight, a=math.sqrt(5))


class TransformerDecoderLayer(nn.Module):
    """Parameter holder with nn.TransformerDecoderLayer's keys (transformer.py:1040-1080)."""

    def __init__(self, d_model, nhead, dim_feedforward):
        super().__init__()
        self.self_attn = _Attention(d_model)
        self.multihead_attn = _Attention(d_model)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1, self.norm2, self.norm3 = nn.LayerNorm(d_model), nn.LayerNorm(d_model), nn.LayerNorm(d_model)

    def param_dict(self):
        sd = dict(self.named_parameters())
        return {k: sd[k] for k in engine.LAYER_KEYS}


class TransformerDecoder(nn.Module):
    """``nn.TransformerDecoder(layer, num_layers, norm=LayerNorm)`` stand-in: keys ``layers.{i}.*`` and ``norm.*``.

    ``forward(tgt [B,t,d], memory [B,S,d], tgt_mask=mask)``: a given ``tgt_mask`` must be the square
    causal mask of ``generate_square_subsequent_mask`` (the only mask the reference ever passes,
    train.py:33, 314); it is verified once per size and then never read -- causality is implicit
    in the attention kernel, as torch does when it detects a causal mask (transformer.py:641, 1243-1252).
    """

    def __init__(self, d_model, nhead, num_layers, dim_feedforward, dropout=0.1):
        super().__init__()
        self.layers = nn.ModuleList([TransformerDecoderLayer(d_model, nhead, dim_feedforward) for _ in range(num_layers)])
        self.norm = nn.LayerNorm(d_model)
        self.nhead, self.p = nhead, float(dropout)
        self._mask_ok = set()

    def _check_mask(self, mask, T):
        if tuple(mask.shape) != (T, T):
            raise RuntimeError(f"tgt_mask has shape {tuple(mask.shape)}, expected {(T, T)}")
        if T not in self._mask_ok:
            ref = torch.triu(torch.full((T, T), float("-inf")), diagonal=1)
            if not torch.equal(mask.detach().float().cpu(), ref):
                raise RuntimeError("only the square causal tgt_mask (generate_square_subsequent_mask) is supported")
            self._mask_ok.add(T)

    def forward(self, tgt, memory, tgt_mask=None, tgt_is_causal=None, **unsupported):
        for k, v in unsupported.items():
            if v is not None:
                raise RuntimeError(f"TransformerDecoder.forward: argument {k} is not supported by the HIP path")
        causal = tgt_mask is not None or bool(tgt_is_causal)
        if tgt_mask is not None:
            self._check_mask(tgt_mask, tgt.shape[1])
        meta = dict(nhead=self.nhead, causal=causal, dropout=_plan(self, self.p), num_layers=len(self.layers))
        params = []
        for layer in self.layers:
            pd = layer.param_dict()
            params.extend(pd[k] for k in engine.LAYER_KEYS)
        params.extend([self.norm.weight, self.norm.bias])
        return DecoderStackFn.apply(tgt, memory, meta, *params)


class ImageCaptioningModel(nn.Module):
    """ref models.py:21-130 (decoder half on HIP kernels; ``vit`` is a feature pass-through)."""

    def __init__(self, vocab_size, d_model=512, nhead=8, num_decoder_layers=6, feature_dim=768):
        super().__init__()
        self.vit = nn.Identity()
        self.feature_projection = FeatureProjection(feature_dim, d_model)
        self.embedding = Embedding(vocab_size, d_model)
        self.pos_encoder = PositionalEncoding(d_model)
        self.transformer_decoder = TransformerDecoder(d_model, nhead, num_decoder_layers, d_model * 4)
        self.output_layer = OutputLayer(d_model, vocab_size)
        self._init_parameters()

    def _init_parameters(self):  # ref models.py:78-82
        for p in self.parameters():
            if p.dim() > 1 and p.requires_grad:
                nn.init.xavier_uniform_(p)

    def generate_square_subsequent_mask(self, sz):  # ref models.py:84-88 (CPU float mask, 0 / -inf)
        mask = (torch.triu(torch.ones(sz, sz)) == 1).transpose(0, 1)
        return mask.float().masked_fill(mask == 0, float("-inf")).masked_fill(mask == 1, float(0.0))

    def project(self, images):
        memory = self.feature_projection(self.vit(images))
        return memory.unsqueeze(1) if memory.dim() == 2 else memory  # ref models.py:104

    def forward(self, images, captions=None, tgt_mask=None):
        memory = self.project(images)
        if captions is None:
            return memory  # ref models.py:126-128
        dp = _plan(self, self.pos_encoder.p)
        # Embedding + transpose + PositionalEncoding + transpose of ref models.py:109-114, one kernel
        tgt = EmbedFn.apply(captions, self.embedding.weight, self.pos_encoder.table(), 0, dp.p,
                            dp.seed(engine.S_EMBED))
        out = self.transformer_decoder(tgt, memory, tgt_mask=tgt_mask)
        return self.output_layer(out)
