"""KV-cached greedy / beam-search decode engine for the caption decoder.

The reference decodes by re-running the WHOLE prefix through all six layers at every step with a
Python loop of per-sample ``.item()`` syncs (ref train.py:312-383): O(T^2) layer-tokens per caption.
This engine runs one token per step:

* the image memory is projected once; each layer's cross-attention K/V (``memory @ Wkv``) is computed
  once per image and shared by all beams of that image;
* self-attention K/V of the new token are written by the projection GEMM straight into the cache
  (``[rows, Tmax, 2d]`` per layer, the GEMM's output stride does the scatter);
* beam reorder never moves the cache: the top-k kernel rewrites an int ancestry table that the
  attention kernel follows (csrc/decode.hip);
* nothing synchronises with the host inside the loop (finished rows keep emitting PAD).

Greedy and beam search are new capabilities (the reference only samples); their
definition and oracle is ``oracle/torch_oracle.py``: START first, ties to the lowest index, END
freezes a hypothesis (it then proposes only PAD at unchanged score), no length normalisation.
"""
import math

import torch

from . import _capi, engine, ops
from ._capi import ACT_GELU, check, lib, ptr, stream

START = 1


class CaptionDecoder:
    def __init__(self, model, dtype="fp32"):
        self.model = model.eval()
        self.dtype = dtype
        dec = model.transformer_decoder
        self.H, self.L = dec.nhead, len(dec.layers)
        self.layers = [layer.param_dict() for layer in dec.layers]
        self.norm = (dec.norm.weight, dec.norm.bias)
        ol = model.output_layer
        self.head = dict(zip(engine.HEAD_KEYS, [ol[0].weight, ol[0].bias, ol[3].weight, ol[3].bias]))
        fp = model.feature_projection
        self.proj = dict(zip(engine.PROJ_KEYS, [fp[0].weight, fp[0].bias, fp[1].weight, fp[1].bias, fp[4].weight,
                                                fp[4].bias, fp[5].weight, fp[5].bias]))
        self.d = model.embedding.weight.shape[1]
        self.V = model.embedding.weight.shape[0]
        self.dp = engine.DropoutPlan(0.0)

    # ---- one decode step for R = N*W rows at position t ------------------------------------------
    def _step(self, tok, t, st):
        m, d, H = self.model, self.d, self.H
        R, W, S, Tmax = st["R"], st["W"], st["S"], st["Tmax"]
        dh = d // H
        scale = 1.0 / math.sqrt(dh)
        l = lib()
        x = ops.embed_fwd(tok.view(R, 1), m.embedding.weight, m.pos_encoder.table(), t0=t).view(R, d)
        for li, P in enumerate(self.layers):
            w_in, b_in = P["self_attn.in_proj_weight"], P["self_attn.in_proj_bias"]
            q = ops.linear_fwd(x, w_in[:d], b_in[:d])
            cache = st["cache"][li]  # [R, Tmax, 2d]
            ops.linear_fwd(x, w_in[d:], b_in[d:], out=cache[:, t, :])  # scatter via the output row stride
            a = torch.empty((R, d), dtype=torch.float32, device=x.device)
            c2 = cache.view(R * Tmax, 2 * d)
            check(l.capdec_decode_attn(ptr(q), d, ptr(c2), ptr(c2[:, d:]), 2 * d, ptr(a), d, ptr(st["anc"][st["cur"]]),
                                       R, H, dh, Tmax, t, W, S, scale, stream()), "capdec_decode_attn")
            s1 = ops.linear_fwd(a, P["self_attn.out_proj.weight"], P["self_attn.out_proj.bias"], addend=x)
            x1, _, _ = ops.layernorm_fwd(s1, P["norm1.weight"], P["norm1.bias"], save_stats=False)
            wc, bc = P["multihead_attn.in_proj_weight"], P["multihead_attn.in_proj_bias"]
            q = ops.linear_fwd(x1, wc[:d], bc[:d])
            kv = st["cross_kv"][li]  # [N*S, 2d]
            c = torch.empty((R, d), dtype=torch.float32, device=x.device)
            check(l.capdec_decode_attn(ptr(q), d, ptr(kv), ptr(kv[:, d:]), 2 * d, ptr(c), d, None, R, H, dh, Tmax, t, W,
                                       S, scale, stream()), "capdec_decode_attn")
            s2 = ops.linear_fwd(c, P["multihead_attn.out_proj.weight"], P["multihead_attn.out_proj.bias"], addend=x1)
            x2, _, _ = ops.layernorm_fwd(s2, P["norm2.weight"], P["norm2.bias"], save_stats=False)
            h = ops.linear_fwd(x2, P["linear1.weight"], P["linear1.bias"], act=ACT_GELU)
            s3 = ops.linear_fwd(h, P["linear2.weight"], P["linear2.bias"], addend=x2)
            x, _, _ = ops.layernorm_fwd(s3, P["norm3.weight"], P["norm3.bias"], save_stats=False)
        y, _, _ = ops.layernorm_fwd(x, self.norm[0], self.norm[1], save_stats=False)
        return engine.head_fwd(self.head, y, self.dp, None)

    def _prepare(self, features, W, Tmax):
        m, d = self.model, self.d
        dev = features.device
        N = features.shape[0]
        S = 1 if features.dim() == 2 else features.shape[1]
        cross = []
        with ops.compute_dtype(self.dtype):
            mem = engine.proj_fwd(self.proj, features.reshape(N * S, -1), self.dp, None)  # [N*S, d]
            for P in self.layers:
                wc, bc = P["multihead_attn.in_proj_weight"], P["multihead_attn.in_proj_bias"]
                cross.append(ops.linear_fwd(mem, wc[d:], bc[d:]))
        R = N * W
        st = dict(N=N, W=W, R=R, S=S, Tmax=Tmax, cross_kv=cross, cur=0,
                  cache=[torch.empty((R, Tmax, 2 * d), dtype=torch.float32, device=dev) for _ in self.layers],
                  anc=[torch.arange(R, dtype=torch.int32, device=dev).view(R, 1).repeat(1, Tmax).contiguous()
                       for _ in range(2)],
                  seq=[torch.zeros((R, Tmax), dtype=torch.int32, device=dev) for _ in range(2)])
        return st

    @torch.no_grad()
    def greedy(self, features, max_len=20, return_gap=False):
        """ids [N, max_len] int64 (without START; PAD after END)."""
        st = self._prepare(features, 1, max_len)
        R, dev = st["R"], features.device
        tok = torch.full((R,), START, dtype=torch.int64, device=dev)
        done = torch.zeros(R, dtype=torch.int32, device=dev)
        gap = torch.full((R,), float("inf"), dtype=torch.float32, device=dev)
        l = lib()
        for t in range(max_len):
            with ops.compute_dtype(self.dtype):
                logits = self._step(tok, t, st)
            check(l.capdec_greedy_select(ptr(logits), logits.stride(0), ptr(done), ptr(st["seq"][0]), ptr(tok), ptr(gap),
                                         R, self.V, max_len, t, stream()), "capdec_greedy_select")
        ids = st["seq"][0].to(torch.int64)
        return (ids, gap) if return_gap else ids

    @torch.no_grad()
    def beam(self, features, beam=3, max_len=20, return_gap=False):
        """ids [N, beam, max_len] int64 (best first), scores [N, beam] fp32 (sum of log-probs)."""
        st = self._prepare(features, beam, max_len)
        N, R, dev = st["N"], st["R"], features.device
        tok = torch.full((R,), START, dtype=torch.int64, device=dev)
        score = [torch.full((N, beam), -1.0e30, dtype=torch.float32, device=dev) for _ in range(2)]
        score[0][:, 0] = 0.0
        fin = [torch.zeros(R, dtype=torch.int32, device=dev) for _ in range(2)]
        gap = torch.full((N,), float("inf"), dtype=torch.float32, device=dev)
        l = lib()
        cur = 0
        for t in range(max_len):
            st["cur"] = cur
            with ops.compute_dtype(self.dtype):
                logits = self._step(tok, t, st)
            nxt = cur ^ 1
            check(l.capdec_beam_select(ptr(logits), logits.stride(0), ptr(score[cur]), ptr(score[nxt]), ptr(fin[cur]),
                                       ptr(fin[nxt]), ptr(st["anc"][cur]), ptr(st["anc"][nxt]), ptr(st["seq"][cur]),
                                       ptr(st["seq"][nxt]), ptr(tok), ptr(gap), N, beam, self.V, max_len, t, stream()),
                  "capdec_beam_select")
            cur = nxt
        ids = st["seq"][cur].to(torch.int64).view(N, beam, max_len)
        out = (ids, score[cur])
        return out + (gap,) if return_gap else out
