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

``sample`` / ``generate_caption`` are the reference's own sampling decode (train.py:289-412) on this
engine: the per-row sentence counters, logit biases, top-5 draw and the loop's stop rule run in one
kernel per step (csrc/decode.hip sample_select_kernel) instead of ``B`` ``.item()`` syncs per step.

Greedy and beam search are new capabilities (the reference only samples); their
definition and oracle is ``oracle/torch_oracle.py``: START first, ties to the lowest index, END
freezes a hypothesis (it then proposes only PAD at unchanged score), no length normalisation.
"""
import math

import torch

import ctypes as C

from . import _capi, engine, ops
from ._capi import ACT_GELU, check, lib, ptr, stream

START = 1


class CaptionDecoder:
    def __init__(self, model, dtype="fp32", fused_step=True):
        """``fused_step``: the decoder stack of a decode step is enqueued by ONE C call (capdec_decode_step) instead
        of ~10 ops per layer issued from Python: same kernels, bit-identical tokens, a fraction of the host time
        (what bounds small batches).  False keeps the op-by-op path (the parity reference of the fused call)."""
        self.model = model.eval()
        self.dtype = dtype
        self.fused_step = fused_step
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
    def _step(self, tok, t, st, collapse=False):
        """``collapse`` (beam search, t = 0): every beam of an image is the same <START> hypothesis, so the step
        runs for ONE row per image; its K/V land in the cache slot of beam 0 (the ancestry table the selection
        writes points every beam there) and the caller replicates the logits.  The cache is then addressed as
        N rows of W*Tmax positions."""
        m, d, H = self.model, self.d, self.H
        S = st["S"]
        R, W, Tmax, every = (st["N"], 1, st["W"] * st["Tmax"], st["W"]) if collapse else (st["R"], st["W"], st["Tmax"], 1)
        dh = d // H
        scale = 1.0 / math.sqrt(dh)
        l = lib()
        x = ops.embed_fwd(tok.view(R, 1), m.embedding.weight, st["pe"], t0=t).view(R, d)
        anc = ptr(st["anc"][st["cur"]])
        # the one-call step takes fp32 matrices only: bf16 storage keeps the op-by-op path, whose products cast operands
        if self.fused_step and ops.COMPUTE_DTYPE != ops.BF16S:
            dff = st["dff"]
            need = st["gemm_need"].get(R)
            if need is None:
                need = st["gemm_need"][R] = max(l.capdec_gemm_workspace_bytes(R, n, k)
                                                for n, k in ((d, d), (2 * d, d), (dff, d), (d, dff)))
            gws = _capi.workspace(max(need, 1), x.device, "gemm:" + ops._WS_TAG, zero=True)
            nb = l.capdec_decode_step_workspace_bytes(R, d, dff)
            ws = _capi.workspace(nb, x.device, "decode_step")
            check(l.capdec_decode_step(C.cast(st["layers_c"], C.c_void_p), self.L, ptr(x), anc, R, H, d, dff, Tmax, t, W, S,
                                       ops.COMPUTE_DTYPE, ptr(ws), ws.numel(), ptr(gws), gws.numel(), stream()),
                  "capdec_decode_step")
            y, _, _ = ops.layernorm_fwd(x, self.norm[0], self.norm[1], save_stats=False)
            return engine.head_fwd(self.head, y, self.dp, None)
        for li, P in enumerate(self.layers):
            V = st["views"][li]  # weight / cache slices made once per decode call (a slice costs ~1.5 us of host time)
            q = ops.linear_fwd(x, V["wq"], V["bq"])
            cache = st["cache"][li]  # [N*W, Tmax, 2d]
            ops.linear_fwd(x, V["wkv"], V["bkv"], out=cache[::every, t, :])  # scatter via the output row stride
            a = torch.empty((R, d), dtype=torch.float32, device=x.device)
            check(l.capdec_decode_attn(ptr(q), d, V["c2k"], V["c2v"], 2 * d, ptr(a), d, anc,
                                       R, H, dh, Tmax, t, W, S, scale, stream()), "capdec_decode_attn")
            s1 = ops.linear_fwd(a, P["self_attn.out_proj.weight"], P["self_attn.out_proj.bias"], addend=x)
            x1, _, _ = ops.layernorm_fwd(s1, P["norm1.weight"], P["norm1.bias"], save_stats=False)
            if S == 1:
                # one key (ref models.py:104): the context is that key's value whatever the query, so the whole
                # cross-attention branch out_proj(v) + bias is a per-image constant, computed once in _prepare;
                # (acc + bias) + x1 rounds exactly like the fused GEMM epilogue does
                u = st["cross_out"][li]
                s2 = (u if W == 1 else u.repeat_interleave(W, dim=0)) + x1
            else:
                q = ops.linear_fwd(x1, V["wcq"], V["bcq"])
                c = torch.empty((R, d), dtype=torch.float32, device=x.device)
                check(l.capdec_decode_attn(ptr(q), d, V["kvk"], V["kvv"], 2 * d, ptr(c), d, None, R, H, dh, Tmax, t,
                                           W, S, scale, stream()), "capdec_decode_attn")
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
        cross, cross_out = [], []
        with ops.compute_dtype(self.dtype):
            mem = engine.proj_fwd(self.proj, features.reshape(N * S, -1), self.dp, None)  # [N*S, d]
            for P in self.layers:
                wc, bc = P["multihead_attn.in_proj_weight"], P["multihead_attn.in_proj_bias"]
                if S == 1:  # only V is ever used; the branch's output is constant per image
                    v = ops.linear_fwd(mem, wc[2 * d:], bc[2 * d:])
                    cross.append(None)
                    cross_out.append(ops.linear_fwd(v, P["multihead_attn.out_proj.weight"],
                                                    P["multihead_attn.out_proj.bias"]))
                else:
                    cross.append(ops.linear_fwd(mem, wc[d:], bc[d:]))
                    cross_out.append(None)
        R = N * W
        cache = [torch.empty((R, Tmax, 2 * d), dtype=torch.float32, device=dev) for _ in self.layers]
        views = []
        for li, P in enumerate(self.layers):
            w_in, b_in = P["self_attn.in_proj_weight"], P["self_attn.in_proj_bias"]
            wc, bc = P["multihead_attn.in_proj_weight"], P["multihead_attn.in_proj_bias"]
            c2 = cache[li].view(-1, 2 * d)
            views.append(dict(wq=w_in[:d], bq=b_in[:d], wkv=w_in[d:], bkv=b_in[d:], wcq=wc[:d], bcq=bc[:d],
                              c2k=ptr(c2), c2v=ptr(c2[:, d:]), kvk=ptr(cross[li]),
                              kvv=None if cross[li] is None else ptr(cross[li][:, d:])))
        # the same pointers as one C array for capdec_decode_step (the tensors stay referenced by `st` / the model)
        layers_c = (_capi.DecodeLayer * len(self.layers))()
        for li, P in enumerate(self.layers):
            lc = layers_c[li]
            lc.w_in, lc.b_in = ptr(P["self_attn.in_proj_weight"]), ptr(P["self_attn.in_proj_bias"])
            lc.w_out, lc.b_out = ptr(P["self_attn.out_proj.weight"]), ptr(P["self_attn.out_proj.bias"])
            lc.ln1_w, lc.ln1_b = ptr(P["norm1.weight"]), ptr(P["norm1.bias"])
            lc.wc_in, lc.bc_in = ptr(P["multihead_attn.in_proj_weight"]), ptr(P["multihead_attn.in_proj_bias"])
            lc.wc_out, lc.bc_out = ptr(P["multihead_attn.out_proj.weight"]), ptr(P["multihead_attn.out_proj.bias"])
            lc.ln2_w, lc.ln2_b = ptr(P["norm2.weight"]), ptr(P["norm2.bias"])
            lc.w1, lc.b1, lc.w2, lc.b2 = ptr(P["linear1.weight"]), ptr(P["linear1.bias"]), ptr(P["linear2.weight"]), ptr(P["linear2.bias"])
            lc.ln3_w, lc.ln3_b = ptr(P["norm3.weight"]), ptr(P["norm3.bias"])
            lc.cache, lc.cross_kv, lc.cross_out = ptr(cache[li]), ptr(cross[li]), ptr(cross_out[li])
        dense = all(P[k].is_contiguous() for P in self.layers for k in engine.LAYER_KEYS)
        if self.fused_step and not dense:
            raise _capi.CapdecError("CaptionDecoder: capdec_decode_step needs contiguous layer parameters (fused_step=False "
                                    "takes strided ones)")
        st = dict(N=N, W=W, R=R, S=S, Tmax=Tmax, cross_kv=cross, cross_out=cross_out, cur=0, views=views,
                  layers_c=layers_c, dff=self.layers[0]["linear1.weight"].shape[0], gemm_need={},
                  pe=m.pos_encoder.table(),
                  cache=cache,
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
                if t == 0 and beam > 1:   # all beams of an image are the same hypothesis: one row per image
                    logits = self._step(tok[::beam].contiguous(), 0, st, collapse=True).repeat_interleave(beam, dim=0)
                else:
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

    @torch.no_grad()
    def sample(self, features, max_len=200, max_sentences=10, min_words_per_sentence=5, temperature=0.7, seed=0,
               check_every=16):
        """The reference's ``generate_caption`` (train.py:289-412) from projected-ViT ``features``:
        ids ``[B, L]`` int64, START first, 0-padded, every row closed with ('.' if a sentence is open) END.
        Same distribution as the reference (torch.multinomial is replaced by inverse-CDF sampling on a
        counter-based uniform keyed by (seed, step, row)); token-exact against
        ``oracle.torch_oracle.reference_sampler`` fed the same uniforms.  The host looks at the stop
        counters every ``check_every`` steps only."""
        st = self._prepare(features, 1, max_len)
        R, dev = st["R"], features.device
        tok = torch.full((R,), START, dtype=torch.int64, device=dev)
        seq = st["seq"][0]
        seq[:, 0] = START
        wsp = torch.zeros(R, dtype=torch.int32, device=dev)
        sg = torch.zeros(R, dtype=torch.int32, device=dev)
        cnt = torch.zeros((2, max_len), dtype=torch.int32, device=dev)
        l = lib()
        steps = max_len - 1
        for i in range(steps):
            with ops.compute_dtype(self.dtype):
                logits = self._step(tok, i, st)
            check(l.capdec_sample_select(ptr(logits), logits.stride(0), ptr(wsp), ptr(sg), ptr(seq), ptr(tok), ptr(cnt[0]),
                                         ptr(cnt[1]), R, self.V, max_len, i, max_len, max_sentences,
                                         min_words_per_sentence, float(temperature), int(seed) & 0xFFFFFFFFFFFFFFFF,
                                         stream()), "capdec_sample_select")
            if (i + 1) % check_every == 0 and bool((cnt[:, :i + 1] == R).any()):
                break
        stop = (cnt == R).any(0).nonzero()
        L = int(stop[0]) + 2 if stop.numel() else max_len  # START + (first stopping step + 1) tokens
        gen = seq[:, :L].to(torch.int64)
        # train.py:387-410: close open rows with ['.'] END, pad with 0 to the longest
        need = gen[:, -1] != 2
        add_p = need & (wsp > 0)
        out = torch.zeros((R, L + 2), dtype=torch.int64, device=dev)
        out[:, :L] = gen
        rows = torch.arange(R, device=dev)
        out[rows[add_p], L] = 4
        end_pos = L + add_p.to(torch.int64)
        out[rows[need], end_pos[need]] = 2
        width = L + int((need.to(torch.int64) * (1 + add_p.to(torch.int64))).max())
        return out[:, :width]


def generate_caption(model, image, device, vocab_idx2word=None, max_len=200, max_sentences=10,
                     min_words_per_sentence=5, seed=0, dtype="fp32"):
    """Drop-in for the reference's ``generate_caption`` (train.py:289; same signature plus ``seed`` /
    ``dtype``): ``model.vit(image)`` features -> KV-cached device sampler.  ``device`` and
    ``vocab_idx2word`` are accepted for signature compatibility (the reference ignores the latter too)."""
    model.eval()
    with torch.no_grad():
        features = model.vit(image)
    dec = getattr(model, "_capdec_decoder", None)
    if dec is None or dec.dtype != dtype:
        dec = CaptionDecoder(model, dtype=dtype)
        object.__setattr__(model, "_capdec_decoder", dec)
    return dec.sample(features, max_len=max_len, max_sentences=max_sentences,
                      min_words_per_sentence=min_words_per_sentence, seed=seed)
