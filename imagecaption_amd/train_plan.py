"""One C call per decoder layer and direction of a training step (capdec_train_layer_fwd / _bwd, include/capdec.h).

``engine.layer_fwd`` / ``layer_bwd`` issue a layer's ~12 + ~25 launches op by op: a Python -> ctypes round trip, a
torch allocation per result and an event pair per weight-gradient hand-over each, ~5 ms of host time per C2 step --
which is what the bf16-storage step (6 ms of GPU work) is bound by.  ``LayerPlans`` holds what the C calls need
instead: per layer a ``capdec_train_layer`` (parameter / gradient pointers into the trainer's flat blocks, the bf16
weight copies of ``ops.Bf16Copies``) and an arena of fixed layout for the layer's activations and backward buffers,
allocated once per (B, T, S).  The launches, their arguments and their order per stream are those of the op-by-op
plan: parameters after a step are bit-identical (tests/test_model_gpu.py).  Same reference lines as engine.py
(torch/nn/modules/transformer.py:1143-1199 as built at ref models.py:53-65; ref train.py:24-61).
"""
import ctypes as C
import os

import torch

from . import _capi, ops
from ._capi import check, lib, ptr

_KEYS = dict(w_in="self_attn.in_proj_weight", b_in="self_attn.in_proj_bias", w_out="self_attn.out_proj.weight",
             b_out="self_attn.out_proj.bias", ln1_w="norm1.weight", ln1_b="norm1.bias", wc_in="multihead_attn.in_proj_weight",
             bc_in="multihead_attn.in_proj_bias", wc_out="multihead_attn.out_proj.weight", bc_out="multihead_attn.out_proj.bias",
             ln2_w="norm2.weight", ln2_b="norm2.bias", w1="linear1.weight", b1="linear1.bias", w2="linear2.weight",
             b2="linear2.bias", ln3_w="norm3.weight", ln3_b="norm3.bias")


def covers(B, T, S, H, d, dff, dtype, n_layers, mem_rows):
    """Whether the one-call plans cover this step: a memory of several positions whose K | V projections the stack
    batches (engine.kv_batched), and -- bf16 storage -- widths the all-bf16 kernels take without padding tails."""
    from . import engine
    if S <= 1 or not engine.kv_batched(n_layers, mem_rows) or d % 4 or dff % 4 or d % H:
        return False
    if dtype == _capi.BF16S and (d % 64 or dff % 64 or not ops.attn16_ok(H, d)):
        return False
    return B * T * (3 * d + 64) < 2 ** 31 and B * T * (dff + 64) < 2 ** 31


class LayerPlans:
    def __init__(self, P_layers, G_layers, B, T, S, H, causal, drop_p, dtype, device, copies=None):
        """``copies``: CAPDEC_BF16S -- {(data_ptr, rows, cols, row stride): (source, bf16 copy)} holding the weight
        operands' persistent copies (``ops.Bf16Copies.memo()``); a missing one raises KeyError."""
        d = P_layers[0]["linear1.weight"].shape[1]
        dff = P_layers[0]["linear1.weight"].shape[0]
        self.B, self.T, self.S, self.H, self.d, self.dff, self.dtype, self.device = B, T, S, H, d, dff, dtype, device
        g = self.geom = _capi.TrainGeom()
        g.B, g.T, g.S, g.H, g.d, g.dff, g.causal, g.dtype, g.drop_p = B, T, S, H, d, dff, int(causal), dtype, float(drop_p)
        # (the op-by-op plan groups weight gradients only where it queues them: with a lane)
        grouped = bool(ops.GROUP_WGRADS and ops.SIDE is not None)
        g.group_wgrads, g.group_bm, g.group_bn = int(grouped), int(ops.GROUP_TILE[0]), int(ops.GROUP_TILE[1])
        g.wgrad_flush_mask = int(os.environ.get("CAPDEC_WGRAD_FLUSH_MASK", "0"), 0)  # experiments; 0 = the library's default
        self.key = (B, T, S, H, bool(causal), float(drop_p), dtype, grouped, tuple(ops.GROUP_TILE))
        need = lib().capdec_train_layer_arena_bytes(C.byref(g))
        if need == 0:
            raise _capi.CapdecError("LayerPlans: geometry not covered by capdec_train_layer_fwd / _bwd")
        self.layers = (_capi.TrainLayer * len(P_layers))()
        self.arenas, self.hold = [], []
        for L, P, G in zip(self.layers, P_layers, G_layers):
            for f, k in _KEYS.items():
                p_, g_ = P[k], G[k]
                if not (p_.is_contiguous() and g_.is_contiguous() and p_.dtype == g_.dtype == torch.float32):
                    raise _capi.CapdecError(f"LayerPlans: {k} and its gradient must be dense fp32 tensors")
                setattr(L, f, ptr(p_))
                setattr(L, "g_" + f, ptr(g_))
            if dtype == _capi.BF16S:
                wq = P["multihead_attn.in_proj_weight"][:d]
                for f, w in (("w_in16", P["self_attn.in_proj_weight"]), ("w_out16", P["self_attn.out_proj.weight"]), ("wcq16", wq),
                             ("wc_out16", P["multihead_attn.out_proj.weight"]), ("w1_16", P["linear1.weight"]),
                             ("w2_16", P["linear2.weight"])):
                    src, c16 = copies[(w.data_ptr(), w.shape[0], w.shape[1], w.stride(0))]
                    setattr(L, f, ptr(c16))
                    setattr(L, "ld_" + f, c16.stride(0))
                    self.hold.append(c16)
            # (zeros: the padding columns behind bf16 rows that only a product's epilogue writes are never read at the
            # widths covers() admits, but they should not hold NaN patterns either)
            arena = torch.zeros(need, dtype=torch.uint8, device=device)
            if arena.data_ptr() % 256:
                raise _capi.CapdecError("LayerPlans: the allocator returned an arena that is not 256-byte aligned")
            L.arena, L.arena_bytes = ptr(arena), need
            self.arenas.append(arena)
        self._views = {}
        self._ws_need = None
        self.x_in = [None] * len(P_layers)  # (x, x16) of each layer's forward call, for its backward

    def _slot(self, l, slot):
        """The tensor of ``slot`` inside layer l's arena as a torch view."""
        v = self._views.get((l, slot))
        if v is None:
            off, nbytes, ld, b16 = C.c_size_t(0), C.c_size_t(0), C.c_int(0), C.c_int(0)
            check(lib().capdec_train_layer_slot(C.byref(self.geom), slot, C.byref(off), C.byref(nbytes), C.byref(ld), C.byref(b16)),
                  "capdec_train_layer_slot")
            raw = self.arenas[l][off.value:off.value + nbytes.value]
            M = self.B * self.T
            v = raw.view(torch.bfloat16 if b16.value else torch.float32).view(M, ld.value)[:, :self.d]
            if b16.value:
                v = ops._zero_padded(v)  # written by capdec_layernorm_fwd_y16, which zeroes the padding
            self._views[(l, slot)] = v
        return v

    def _workspaces(self):
        """(main, lane) capdec_gemm workspaces covering every product of a layer under the plans pinned right now."""
        tuned = len(ops._TUNED)
        if self._ws_need is None or self._ws_need[0] != tuned:
            M, Ms, d, dff = self.B * self.T, self.B * self.S, self.d, self.dff
            shapes = [(M, 3 * d, d), (M, d, d), (M, dff, d), (M, d, dff), (M, d, 3 * d), (d, dff, M), (dff, d, M), (d, d, M),
                      (2 * d, d, Ms), (3 * d, d, M)]
            self._ws_need = (tuned, max(lib().capdec_gemm_workspace_bytes(*s) for s in shapes))
        need = self._ws_need[1]
        main = _capi.workspace(need, self.device, "gemm:main", zero=True)
        lane = _capi.workspace(need, self.device, "gemm:side", zero=True) if ops.SIDE is not None else main
        return main, lane

    def forward(self, x, kv_all, dropout_key):
        """x [B*T, d] fp32 -> the last layer's output (a view of its arena).  ``kv_all`` [B*S, L*2d]: the batched K | V
        projection of the memory (bf16 in bf16-storage mode)."""
        l_ = lib()
        s16 = self.dtype == _capi.BF16S
        d = self.d
        x16 = ops._bf16_operand(x, "LayerPlans.x") if s16 else None
        main, _ = self._workspaces()
        st = _capi.stream()
        for l in range(len(self.layers)):
            self.x_in[l] = (x, x16)
            kv = kv_all[:, l * 2 * d:(l + 1) * 2 * d]
            check(l_.capdec_train_layer_fwd(C.byref(self.layers[l]), C.byref(self.geom), l, dropout_key, ptr(x), ptr(x16),
                                            x16.stride(0) if s16 else 0, ptr(kv), kv.stride(0), ptr(main), main.numel(), st),
                  "capdec_train_layer_fwd")
            x = self._slot(l, _capi.TL_X3)
            x16 = self._slot(l, _capi.TL_X3_16) if s16 else None
        return x

    def backward(self, dy, mem, kv_all, dkv_all, dropout_key, after_layer=None):
        """dy [B*T, d] fp32 at the last layer's output -> dx at the first layer's input (a view of its arena); dK | dV
        of every layer land in ``dkv_all`` [B*S, L*2d]; weight gradients go to the gradient tensors the plan was built
        over, on the lane (``ops.SIDE``) when there is one."""
        l_ = lib()
        s16 = self.dtype == _capi.BF16S
        d = self.d
        mem16 = ops._bf16_operand(mem, "LayerPlans.mem") if s16 else None
        main, lane_ws = self._workspaces()
        st = _capi.stream()
        side = ops.SIDE
        lane = side.stream.cuda_stream if side is not None else st
        for l in range(len(self.layers) - 1, -1, -1):
            x, x16 = self.x_in[l]
            kv = kv_all[:, l * 2 * d:(l + 1) * 2 * d]
            dkv = dkv_all[:, l * 2 * d:(l + 1) * 2 * d]
            check(l_.capdec_train_layer_bwd(C.byref(self.layers[l]), C.byref(self.geom), l, dropout_key, ptr(dy), ptr(x), ptr(x16),
                                            x16.stride(0) if s16 else 0, ptr(mem), ptr(mem16), mem16.stride(0) if s16 else 0,
                                            ptr(kv), kv.stride(0), ptr(dkv), dkv.stride(0), ptr(main), main.numel(), ptr(lane_ws),
                                            lane_ws.numel(), st, lane), "capdec_train_layer_bwd")
            if side is not None:
                side.pending = True
                side.keep.extend((x, dy, mem, kv_all, dkv_all) + ((x16, mem16) if s16 else ()))
            dy = self._slot(l, _capi.TL_DX)
            if after_layer is not None:
                after_layer(l)
        self.x_in = [None] * len(self.layers)
        return dy
