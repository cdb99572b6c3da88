"""ctypes binding of libcapdec.so (include/capdec.h).  Fails loudly when the library is missing.

There is no CPU fallback anywhere in this package: every op goes through this module, and a
missing / unloadable ``libcapdec.so`` raises at first use.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CAPDEC_LIBRARY") or os.path.join(_HERE, "libcapdec.so")  # override: kernel experiments

F32, BF16, F32X3, BF16S = 0, 1, 2, 3
ACT_NONE, ACT_GELU, ACT_GELU_BWD = 0, 1, 2

_vp, _i, _f, _u64, _sz = C.c_void_p, C.c_int, C.c_float, C.c_uint64, C.c_size_t


class Epilogue(C.Structure):
    _fields_ = [("bias", _vp), ("addend", _vp), ("pre", _vp), ("aux", _vp), ("colsum", _vp),
                ("ld_addend", _i), ("act", _i), ("drop_p", _f), ("drop_seed", _u64),
                ("ce_partials", _vp), ("ce_lse", _vp), ("ce_targets", _vp), ("ce_eps", _f), ("ce_grad_scale", _f),
                ("c16", _vp), ("ldc16", _i), ("bf16_unpadded", _i)]


class CastDesc(C.Structure):
    _fields_ = [("src", _vp), ("dst", _vp), ("lds", _i), ("ldd", _i), ("R", _i), ("C", _i),
                ("vec", _i), ("first_block", _i), ("blocks", _i), ("reserved", _i)]


CAST_MULTI_MAX = 256
GEMM_GROUP_MAX = 8


class GemmDesc(C.Structure):
    """capdec_gemm_desc of include/capdec.h."""
    _fields_ = [("transA", _i), ("transB", _i), ("M", _i), ("N", _i), ("K", _i), ("A", _vp), ("lda", _i), ("B", _vp),
                ("ldb", _i), ("C", _vp), ("ldc", _i), ("ep", C.POINTER(Epilogue))]


class TrainGeom(C.Structure):
    """capdec_train_geom of include/capdec.h."""
    _fields_ = [(n, _i) for n in ("B", "T", "S", "H", "d", "dff", "causal", "dtype")] + [("drop_p", _f)] + \
               [(n, _i) for n in ("group_wgrads", "group_bm", "group_bn", "wgrad_flush_mask")]


TRAIN_LAYER_PARAMS = ("w_in", "b_in", "w_out", "b_out", "ln1_w", "ln1_b", "wc_in", "bc_in", "wc_out", "bc_out", "ln2_w", "ln2_b",
                      "w1", "b1", "w2", "b2", "ln3_w", "ln3_b")
TRAIN_LAYER_COPIES = ("w_in16", "w_out16", "wcq16", "wc_out16", "w1_16", "w2_16")


class TrainLayer(C.Structure):
    """capdec_train_layer of include/capdec.h."""
    _fields_ = [(n, _vp) for n in TRAIN_LAYER_PARAMS] + [("g_" + n, _vp) for n in TRAIN_LAYER_PARAMS] + \
               [(n, _vp) for n in TRAIN_LAYER_COPIES] + [("ld_" + n, _i) for n in TRAIN_LAYER_COPIES] + \
               [("arena", _vp), ("arena_bytes", _sz)]


TL_X3, TL_X3_16, TL_DX, TL_SLOTS = 19, 20, 37, 41

# name -> (restype, argtypes); mirrors include/capdec.h one to one
SIGNATURES = {
    "capdec_train_layer_arena_bytes": (_sz, [C.POINTER(TrainGeom)]),
    "capdec_train_layer_slot": (_i, [C.POINTER(TrainGeom), _i, C.POINTER(_sz), C.POINTER(_sz), C.POINTER(_i), C.POINTER(_i)]),
    "capdec_train_layer_fwd": (_i, [C.POINTER(TrainLayer), C.POINTER(TrainGeom), _i, _u64, _vp, _vp, _i, _vp, _i, _vp, _sz, _vp]),
    "capdec_train_layer_bwd": (_i, [C.POINTER(TrainLayer), C.POINTER(TrainGeom), _i, _u64, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _i,
                                    _vp, _i, _vp, _sz, _vp, _sz, _vp, _vp]),
    "capdec_abi_version": (_i, []),
    "capdec_last_error": (C.c_char_p, []),
    "capdec_gemm_workspace_bytes": (_sz, [_i, _i, _i]),
    "capdec_gemm_tune": (_i, [_i, _i, _i, _i]),
    "capdec_gemm_plan_set": (_i, [_i, _i, _i, _i, _i, _i, _i, _i, _i, _i]),
    "capdec_gemm_plan_clear": (None, []),
    "capdec_gemm": (_i, [_i, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _i, C.POINTER(Epilogue), _i, _vp, _sz, _vp]),
    "capdec_gemm_grouped": (_i, [C.POINTER(GemmDesc), _i, _i, _i, _vp]),
    "capdec_layernorm_bwd_workspace_bytes": (_sz, [_i, _i]),
    "capdec_layernorm2_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _f, _vp]),
    "capdec_layernorm_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _i, _f, _u64, _vp]),
    "capdec_layernorm_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _u64, _f, _u64,
                                  _vp, _sz, _vp]),
    "capdec_layernorm_bwd_dx": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _f, _u64, _f, _u64, _vp]),
    "capdec_layernorm_bwd_dparams": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "capdec_layernorm_fwd_y16": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _f, _i, _f, _u64, _vp]),
    "capdec_layernorm_bwd_dx16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _sz, _i, _i, _i, _f, _u64, _f, _u64,
                                       _vp]),
    "capdec_layernorm_bwd_noalloc_max_bytes": (_sz, []),
    "capdec_layernorm_bwd_noalloc": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _u64, _f,
                                          _u64, _vp]),
    "capdec_embed_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _u64, _vp]),
    "capdec_embed_bad_ids": (_i, [C.POINTER(C.c_uint), _i]),
    "capdec_embed_bwd_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "capdec_embed_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _f, _u64, _vp, _sz, _vp]),
    "capdec_embed_bwd_plan": (_i, [_vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    "capdec_embed_bwd_apply": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _f, _u64, _vp, _sz, _vp]),
    "capdec_attn_tune": (_i, [_i]),
    "capdec_attn_fwd": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _f, _f, _u64, _vp]),
    "capdec_attn_bwd": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _i, _vp, _i,
                             _i, _i, _i, _i, _i, _i, _f, _f, _u64, _vp]),
    "capdec_attn_fwd_kv16": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _f, _f, _u64, _vp]),
    "capdec_attn_bwd_kv16": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _i, _vp, _i,
                                  _i, _i, _i, _i, _i, _i, _f, _f, _u64, _vp]),
    "capdec_attn_bwd_dkv16": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _i, _vp, _i,
                                   _i, _i, _i, _i, _i, _i, _f, _f, _u64, _vp]),
    "capdec_attn_fwd_bf16": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _f, _f, _u64, _vp]),
    "capdec_attn_bwd_bf16": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _i, _vp, _i,
                                  _i, _i, _i, _i, _i, _i, _f, _f, _u64, _vp]),
    "capdec_softmax_ce": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _f, _f, _vp]),
    "capdec_gemm_ce_tiles": (_i, [_i, _i, _i, _i]),
    "capdec_ce_finalize": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _f, _vp]),
    "capdec_ce_dlogits": (_i, [_vp, _i, _vp, _vp, _i, _i, _f, _f, _vp]),
    "capdec_cast_bf16": (_i, [_vp, _i, _vp, _i, _i, _i, _vp]),
    "capdec_cast_bf16_multi_plan": (_i, [C.POINTER(CastDesc), _i, _vp, C.POINTER(_i)]),
    "capdec_cast_bf16_multi": (_i, [_vp, _i, _i, _vp]),
    "capdec_copy_f32_multi_plan": (_i, [C.POINTER(CastDesc), _i, _vp, C.POINTER(_i)]),
    "capdec_copy_f32_multi": (_i, [_vp, _i, _i, _vp]),
    "capdec_shift_captions": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp]),
    "capdec_ce_dlogits_bf16": (_i, [_vp, _i, _vp, _vp, _i, _i, _f, _f, _vp, _i, _vp]),
    "capdec_ce_dlogits_bf16_inplace": (_i, [_vp, _i, _vp, _vp, _i, _i, _f, _f, _vp]),
    "capdec_reduce_workspace_bytes": (_sz, []),
    "capdec_sum": (_i, [_vp, _sz, _f, _vp, _vp, _sz, _vp]),
    "capdec_sumsq": (_i, [_vp, _sz, _vp, _vp, _sz, _vp]),
    "capdec_clip_scale": (_i, [_vp, _sz, _vp, _f, _vp]),
    "capdec_adamw": (_i, [_vp, _vp, _vp, _vp, _sz, _f, _f, _f, _f, _f, _i, _vp, _f, _vp]),
    "capdec_decode_attn": (_i, [_vp, _i, _vp, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _vp]),
    "capdec_decode_step_workspace_bytes": (_sz, [_i, _i, _i]),
    "capdec_decode_step": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp, _sz, _vp]),
    "capdec_greedy_select": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "capdec_beam_select": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "capdec_sample_select": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _u64, _vp]),
    "capdec_probe_mfma": (_i, [_i, _i, _i, _vp, C.POINTER(C.c_double), _vp]),
    "capdec_probe_copy": (_i, [_vp, _vp, _sz, _vp]),
    "capdec_cross1_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _u64, _f, _u64, _vp]),
    "capdec_cross1_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _f, _u64, _vp]),
    "capdec_cross1_spread": (_i, [_vp, _i, _vp, _i, _i, _i, _vp]),
    "capdec_cross1_gather": (_i, [_vp, _vp, _vp, _i, _vp, _i, _vp, _i, _i, _i, _vp]),
    "capdec_pe_add": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _u64, _vp]),
    "capdec_act_dropout_bwd": (_i, [_vp, _vp, _vp, _sz, _f, _u64, _vp]),
}

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m imagecaption_amd.build` "
                "(hipcc --offload-arch=gfx950).  There is no CPU / PyTorch fallback for the capdec ops.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


class DecodeLayer(C.Structure):
    """capdec_decode_layer of include/capdec.h."""
    _fields_ = [(n, C.c_void_p) for n in ("w_in", "b_in", "w_out", "b_out", "ln1_w", "ln1_b", "wc_in", "bc_in", "wc_out",
                                          "bc_out", "ln2_w", "ln2_b", "w1", "b1", "w2", "b2", "ln3_w", "ln3_b", "cache",
                                          "cross_kv", "cross_out")]


class CapdecError(RuntimeError):
    pass


def check(rc, what=""):
    if rc != 0:
        msg = lib().capdec_last_error().decode("utf-8", "replace")
        raise CapdecError(f"{what or 'capdec'} failed (code {rc}): {msg}")


def ptr(t):
    return None if t is None else t.data_ptr()


try:  # the raw handle of torch's current stream without building a torch.cuda.Stream object (~8 us -> < 1 us per op)
    _raw_stream, _cur_device = torch._C._cuda_getCurrentRawStream, torch._C._cuda_getDevice
except AttributeError:  # CPU-only torch builds / future renames: the public, slower path
    _raw_stream = _cur_device = None


def stream():
    if _raw_stream is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


# ---- per-device scratch workspace (borrowed by the C ABI; grown on demand, never inside capture) ----
_ws = {}


def workspace(nbytes, device, tag="main", zero=False):
    """``zero``: the buffer is zero-filled when (re)allocated -- capdec_gemm's split-K ticket words need that once
    and keep themselves zero afterwards, so that buffer (tags "gemm:*") is never lent to any other op."""
    key = (device.type, device.index, tag)
    w = _ws.get(key)
    if w is None or w.numel() < nbytes:
        size = max(int(nbytes), 64 << 20)
        w = (torch.zeros if zero else torch.empty)(size, dtype=torch.uint8, device=device)
        _ws[key] = w
    return w
