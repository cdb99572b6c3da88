"""Raw (non-autograd) tensor-level wrappers over the C ABI.  Plumbing only: shapes, strides, pointers.

Every function enqueues hand-written gfx950 kernels on torch's current HIP stream.  2-D operands
may be row-strided views (``stride(1) == 1``), e.g. the q / k / v slices of a packed projection.
"""
import ctypes as C
import math
import os
import sys

import torch

from . import _capi
from ._capi import ACT_GELU_BWD, ACT_NONE, F32, check, lib, ptr, stream

__all__ = ["compute_dtype", "gemm", "linear_fwd", "linear_dgrad", "linear_wgrad", "layernorm_fwd", "layernorm_bwd", "embed_fwd",
           "embed_bwd", "attn_fwd", "attn_bwd", "softmax_ce", "ce_tiles", "ce_finalize", "ce_dlogits", "ce_dlogits_bf16", "cast_bf16", "reduce_sum", "sumsq", "adamw", "clip_scale", "pe_add",
           "act_dropout_bwd"]


# Arithmetic type of the matrix products.  F32: exact fp32 MFMA (reference precision, the default).
# BF16: operands rounded to bf16 on load, bf16 MFMA, fp32 accumulation / epilogue / storage
# ("autocast"-style mixed precision; tolerances are stated separately in the tests).
# BF16S: the same arithmetic on bf16 COPIES of the operands (capdec_cast_bf16): the products read half the bytes and
# convert nothing; results, epilogues, norms, attention, loss and optimizer state stay fp32.
COMPUTE_DTYPE = F32
BF16S = _capi.BF16S

# bf16 copies made during one training step: (data_ptr, rows, cols, row stride) -> (source, copy).  None = off: every
# product casts its operands afresh.  Only a caller that knows when tensors change may switch it on (CaptionTrainer:
# a dict per step, dropped at the step's end -- weights change in the optimizer, buffers are reused by the next step);
# holding the source keeps its memory from being handed to another tensor while the entry lives.
BF16_MEMO = None


def _bf16_pitch(cols):
    """Row stride (elements) of a bf16 copy: the row rounded up to 64 -- the zero padding behind the row is what lets
    the LDS-DMA GEMM kernel (which cannot mask a load) run the reduction tail; a pitch that is a multiple of 1 KiB
    would put every row of a GEMM tile on the same few L2 / HBM channels, so such pitches get one more 128-byte line."""
    ldd = (cols + 63) // 64 * 64
    if ldd % 512 == 0:
        ldd += 64
    return ldd


def cast_bf16(t):
    """bf16 copy of an fp32 matrix (row-strided views welcome): [R, C] view of a [R, round_up(C, 8)] buffer whose
    padding columns are zero -- the form capdec_gemm's CAPDEC_BF16S operands take, in either orientation."""
    _mat(t, "cast_bf16.src")
    R, Cc = t.shape
    ldd = _bf16_pitch(Cc)
    dst = torch.empty((R, ldd), dtype=torch.bfloat16, device=t.device)
    check(lib().capdec_cast_bf16(ptr(t), t.stride(0), ptr(dst), ldd, R, Cc, stream()), "capdec_cast_bf16")
    return _zero_padded(dst[:, :Cc])


def _zero_padded(view):
    """Mark a bf16 [R, C] view whose buffer holds ZEROS in the columns C .. pitch-1 behind every row (the cast kernels
    write them).  Only such operands may run a reduction over C % 64 != 0 on the LDS-DMA products, which cannot mask a
    load; any other bf16 view (a column slice of a wider matrix, an uninitialised tail: the mark does not survive
    slicing) is flagged to capdec_gemm (capdec_epilogue.bf16_unpadded) and takes the register-staged kernel."""
    view._capdec_zero_padded = True
    return view


def _unpadded_bits(a, b, trans_a, trans_b, K):
    if K % 64 == 0:
        return 0
    bits = 0
    if not trans_a and not getattr(a, "_capdec_zero_padded", False):
        bits |= 1
    if trans_b and not getattr(b, "_capdec_zero_padded", False):
        bits |= 2
    return bits


def _bf16_operand(t, name):
    if t.dtype == torch.bfloat16:
        if not t.is_cuda or t.dim() != 2 or t.stride(1) != 1 or t.stride(0) % 8 or t.stride(0) < (t.shape[1] + 7) // 8 * 8 \
                or t.data_ptr() % 16:
            raise _capi.CapdecError(f"{name}: a bf16 operand must be a 16-byte aligned 2-D view with unit column stride and a "
                                    f"row stride that is a multiple of 8 covering the row rounded up to 8 (ops.cast_bf16 makes "
                                    f"one), got {tuple(t.shape)} / {t.stride()}")
        return t
    if BF16_MEMO is None:
        return cast_bf16(t)
    key = (t.data_ptr(), t.shape[0], t.shape[1], t.stride(0))
    hit = BF16_MEMO.get(key)
    if hit is None:
        hit = BF16_MEMO[key] = (t, cast_bf16(t))
    return hit[1]


class Bf16Copies:
    """bf16 copies of a fixed set of fp32 matrices (a trainer's weight operands: same addresses every step), all
    refreshed by ONE launch (capdec_cast_bf16_multi).  ``sources`` are 2-D fp32 views; ``copies`` may bring the
    destination views to adopt (what ``cast_bf16`` returned for them), else they are allocated here."""

    def __init__(self, sources, copies=None):
        import ctypes as C
        if not 0 < len(sources) <= _capi.CAST_MULTI_MAX:
            raise _capi.CapdecError(f"Bf16Copies: 1 .. {_capi.CAST_MULTI_MAX} matrices per table, got {len(sources)}")
        self.sources = list(sources)
        for t in self.sources:
            _mat(t, "Bf16Copies.src")
        if copies is None:
            # (capdec_cast_bf16_multi writes the zero padding behind every row, like capdec_cast_bf16)
            copies = [_zero_padded(torch.empty((t.shape[0], _bf16_pitch(t.shape[1])), dtype=torch.bfloat16,
                                               device=t.device)[:, :t.shape[1]]) for t in self.sources]
        self.copies = list(copies)
        descs = (_capi.CastDesc * len(self.sources))()
        for d, t, c in zip(descs, self.sources, self.copies):
            if c.dtype != torch.bfloat16 or tuple(c.shape) != tuple(t.shape) or c.stride(1) != 1 or c.device != t.device:
                raise _capi.CapdecError(f"Bf16Copies: copy {tuple(c.shape)} / {c.dtype} does not fit source {tuple(t.shape)}")
            d.src, d.dst, d.lds, d.ldd, d.R, d.C = ptr(t), ptr(c), t.stride(0), c.stride(0), t.shape[0], t.shape[1]
        self.table = torch.empty(C.sizeof(descs), dtype=torch.uint8, device=self.sources[0].device)
        total = C.c_int(0)
        check(lib().capdec_cast_bf16_multi_plan(descs, len(self.sources), ptr(self.table), C.byref(total)),
              "capdec_cast_bf16_multi_plan")
        self.total_blocks = total.value

    def refresh(self):
        check(lib().capdec_cast_bf16_multi(ptr(self.table), len(self.sources), self.total_blocks, stream()),
              "capdec_cast_bf16_multi")

    def memo(self):
        """The BF16_MEMO entries of these matrices (valid until a source changes: call refresh() first)."""
        return {(t.data_ptr(), t.shape[0], t.shape[1], t.stride(0)): (t, c) for t, c in zip(self.sources, self.copies)}


class StackedRows:
    """One fp32 block [sum R_i, C] holding copies of matrices that live apart (row-strided fp32 views of equal width; a
    [n] vector counts as [n, 1]), refreshed by ONE launch (capdec_copy_f32_multi): the trainer's stacked cross-attention
    K | V projection weights and biases of all decoder layers, in place of a torch.cat per step."""

    def __init__(self, sources):
        if not 0 < len(sources) <= _capi.CAST_MULTI_MAX:
            raise _capi.CapdecError(f"StackedRows: 1 .. {_capi.CAST_MULTI_MAX} matrices, got {len(sources)}")
        vec = sources[0].dim() == 1
        self.sources = [(_req(t, "StackedRows.src").unsqueeze(1) if vec else _mat(t, "StackedRows.src")) for t in sources]
        cols = self.sources[0].shape[1]
        if any(t.shape[1] != cols for t in self.sources):
            raise _capi.CapdecError("StackedRows: matrices of different widths")
        rows = sum(t.shape[0] for t in self.sources)
        dev = self.sources[0].device
        self.block = torch.empty((rows, cols), dtype=torch.float32, device=dev)
        descs = (_capi.CastDesc * len(self.sources))()
        r0 = 0
        for d, t in zip(descs, self.sources):
            dst = self.block[r0:r0 + t.shape[0]]
            d.src, d.dst, d.lds, d.ldd, d.R, d.C = ptr(t), ptr(dst), t.stride(0), cols, t.shape[0], cols
            r0 += t.shape[0]
        self.table = torch.empty(C.sizeof(descs), dtype=torch.uint8, device=dev)
        total = C.c_int(0)
        check(lib().capdec_copy_f32_multi_plan(descs, len(self.sources), ptr(self.table), C.byref(total)),
              "capdec_copy_f32_multi_plan")
        self.total_blocks = total.value
        self.view = self.block.view(-1) if vec else self.block

    def refresh(self):
        check(lib().capdec_copy_f32_multi(ptr(self.table), len(self.sources), self.total_blocks, stream()),
              "capdec_copy_f32_multi")
        return self.view


def shift_captions(captions):
    """(captions[:, :-1], captions[:, 1:]) as dense int64 blocks, one launch (ref train.py:29-30)."""
    _req(captions, "shift_captions.captions", torch.int64)
    if captions.dim() != 2 or captions.stride(1) != 1 or captions.shape[1] < 2:
        raise _capi.CapdecError(f"shift_captions: need a [B, T + 1] int64 matrix with unit column stride, got {tuple(captions.shape)}")
    B, T1 = captions.shape
    out = torch.empty((2, B, T1 - 1), dtype=torch.int64, device=captions.device)
    check(lib().capdec_shift_captions(ptr(captions), captions.stride(0), B, T1, ptr(out[0]), ptr(out[1]), stream()),
          "capdec_shift_captions")
    return out[0], out[1]


class compute_dtype:
    """with ops.compute_dtype("bf16"): ...  -- scoped override of COMPUTE_DTYPE."""

    def __init__(self, name):
        self.value = {"f32": F32, "fp32": F32, "bf16": _capi.BF16, "f32x3": _capi.F32X3, "fp32x3": _capi.F32X3,
                      "bf16s": _capi.BF16S}[str(name).lower()]

    def __enter__(self):
        global COMPUTE_DTYPE
        self.prev, COMPUTE_DTYPE = COMPUTE_DTYPE, self.value

    def __exit__(self, *a):
        global COMPUTE_DTYPE
        COMPUTE_DTYPE = self.prev


# ---- the weight-gradient lane -----------------------------------------------------------------------
# HIP multiplexes all streams of one priority over 4 hardware queues by reference count, so a second stream may
# land on the SAME queue as the caller's stream and then runs strictly in order with it (observed once RCCL had
# created its streams: sum of kernel durations = GPU busy time, the overlap was gone).  Which queue a stream gets
# cannot be asked, but it can be measured: the lane is chosen at construction by racing a short MFMA probe on the
# caller's stream against one on each candidate and keeping the first candidate that really runs beside it.
# (High-priority streams would separate the queues by construction, but on this ROCm stack they deadlock -- the
# 4th one created in a process, or any of them after a hipGraph launch: tools/hp_stream_probe.py.)
_LANES = {}
LANE_INFO = {}  # the same key -> (candidates tried, whether the kept one was measured to run beside the others)


def _runs_beside(base, cand, device):
    with torch.cuda.stream(base):
        return _runs_beside_current(cand, device)


def _runs_beside_current(cand, device):
    import ctypes as C
    import time
    l = lib()
    out = torch.zeros(16, dtype=torch.float32, device=device)

    def race(both):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        check(l.capdec_probe_mfma(F32, 8, 6000, ptr(out), C.POINTER(C.c_double)(), stream()), "capdec_probe_mfma")
        if both:
            with torch.cuda.stream(cand):
                check(l.capdec_probe_mfma(F32, 8, 6000, ptr(out), C.POINTER(C.c_double)(), stream()), "capdec_probe_mfma")
        torch.cuda.synchronize(device)
        return time.perf_counter() - t0

    race(True)  # warm-up (module load, clocks)
    single = min(race(False) for _ in range(2))
    pair = min(race(True) for _ in range(2))
    return pair < 1.5 * single


def lane_stream(device, slot=0):
    """A normal-priority stream verified to execute concurrently with the current stream and with the lanes of
    lower slot numbers (cached per device, current stream and slot).  Slot 0: weight gradients; slot 1: the DP
    reducer's collectives."""
    device = torch.device(device)
    cur = torch.cuda.current_stream(device)
    index = device.index if device.index is not None else torch.cuda.current_device()
    key = (index, cur.cuda_stream, slot)
    if key not in _LANES:
        others = [lane_stream(device, s) for s in range(slot)]
        cand, accepted = None, False
        for attempt in range(8):
            cand = torch.cuda.Stream(device=device)
            if _runs_beside_current(cand, device) and all(_runs_beside(o, cand, device) for o in others):
                accepted = True
                break
        LANE_INFO[key] = (attempt + 1, accepted)
        if os.environ.get("CAPDEC_DEBUG_LANE"):
            print(f"[capdec] lane stream (slot {slot}): candidate #{attempt} {'runs beside' if accepted else 'last tried for'} "
                  f"stream {cur.cuda_stream:#x}", file=sys.stderr, flush=True)
        _LANES[key] = cand  # the last candidate if none qualified: correct, merely without overlap
    return _LANES[key]


class SideStream:
    """The weight-gradient lane of a training step.  Weight-gradient GEMMs depend on nothing later in
    backward, so they fill the matrix pipes while the critical path runs its latency-bound kernels
    (LayerNorm / attention backward, small input-gradient GEMMs with ragged tile counts).  The lane has
    its own split-K workspace; outputs are always allocated by the caller on the critical-path stream,
    so the caching allocator never sees a cross-stream free."""

    def __init__(self, device):
        self.stream = lane_stream(device)
        self.streams = (self.stream,)
        self.keep = []       # tensors the lane still reads; released at join()
        self.pending = False
        # the hand-over costs host time on every weight gradient (~70 per step, and the bf16-storage step is bound by
        # its host side): events are reused in rotation (a wait refers to the record that preceded it, so re-recording
        # an event later is harmless) and the lane is entered by setting torch's current stream directly
        self._events = [torch.cuda.Event() for _ in range(32)]
        self._next = 0
        self._index = self.stream.device_index

    def _current(self):
        try:  # torch.cuda.current_stream() minus its device-index resolution (most of its cost)
            sid, idx, typ = torch._C._cuda_getCurrentStream(self._index)
            return torch.cuda.Stream(stream_id=sid, device_index=idx, device_type=typ)
        except (AttributeError, TypeError, ValueError):
            return torch.cuda.current_stream(self._index)

    def run(self, fn, *tensors):
        """Enqueue fn() on the lane, ordered after everything enqueued so far on the current stream."""
        global _WS_TAG
        cur = self._current()
        ev = self._events[self._next]
        self._next = (self._next + 1) % len(self._events)
        ev.record(cur)
        self.stream.wait_event(ev)
        prev, _WS_TAG = _WS_TAG, "side"
        torch.cuda.set_stream(self.stream)
        try:
            fn()
        finally:
            torch.cuda.set_stream(cur)
            _WS_TAG = prev
        self.keep.extend(tensors)
        self.pending = True

    def join(self):
        wgrad_flush()
        if self.pending:
            torch.cuda.current_stream().wait_stream(self.stream)
            self.keep.clear()
            self.pending = False


_WS_TAG = "main"
SIDE = None  # set by CaptionTrainer around backward


# Optional per-launch timing (bench.py): when PROFILE is a list, GEMM / attention launches are
# bracketed by events on torch's current stream -- the stream the kernels are launched on.
PROFILE = None


class _Timed:
    def __init__(self, kind, work, nbytes=0):
        self.rec = None
        if PROFILE is not None:
            self.rec = (kind, work, nbytes, torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))

    def __enter__(self):
        if self.rec is not None:
            self.rec[3].record()

    def __exit__(self, *a):
        if self.rec is not None:
            self.rec[4].record()
            PROFILE.append(self.rec)


def _req(t, name, dtype=torch.float32):
    if not t.is_cuda:
        raise _capi.CapdecError(f"{name}: tensor is on {t.device}; the capdec ops run on the GPU only (no CPU fallback)")
    if t.dtype != dtype:
        raise _capi.CapdecError(f"{name}: dtype {t.dtype}, expected {dtype}")
    return t


def _mat(t, name):
    _req(t, name)
    if t.dim() != 2 or t.stride(1) != 1:
        raise _capi.CapdecError(f"{name}: need a 2-D tensor with unit column stride, got {tuple(t.shape)} / {t.stride()}")
    return t


_WS_NEED = {}  # (M, N, K) -> capdec_gemm_workspace_bytes, asked once per shape

# ---- measured GEMM plans ------------------------------------------------------------------------------------
# The library's cost model picks a (tile, split-K, k-tile) plan per shape; what is really fastest depends on how a
# shape's tiles fall on 256 CUs x 8 XCDs, so callers with a fixed set of shapes (a training step, a decode loop)
# can MEASURE instead: record the products of one pass (``TUNE_RECORD``), time every candidate plan of each on
# synthetic operands of the same geometry (outside any timed region), and pin the winners in the library
# (capdec_gemm_plan_set).  A plan changes speed and the rounding order of the K sum, nothing else.
TUNE_RECORD = None  # dict while recording: (trans_a, trans_b, M, N, K, dtype) -> (lda, ldb, ldc, reads logits as dlogits)
_TUNED = {}         # the same key -> (bm, bn, split, bk) already pinned in this process


def _plan_candidates(dtype, M, N, K):
    tiles_by = {F32: [(128, 128), (128, 64), (64, 128), (64, 64)], _capi.BF16: [(128, 128), (64, 64)],
                _capi.F32X3: [(128, 128), (128, 64), (64, 64)], _capi.BF16S: [(256, 256), (256, 128), (128, 128), (64, 64)]}[dtype]
    bks = (16, 32) if dtype == F32 else (0,)
    for bm, bn in tiles_by:
        tiles = -(-M // bm) * -(-N // bn)
        for split in (1, 2, 3, 4, 5, 6, 8, 12, 16):
            if split > 1 and (K // split < 128 or tiles > 4096 or tiles * split > 4096):
                continue  # a chip-filling grid gains nothing from K slices; the ticket array holds 4096 tiles
            for bk in bks:
                yield bm, bn, split, bk
            if dtype == _capi.BF16S and (bm == 128 or (bm, bn) == (256, 256)):
                yield bm, bn, split, 32  # the 128 x 128 LDS-DMA kernel with two image pairs / the 256 x 256 eight-phase kernel


def tune_gemm_plans(signatures, reps=4, min_gain=0.03, verbose=False):
    """Time every candidate plan of every recorded product and pin the fastest (if it beats the cost model's own
    choice by ``min_gain``).  ``signatures``: the dict ``TUNE_RECORD`` collected.  Returns {key: (plan, us, us_default)}."""
    global COMPUTE_DTYPE
    l = lib()
    out = {}
    saved = COMPUTE_DTYPE

    def timed(fn):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    try:
        for key, (lda, ldb, ldc, has_ce) in sorted(signatures.items()):
            if key in _TUNED:
                continue
            ta, tb, M, N, K, dtype = key
            dev = torch.device("cuda", torch.cuda.current_device())
            a, b = torch.randn((K if ta else M, lda), device=dev), torch.randn((N if tb else K, ldb), device=dev)
            if dtype == _capi.BF16S:  # the recorded strides are those of the bf16 copies
                a, b = a.to(torch.bfloat16), b.to(torch.bfloat16)
                if not ta:
                    a[:, K:] = 0  # the casts' zero padding behind the reduction extent: what lets the LDS-DMA kernels
                if tb:            # take a K that is not a multiple of 64 (the vocabulary gradient's K = V)
                    b[:, K:] = 0
            a, b = a[:, :(M if ta else K)], b[:, :(K if tb else N)]
            if dtype == _capi.BF16S:
                a._capdec_zero_padded = b._capdec_zero_padded = True
            c = torch.empty((M, ldc), device=dev)[:, :N]
            ce = None
            if has_ce:  # the loaders that turn logits into dlogits are their own instantiation: time THAT kernel
                rows, V = a.shape
                ce = (torch.full((rows,), math.log(V), device=dev), torch.randint(0, V, (rows,), device=dev), 0.1, 1.0 / rows)
            COMPUTE_DTYPE = dtype
            check(l.capdec_gemm_tune(0, 0, 0, 0), "capdec_gemm_tune")
            _WS_NEED.clear()
            t_default = timed(lambda: gemm(a, b, ta, tb, out=c, ce=ce))
            best = (None, t_default)
            for bm, bn, split, bk in _plan_candidates(dtype, M, N, K):
                check(l.capdec_gemm_tune(bm, bn, split, bk), "capdec_gemm_tune")
                _WS_NEED.clear()  # the workspace need follows the forced plan
                t = timed(lambda: gemm(a, b, ta, tb, out=c, ce=ce))
                if t < best[1]:
                    best = ((bm, bn, split, bk), t)
            check(l.capdec_gemm_tune(0, 0, 0, 0), "capdec_gemm_tune")
            if best[0] is not None and best[1] < (1.0 - min_gain) * t_default:
                check(l.capdec_gemm_plan_set(ta, tb, M, N, K, dtype, *best[0]), "capdec_gemm_plan_set")
                _TUNED[key] = best[0]
            else:
                _TUNED[key] = None
            out[key] = (_TUNED[key], best[1], t_default)
            if verbose:
                print(f"[capdec tune] {'nt'[ta]}{'nt'[tb]} {M}x{N}x{K} dtype {dtype}: default {t_default:7.1f} us -> "
                      f"{best[0]} {best[1]:7.1f} us", file=sys.stderr, flush=True)
    finally:
        COMPUTE_DTYPE = saved
        l.capdec_gemm_tune(0, 0, 0, 0)
        _WS_NEED.clear()
    return out


def gemm(a, b, trans_a, trans_b, out=None, bias=None, addend=None, pre=None, aux=None, colsum=None, act=ACT_NONE,
         drop_p=0.0, seed=0, ce_partials=None, ce=None, ce_targets=None):
    """out[M,N] = epilogue(op(a) @ op(b)); see capdec_gemm in include/capdec.h.

    ``ce_partials``: [M, ce_tiles(M, N, K)] x 4 floats receiving the per-tile softmax partials of the logits this
    product writes (fused cross-entropy, forward); with ``ce_targets`` (int64 [M]) their fourth component carries the
    rows' fp32 target logits, which a bf16 ``out`` requires.  ``ce = (lse, targets, eps, grad_scale)``: ``a`` holds
    LOGITS and is read as dlogits (fused cross-entropy, backward; exact-fp32 compute type only)."""
    if COMPUTE_DTYPE == _capi.BF16S:
        if ce is not None:
            raise _capi.CapdecError("gemm: the bf16-storage products take dlogits from ops.ce_dlogits_bf16, not ce=")
        a, b = _bf16_operand(a, "gemm.a"), _bf16_operand(b, "gemm.b")
    else:
        _mat(a, "gemm.a"), _mat(b, "gemm.b")
    M, K = (a.shape[1], a.shape[0]) if trans_a else (a.shape[0], a.shape[1])
    Kb, N = (b.shape[1], b.shape[0]) if trans_b else (b.shape[0], b.shape[1])
    if K != Kb:
        raise _capi.CapdecError(f"gemm: inner dimensions differ ({K} vs {Kb})")
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    out16 = out.dtype == torch.bfloat16  # the result is wanted as bf16 only (capdec_epilogue.c16)
    if out16:
        _bf16_operand(out, "gemm.out")
    else:
        _mat(out, "gemm.out")
    if tuple(out.shape) != (M, N):
        raise _capi.CapdecError(f"gemm: out is {tuple(out.shape)}, expected {(M, N)}")
    if TUNE_RECORD is not None:
        TUNE_RECORD.setdefault((int(trans_a), int(trans_b), M, N, K, COMPUTE_DTYPE),
                               (a.stride(0), b.stride(0), out.stride(0), ce is not None))
    ep = _capi.Epilogue()
    ep.bias, ep.addend, ep.pre, ep.aux, ep.colsum = ptr(bias), ptr(addend), ptr(pre), ptr(aux), ptr(colsum)
    ep.ld_addend = addend.stride(0) if addend is not None else 0
    # pre / aux are fp32 [M, N] blocks addressed with the C row stride; with a bf16-only result there is no fp32 C
    # and the ldc argument carries THEIR row stride instead (include/capdec.h, capdec_epilogue.c16)
    ldc = out.stride(0)
    if out16:
        ldc = pre.stride(0) if pre is not None else (aux.stride(0) if aux is not None else N)
    for t, n in ((pre, "pre"), (aux, "aux")):
        if t is not None and (t.stride(0) != ldc or tuple(t.shape) != (M, N)):
            raise _capi.CapdecError(f"gemm: {n} must have the shape and row stride of out (of each other for a bf16 out)")
    if bias is not None and bias.numel() != N:
        raise _capi.CapdecError("gemm: bias must have N elements")
    if addend is not None and tuple(addend.shape) != (M, N):
        raise _capi.CapdecError("gemm: addend must be [M,N]")
    if colsum is not None and colsum.numel() != M:
        raise _capi.CapdecError("gemm: colsum must have M elements")
    ep.act, ep.drop_p, ep.drop_seed = act, float(drop_p), int(seed) & 0xFFFFFFFFFFFFFFFF
    if COMPUTE_DTYPE == _capi.BF16S:
        ep.bf16_unpadded = _unpadded_bits(a, b, trans_a, trans_b, K)
    if out16:
        if ce_partials is not None and ce_targets is None:
            raise _capi.CapdecError("gemm: a bf16 result with ce_partials needs ce_targets (the loss's target logits)")
        ep.c16, ep.ldc16 = ptr(out), out.stride(0)
    l = lib()
    if ce_partials is not None:
        tiles = l.capdec_gemm_ce_tiles(M, N, K, COMPUTE_DTYPE)
        if tuple(ce_partials.shape) != (M, tiles, 4) or not ce_partials.is_contiguous():
            raise _capi.CapdecError(f"gemm: ce_partials must be a contiguous [M, {tiles}, 4] tensor")
        ep.ce_partials = ptr(_req(ce_partials, "gemm.ce_partials"))
        if ce_targets is not None:
            if _req(ce_targets, "gemm.ce_targets", torch.int64).numel() != M or not ce_targets.is_contiguous():
                raise _capi.CapdecError("gemm: ce_targets must be a contiguous int64 tensor with one entry per row")
            ep.ce_targets = ptr(ce_targets)
    elif ce_targets is not None:
        raise _capi.CapdecError("gemm: ce_targets goes with ce_partials")
    if ce is not None:
        lse, targets, eps, gs = ce
        rows = a.shape[0]
        if _req(lse, "gemm.ce.lse").numel() != rows or _req(targets, "gemm.ce.targets", torch.int64).numel() != rows \
                or not (lse.is_contiguous() and targets.is_contiguous()):
            raise _capi.CapdecError("gemm: ce needs contiguous lse / targets with one entry per row of the logits block")
        ep.ce_lse, ep.ce_targets, ep.ce_eps, ep.ce_grad_scale = ptr(lse), ptr(targets), float(eps), float(gs)
    need = _WS_NEED.get((M, N, K))
    if need is None:
        need = _WS_NEED[(M, N, K)] = l.capdec_gemm_workspace_bytes(M, N, K)
    ws = _capi.workspace(need, a.device, "gemm:" + _WS_TAG, zero=True) if need else None
    args = (int(trans_a), int(trans_b), M, N, K, ptr(a), a.stride(0), ptr(b), b.stride(0), None if out16 else ptr(out), ldc,
            C.byref(ep), COMPUTE_DTYPE, ptr(ws), ws.numel() if ws is not None else 0, stream())
    if PROFILE is None:  # the common case: no per-launch events, no label formatting
        check(l.capdec_gemm(*args), "capdec_gemm")
        return out
    with _Timed(f"gemm_{'t' if trans_a else 'n'}{'t' if trans_b else 'n'}_{M}x{N}x{K}", 2.0 * M * N * K):
        check(l.capdec_gemm(*args), "capdec_gemm")
    return out


def linear_fwd(x, w, bias=None, out=None, act=ACT_NONE, pre=None, drop_p=0.0, seed=0, addend=None, ce_partials=None,
               ce_targets=None):
    """y = addend + dropout(act(x @ w.T + bias)); x [M,K], w [N,K]."""
    return gemm(x, w, False, True, out=out, bias=bias, addend=addend, pre=pre, act=act, drop_p=drop_p, seed=seed,
                ce_partials=ce_partials, ce_targets=ce_targets)


def linear_dgrad(dy, w, out=None, addend=None, gelu_pre=None, drop_p=0.0, seed=0, ce=None):
    """dx = addend + dropout_mask * gelu'(gelu_pre) * (dy @ w); dy [M,N], w [N,K].  ``ce``: see gemm (dy = logits)."""
    act = ACT_GELU_BWD if gelu_pre is not None else ACT_NONE
    return gemm(dy, w, False, False, out=out, addend=addend, aux=gelu_pre, act=act, drop_p=drop_p, seed=seed, ce=ce)


# ---- grouped weight gradients --------------------------------------------------------------------------------------
# The weight gradients of a decoder layer are small products (64 - 256 output tiles each) that are all ready by the end
# of the layer's backward and that nothing needs before the optimizer: with a side lane active, linear_wgrad only queues
# them (caller-owned outputs) and wgrad_flush() -- called by the engine at the end of a block's backward -- hands them to
# the lane in ONE hand-over; with GROUP_WGRADS on, the exact-fp32 ones go out as launches of up to GEMM_GROUP_MAX
# products each (capdec_gemm_grouped: one chip-filling, unsplit grid).
# Off by default: measured at C2 with the one-call layer plans, single launches under the tuned plans, handed over at two
# points inside a layer, run the step in 14.64 ms against 14.93 with one grouped launch per layer (profiles/r03_grouped_gemm.txt)
GROUP_WGRADS = os.environ.get("CAPDEC_GROUP_WGRADS", "0") == "1"
# block tile of the grouped launches ((0, 0) = the library's default, 64 x 64); CAPDEC_GROUP_TILE=128x64 for experiments
GROUP_TILE = tuple(int(v) for v in os.environ.get("CAPDEC_GROUP_TILE", "0x0").split("x"))
_WGRAD_QUEUE = []


def gemm_grouped(products, tile=(0, 0)):
    """``products``: [(a, b, trans_a, trans_b, out, colsum)], fp32 matrices, one layout -- one launch (capdec_gemm_grouped)."""
    n = len(products)
    descs = (_capi.GemmDesc * n)()
    eps = (_capi.Epilogue * n)()
    flops = 0.0
    for d, ep, (a, b, ta, tb, out, colsum) in zip(descs, eps, products):
        _mat(a, "gemm_grouped.a"), _mat(b, "gemm_grouped.b"), _mat(out, "gemm_grouped.out")
        M, K = (a.shape[1], a.shape[0]) if ta else (a.shape[0], a.shape[1])
        Kb, N = (b.shape[1], b.shape[0]) if tb else (b.shape[0], b.shape[1])
        if K != Kb or tuple(out.shape) != (M, N) or (colsum is not None and colsum.numel() != M):
            raise _capi.CapdecError(f"gemm_grouped: shapes {tuple(a.shape)} x {tuple(b.shape)} -> {tuple(out.shape)} do not fit")
        ep.colsum = ptr(colsum)
        d.transA, d.transB, d.M, d.N, d.K = int(ta), int(tb), M, N, K
        d.A, d.lda, d.B, d.ldb, d.C, d.ldc = ptr(a), a.stride(0), ptr(b), b.stride(0), ptr(out), out.stride(0)
        d.ep = C.pointer(ep)
        flops += 2.0 * M * N * K
    args = (descs, n, int(tile[0]), int(tile[1]), stream())
    if PROFILE is None:
        check(lib().capdec_gemm_grouped(*args), "capdec_gemm_grouped")
        return
    with _Timed(f"gemm_grouped_{n}", flops):
        check(lib().capdec_gemm_grouped(*args), "capdec_gemm_grouped")


def _groupable(t):
    return t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1 and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0


def wgrad_flush(stage=False):
    """Hand the queued weight gradients to the side lane: ONE hand-over (event record + wait, stream switch) for all
    of them; the exact-fp32 ones without cross-entropy fusion go out as grouped launches, the rest one by one.
    ``stage=True`` marks a point INSIDE a block (end of its FFN / cross-attention part): the queue is handed over
    there only when its products would go out one by one anyway, so that the lane starts them while the block's
    backward is still running; products that wait for a grouped launch stay queued until the block's end."""
    global _WGRAD_QUEUE, COMPUTE_DTYPE
    if not _WGRAD_QUEUE:
        return
    if stage and GROUP_WGRADS and TUNE_RECORD is None and all(e[5] == F32 and e[4] is None for e in _WGRAD_QUEUE):
        return
    q, _WGRAD_QUEUE = _WGRAD_QUEUE, []
    group = [e for e in q if e[5] == F32 and e[4] is None and GROUP_WGRADS and TUNE_RECORD is None and _groupable(e[0])
             and _groupable(e[1]) and _groupable(e[2])]
    if len(group) < 2:
        group = []
    single = [e for e in q if not any(e is g for g in group)]
    # longest reductions first: their tiles start first and the short ones fill the tail of the launch
    group.sort(key=lambda e: -e[0].shape[0])

    def issue():
        global COMPUTE_DTYPE
        for i in range(0, len(group), _capi.GEMM_GROUP_MAX):
            part = group[i:i + _capi.GEMM_GROUP_MAX]
            if len(part) == 1:
                single.append(part[0])
            else:
                gemm_grouped([(dy, x, True, False, dw, db) for dy, x, dw, db, _, _ in part], GROUP_TILE)
        saved = COMPUTE_DTYPE
        try:
            for dy, x, dw, db, ce, dtype in single:
                COMPUTE_DTYPE = dtype
                gemm(dy, x, True, False, out=dw, colsum=db, ce=ce)
        finally:
            COMPUTE_DTYPE = saved

    if SIDE is None:  # (the lane went away between queueing and flushing: run them here)
        issue()
    else:
        SIDE.run(issue, *[t for e in q for t in e[:2]])


def linear_wgrad(dy, x, dw=None, db=None, ce=None):
    """dw[N,K] = dy.T @ x, db[N] = column sums of dy (fused into the same kernel).  ``ce``: see gemm (dy = logits).

    With a SideStream active and caller-owned outputs, the product is enqueued on the side stream
    (ordered after everything enqueued so far on the current stream) and the call returns at once."""
    if SIDE is not None and dw is not None:
        # queued: wgrad_flush() (the engine calls it at the end of a block's backward) hands the block's weight
        # gradients to the lane together -- one hand-over instead of one per product
        if COMPUTE_DTYPE == _capi.BF16S:  # bf16 copies are made (or found) on the caller's stream, not on the lane
            dy, x = _bf16_operand(dy, "linear_wgrad.dy"), _bf16_operand(x, "linear_wgrad.x")
        _WGRAD_QUEUE.append((dy, x, dw, db, ce, COMPUTE_DTYPE))
        return dw
    return gemm(dy, x, True, False, out=dw, colsum=db, ce=ce)


def layernorm_fwd(x, gamma, beta, eps=1e-5, act=ACT_NONE, drop_p=0.0, seed=0, save_stats=True, out=None):
    _mat(x, "layernorm.x")
    if not x.is_contiguous():
        raise _capi.CapdecError("layernorm_fwd: x must be contiguous")
    M, d = x.shape
    y = torch.empty_like(x) if out is None else out
    mean = torch.empty(M, dtype=torch.float32, device=x.device) if save_stats else None
    rstd = torch.empty(M, dtype=torch.float32, device=x.device) if save_stats else None
    if COMPUTE_DTYPE == _capi.BF16S and BF16_MEMO is not None:
        # bf16 storage inside a training step: every LayerNorm result is the operand of a matrix product, so the kernel
        # leaves its bf16 copy as well and the step's memo hands it to that product (no cast pass)
        y16 = torch.empty((M, _bf16_pitch(d)), dtype=torch.bfloat16, device=x.device)
        check(lib().capdec_layernorm_fwd_y16(ptr(x), ptr(gamma), ptr(beta), ptr(y), ptr(y16), y16.stride(0), ptr(mean),
                                             ptr(rstd), M, d, float(eps), act, float(drop_p), int(seed), stream()),
              "capdec_layernorm_fwd_y16")
        BF16_MEMO[(y.data_ptr(), M, d, y.stride(0))] = (y, _zero_padded(y16[:, :d]))  # the kernel zeroes the padding
        return y, mean, rstd
    check(lib().capdec_layernorm_fwd(ptr(x), ptr(gamma), ptr(beta), ptr(y), ptr(mean), ptr(rstd), M, d, float(eps), act,
                                     float(drop_p), int(seed), stream()), "capdec_layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy, x, gamma, beta, mean, rstd, act=ACT_NONE, drop_p=0.0, seed=0, drop2_p=0.0, seed2=0,
                  want_dx_drop=False, dgamma=None, dbeta=None, dx_drop16=False):
    """``dx_drop16``: dx_drop comes back as a bf16 matrix ONLY (capdec_layernorm_bwd_dx16; with drop2_p == 0 it is the bf16
    copy of dx) -- for callers whose only readers of it are bf16-storage matrix products."""
    _mat(x, "layernorm_bwd.x"), _mat(dy, "layernorm_bwd.dy")
    if not (x.is_contiguous() and dy.is_contiguous()):
        raise _capi.CapdecError("layernorm_bwd: x and dy must be contiguous")
    M, d = x.shape
    dx = torch.empty_like(x)
    dx_drop = torch.empty_like(x) if want_dx_drop and not dx_drop16 else None
    if dgamma is None:
        dgamma = torch.empty(d, dtype=torch.float32, device=x.device)
    if dbeta is None:
        dbeta = torch.empty(d, dtype=torch.float32, device=x.device)
    l = lib()
    need = l.capdec_layernorm_bwd_workspace_bytes(M, d)
    if dx_drop16:
        dd16 = torch.empty((M, _bf16_pitch(d)), dtype=torch.bfloat16, device=x.device)
        part = torch.empty(need, dtype=torch.uint8, device=x.device) if SIDE is not None else _capi.workspace(need, x.device)
        check(l.capdec_layernorm_bwd_dx16(ptr(dy), ptr(x), ptr(gamma), ptr(beta), ptr(mean), ptr(rstd), ptr(dx), ptr(dd16),
                                          dd16.stride(0), ptr(part), part.numel(), M, d, act, float(drop_p), int(seed),
                                          float(drop2_p), int(seed2), stream()), "capdec_layernorm_bwd_dx16")
        fin = lambda: check(l.capdec_layernorm_bwd_dparams(ptr(part), ptr(dgamma), ptr(dbeta), M, d, stream()),
                            "capdec_layernorm_bwd_dparams")
        if SIDE is not None:
            SIDE.run(fin, part, dgamma, dbeta)
        else:
            fin()
        return dx, _zero_padded(dd16[:, :d]), dgamma, dbeta
    if SIDE is not None:
        # the parameter gradients are weight gradients: their reduction runs on the weight-gradient lane, from a
        # private block of partials (the shared workspace would be overwritten by the next call on this stream)
        part = torch.empty(need, dtype=torch.uint8, device=x.device)
        check(l.capdec_layernorm_bwd_dx(ptr(dy), ptr(x), ptr(gamma), ptr(beta), ptr(mean), ptr(rstd), ptr(dx), ptr(dx_drop),
                                        ptr(part), part.numel(), M, d, act, float(drop_p), int(seed), float(drop2_p),
                                        int(seed2), stream()), "capdec_layernorm_bwd_dx")
        SIDE.run(lambda: check(l.capdec_layernorm_bwd_dparams(ptr(part), ptr(dgamma), ptr(dbeta), M, d, stream()),
                               "capdec_layernorm_bwd_dparams"), part, dgamma, dbeta)
        return dx, dx_drop, dgamma, dbeta
    ws = _capi.workspace(need, x.device)
    check(l.capdec_layernorm_bwd(ptr(dy), ptr(x), ptr(gamma), ptr(beta), ptr(mean), ptr(rstd), ptr(dx), ptr(dx_drop),
                                 ptr(dgamma), ptr(dbeta), M, d, act, float(drop_p), int(seed), float(drop2_p),
                                 int(seed2), ptr(ws), ws.numel(), stream()), "capdec_layernorm_bwd")
    return dx, dx_drop, dgamma, dbeta


def embed_fwd(ids, E, pe, t0=0, drop_p=0.0, seed=0):
    """out[b,t,:] = dropout(E[ids[b,t]] + pe[t0+t]); ids int64 [B,T], E [V,d], pe [>=t0+T, d] contiguous."""
    _req(ids, "embed.ids", torch.int64)
    ids = ids.contiguous()
    B, T = ids.shape
    V, d = E.shape
    if pe.shape[0] < t0 + T or pe.shape[-1] != d or not pe.is_contiguous():
        raise _capi.CapdecError("embed_fwd: positional table too short / wrong width / not contiguous")
    out = torch.empty((B, T, d), dtype=torch.float32, device=E.device)
    check(lib().capdec_embed_fwd(ptr(ids), ptr(_req(E, "embed.E")), ptr(_req(pe, "embed.pe")), ptr(out), B, T, d, V,
                                 int(t0), float(drop_p), int(seed), stream()), "capdec_embed_fwd")
    return out


def check_ids():
    """Raise IndexError if any token id given to embed_fwd on the current device since the last check was outside
    [0, V): the kernels clamp such ids (an out-of-bounds guard), the reference's nn.Embedding raises.  Synchronises."""
    n = C.c_uint(0)
    check(lib().capdec_embed_bad_ids(C.byref(n), 1), "capdec_embed_bad_ids")
    if n.value:
        raise IndexError(f"{n.value} token id(s) outside the vocabulary were given to the embedding since the last check "
                         "(clamped on the device; the reference raises an index error: is the dataset's vocabulary the "
                         "checkpoint's?)")


def embed_bwd_workspace(B, T, d, V, device):
    """A private workspace for the plan / apply pair (the shared per-device workspace is reused by every GEMM)."""
    return torch.empty(lib().capdec_embed_bwd_workspace_bytes(B, T, d, V), dtype=torch.uint8, device=device)


def embed_bwd_plan(ids, d, V, ws):
    """The ids-only half of embed_bwd (stable sort of the rows by token) into ``ws``; see capdec_embed_bwd_plan."""
    ids = _req(ids, "embed_bwd_plan.ids", torch.int64).contiguous()
    B, T = ids.shape
    check(lib().capdec_embed_bwd_plan(ptr(ids), B, T, d, V, ptr(ws), ws.numel(), stream()), "capdec_embed_bwd_plan")


def embed_bwd(ids, dout, V, drop_p=0.0, seed=0, dE=None, planned_ws=None):
    """dE[v] = sum of the (dropout-masked) rows of dout whose id is v.  ``planned_ws``: a workspace already filled by
    ``embed_bwd_plan`` for these ids (then only the summation runs here)."""
    ids = ids.contiguous()
    B, T = ids.shape
    d = dout.shape[-1]
    dout = dout.contiguous()
    if dE is None:
        dE = torch.empty((V, d), dtype=torch.float32, device=dout.device)
    l = lib()
    if planned_ws is not None:
        check(l.capdec_embed_bwd_apply(ptr(ids), ptr(_req(dout, "embed_bwd.dout")), ptr(dE), B, T, d, V, float(drop_p),
                                       int(seed), ptr(planned_ws), planned_ws.numel(), stream()), "capdec_embed_bwd_apply")
        return dE
    ws = _capi.workspace(l.capdec_embed_bwd_workspace_bytes(B, T, d, V), dout.device, _WS_TAG)
    check(l.capdec_embed_bwd(ptr(ids), ptr(_req(dout, "embed_bwd.dout")), ptr(dE), B, T, d, V, float(drop_p), int(seed),
                             ptr(ws), ws.numel(), stream()), "capdec_embed_bwd")
    return dE


def cross1_fwd(U, bias, addend, B, T, H, attn_p=0.0, attn_seed=0, out_p=0.0, out_seed=0):
    """Cross-attention branch over ONE memory position: addend + dropout_out(bias + sum_h keep_h/(1-p) U[b,h,:]);
    U [B*H, d] = per-image, per-head out-projected values, addend [B*T, d].  See capdec_cross1_fwd."""
    for t, n in ((U, "cross1.U"), (addend, "cross1.addend")):
        if not _mat(t, n).is_contiguous():
            raise _capi.CapdecError(f"{n}: must be contiguous")
    d = U.shape[1]
    if U.shape[0] != B * H or tuple(addend.shape) != (B * T, d) or _req(bias, "cross1.bias").numel() != d:
        raise _capi.CapdecError(f"cross1_fwd: shapes U {tuple(U.shape)} bias {tuple(bias.shape)} addend {tuple(addend.shape)} "
                                f"do not match B={B} T={T} H={H}")
    out = torch.empty_like(addend)
    check(lib().capdec_cross1_fwd(ptr(U), ptr(bias), ptr(addend), ptr(out), B, T, H, d, float(attn_p),
                                  int(attn_seed) & 0xFFFFFFFFFFFFFFFF, float(out_p), int(out_seed) & 0xFFFFFFFFFFFFFFFF,
                                  stream()), "capdec_cross1_fwd")
    return out


def cross1_bwd(g, B, T, H, attn_p=0.0, attn_seed=0):
    """(dU [B*H, d], dbias_part [B, d]) from g [B*T, d], the gradient at the branch output (output dropout applied)."""
    if not _mat(g, "cross1.g").is_contiguous() or g.shape[0] != B * T:
        raise _capi.CapdecError(f"cross1_bwd: g {tuple(g.shape)} must be a contiguous [B*T, d] with B={B} T={T}")
    d = g.shape[1]
    dU = torch.empty((B * H, d), dtype=torch.float32, device=g.device)
    dbp = torch.empty((B, d), dtype=torch.float32, device=g.device)
    check(lib().capdec_cross1_bwd(ptr(g), ptr(dU), ptr(dbp), B, T, H, d, float(attn_p), int(attn_seed) & 0xFFFFFFFFFFFFFFFF,
                                  stream()), "capdec_cross1_bwd")
    return dU, dbp


def cross1_spread(v, B, H):
    """vblk [B*H, d]: row (b, h) = v[b] inside head h's slice, zero elsewhere (v: a [B, d] row-strided fp32 view)."""
    _mat(v, "cross1_spread.v")
    d = v.shape[1]
    vblk = torch.empty((B * H, d), dtype=torch.float32, device=v.device)
    check(lib().capdec_cross1_spread(ptr(v), v.stride(0), ptr(vblk), B, H, d, stream()), "capdec_cross1_spread")
    return vblk


def cross1_gather(dvblk, dbias_part, dk, dv, dbias, B, H):
    """dk = 0, dv[b, j] = dvblk[(b, head of j), j], dbias = column sums of dbias_part (see capdec_cross1_gather)."""
    d = dvblk.shape[1]
    _mat(dk, "cross1_gather.dk"), _mat(dv, "cross1_gather.dv")
    check(lib().capdec_cross1_gather(ptr(_req(dvblk, "cross1_gather.dvblk")), ptr(_req(dbias_part, "cross1_gather.dbias_part")),
                                     ptr(dk), dk.stride(0), ptr(dv), dv.stride(0), ptr(_req(dbias, "cross1_gather.dbias")),
                                     B, H, d, stream()), "capdec_cross1_gather")


def attn_fwd(q, k, v, B, H, Tq, Tk, causal, drop_p=0.0, seed=0, out=None, save_lse=True):
    """q [B*Tq, >=H*dh] / k, v [B*Tk, ...] row-strided 2-D views; returns o [B*Tq, H*dh], lse [B,H,Tq].
    bf16 ``k`` / ``v`` (both, Tk > 1; views of a block a GEMM wrote as bf16) go through capdec_attn_fwd_kv16.
    bf16 ``q``: every matrix is bf16, ``o`` included (capdec_attn_fwd_bf16, head dims 64 / 128)."""
    if q.dtype == torch.bfloat16:
        return _attn_fwd_bf16(q, k, v, B, H, Tq, Tk, causal, drop_p, seed, out, save_lse)
    _mat(q, "attn.q")
    kv16 = k.dtype == torch.bfloat16
    if kv16:
        _bf16_operand(k, "attn.k"), _bf16_operand(v, "attn.v")
    else:
        _mat(k, "attn.k"), _mat(v, "attn.v")
    fwd = lib().capdec_attn_fwd_kv16 if kv16 else lib().capdec_attn_fwd
    D = q.shape[1]
    dh = D // H
    o = torch.empty((B * Tq, D), dtype=torch.float32, device=q.device) if out is None else out
    lse = torch.empty((B, H, Tq), dtype=torch.float32, device=q.device) if save_lse else None
    # algorithmic bytes: read Q, K, V, write O
    with _Timed(f"attn_fwd_{'causal' if causal else 'cross'}_{Tq}x{Tk}", 4.0 * B * H * Tq * Tk * dh,
                4.0 * B * D * (2 * Tq + (1 if kv16 else 2) * Tk)):
        check(fwd(ptr(q), q.stride(0), ptr(k), k.stride(0), ptr(v), v.stride(0), ptr(o), o.stride(0),
                                    ptr(lse), B, H, Tq, Tk, dh, int(causal), 1.0 / math.sqrt(dh), float(drop_p),
                                    int(seed), stream()), "capdec_attn_fwd")
    return o, lse


def bf16_matrix(rows, cols, device):
    """An uninitialised bf16 [rows, cols] view with the row pitch of ops.cast_bf16 (for results that a kernel writes
    as bf16: ``gemm(out=...)``, the bf16 attention).  The padding columns are NOT zeroed: fine whenever cols is a
    multiple of 64 (the LDS-DMA products then never read them); zero them yourself otherwise."""
    return torch.empty((rows, _bf16_pitch(cols)), dtype=torch.bfloat16, device=device)[:, :cols]


def attn16_ok(H, D):
    """The all-bf16 attention kernels (capdec_attn_*_bf16) cover head dims 64 and 128."""
    return D % H == 0 and D // H in (64, 128)


def _attn_fwd_bf16(q, k, v, B, H, Tq, Tk, causal, drop_p, seed, out, save_lse):
    for t, n in ((q, "attn.q"), (k, "attn.k"), (v, "attn.v")):
        if t.dtype != torch.bfloat16:
            raise _capi.CapdecError(f"attn_fwd: bf16 q needs bf16 k and v, {n} is {t.dtype}")
        _bf16_operand(t, n)
    D = q.shape[1]
    dh = D // H
    o = bf16_matrix(B * Tq, D, q.device) if out is None else _bf16_operand(out, "attn.out")
    lse = torch.empty((B, H, Tq), dtype=torch.float32, device=q.device) if save_lse else None
    # algorithmic bytes: read Q, K, V, write O, all bf16
    with _Timed(f"attn_fwd_{'causal' if causal else 'cross'}_{Tq}x{Tk}", 4.0 * B * H * Tq * Tk * dh,
                2.0 * B * D * (2 * Tq + 2 * Tk)):
        check(lib().capdec_attn_fwd_bf16(ptr(q), q.stride(0), ptr(k), k.stride(0), ptr(v), v.stride(0), ptr(o), o.stride(0),
                                         ptr(lse), B, H, Tq, Tk, dh, int(causal), 1.0 / math.sqrt(dh), float(drop_p),
                                         int(seed), stream()), "capdec_attn_fwd_bf16")
    return o, lse


def _attn_bwd_bf16(q, k, v, o, do, lse, B, H, Tq, Tk, causal, drop_p, seed, dq, dk, dv):
    D = q.shape[1]
    dh = D // H
    dq = bf16_matrix(B * Tq, D, q.device) if dq is None else dq
    dk = bf16_matrix(B * Tk, D, q.device) if dk is None else dk
    dv = bf16_matrix(B * Tk, D, q.device) if dv is None else dv
    for t, n in ((q, "q"), (k, "k"), (v, "v"), (o, "o"), (do, "do"), (dq, "dq"), (dk, "dk"), (dv, "dv")):
        if t.dtype != torch.bfloat16:
            raise _capi.CapdecError(f"attn_bwd: bf16 q needs every matrix bf16, {n} is {t.dtype}")
        _bf16_operand(t, "attn_bwd." + n)
    delta = torch.empty((B, H, Tq), dtype=torch.float32, device=q.device)
    # algorithmic bytes: read Q, O, dO, K, V; write dQ, dK, dV, all bf16
    with _Timed(f"attn_bwd_{'causal' if causal else 'cross'}_{Tq}x{Tk}", 10.0 * B * H * Tq * Tk * dh,
                2.0 * B * D * (4 * Tq + 4 * Tk)):
        check(lib().capdec_attn_bwd_bf16(ptr(q), q.stride(0), ptr(k), k.stride(0), ptr(v), v.stride(0), ptr(o), o.stride(0),
                                         ptr(do), do.stride(0), ptr(_req(lse, "attn_bwd.lse")), ptr(delta), ptr(dq),
                                         dq.stride(0), ptr(dk), dk.stride(0), ptr(dv), dv.stride(0), B, H, Tq, Tk, dh,
                                         int(causal), 1.0 / math.sqrt(dh), float(drop_p), int(seed), stream()),
              "capdec_attn_bwd_bf16")
    return dq, dk, dv


def attn_bwd(q, k, v, o, do, lse, B, H, Tq, Tk, causal, drop_p=0.0, seed=0, dq=None, dk=None, dv=None, need_dq=True):
    """``need_dq=False`` is accepted for Tk == 1 only (dq is exactly zero there and is then not written).
    bf16 ``dk`` / ``dv`` (both, Tk > 1): the gradients are written as bf16 matrices (capdec_attn_bwd_dkv16).
    bf16 ``q``: every matrix is bf16 (capdec_attn_bwd_bf16)."""
    if q.dtype == torch.bfloat16:
        return _attn_bwd_bf16(q, k, v, o, do, lse, B, H, Tq, Tk, causal, drop_p, seed, dq, dk, dv)
    D = q.shape[1]
    dh = D // H
    _mat(do, "attn_bwd.do")
    if not need_dq and Tk != 1:
        raise _capi.CapdecError("attn_bwd: need_dq=False requires Tk == 1")
    if need_dq:
        dq = torch.empty((B * Tq, D), dtype=torch.float32, device=q.device) if dq is None else dq
    dk = torch.empty((B * Tk, D), dtype=torch.float32, device=q.device) if dk is None else dk
    dv = torch.empty((B * Tk, D), dtype=torch.float32, device=q.device) if dv is None else dv
    delta = torch.empty((B, H, Tq), dtype=torch.float32, device=q.device)
    # algorithmic bytes: read Q, K, V, O, dO; write dQ, dK, dV
    kv16 = dk.dtype == torch.bfloat16
    if kv16 != (dv.dtype == torch.bfloat16):
        raise _capi.CapdecError("attn_bwd: dk and dv must have the same dtype")
    in16 = k.dtype == torch.bfloat16  # K / V themselves are bf16: then so are dK / dV
    if in16 and not (kv16 and v.dtype == torch.bfloat16):
        raise _capi.CapdecError("attn_bwd: bf16 k / v need bf16 dk / dv")
    fn = lib().capdec_attn_bwd_kv16 if in16 else (lib().capdec_attn_bwd_dkv16 if kv16 else lib().capdec_attn_bwd)
    # algorithmic bytes: read Q, O, dO (fp32), K, V; write dQ (fp32), dK, dV -- bf16 blocks count half
    with _Timed(f"attn_bwd_{'causal' if causal else 'cross'}_{Tq}x{Tk}", 10.0 * B * H * Tq * Tk * dh,
                4.0 * B * D * (4 * Tq + ((2 if in16 else 3) if kv16 else 4) * Tk)):
        check(fn(ptr(q), q.stride(0), ptr(k), k.stride(0), ptr(v), v.stride(0), ptr(o), o.stride(0),
                                    ptr(do), do.stride(0), ptr(lse), ptr(delta), ptr(dq), dq.stride(0) if dq is not None else D,
                                    ptr(dk),
                                    dk.stride(0), ptr(dv), dv.stride(0), B, H, Tq, Tk, dh, int(causal),
                                    1.0 / math.sqrt(dh), float(drop_p), int(seed), stream()), "capdec_attn_bwd")
    return dq, dk, dv


def softmax_ce(logits, targets, eps=0.1, grad_scale=None, want_grad=True, in_place=False):
    """Label-smoothed CE rows + (optionally) dlogits; logits [M,V], targets int64 [M]."""
    _mat(logits, "softmax_ce.logits")
    _req(targets, "softmax_ce.targets", torch.int64)
    M, V = logits.shape
    loss_rows = torch.empty(M, dtype=torch.float32, device=logits.device)
    dl = None
    if want_grad:
        dl = logits if in_place else torch.empty((M, V), dtype=torch.float32, device=logits.device)
    gs = 1.0 / M if grad_scale is None else grad_scale
    check(lib().capdec_softmax_ce(ptr(logits), logits.stride(0), ptr(targets.contiguous()), ptr(loss_rows), ptr(dl),
                                  dl.stride(0) if dl is not None else 0, M, V, float(eps), float(gs), stream()),
          "capdec_softmax_ce")
    return loss_rows, dl


def ce_tiles(M, V, K):
    """Column tiles of the vocabulary product [M, K] x [V, K]^T in the current compute type (= second dimension of
    its cross-entropy partials)."""
    return lib().capdec_gemm_ce_tiles(M, V, K, COMPUTE_DTYPE)


def ce_finalize(partials, logits, targets, eps=0.1):
    """(lse [M], loss_rows [M]) from the per-tile softmax partials the vocabulary product left (fused cross-entropy).
    bf16 ``logits`` (only their shape is used): the target logits are the partials' fourth components, i.e. the
    product ran with ``ce_targets``."""
    z16 = logits.dtype == torch.bfloat16
    if not z16:
        _mat(logits, "ce_finalize.logits")
    M, V = logits.shape
    targets = _req(targets, "ce_finalize.targets", torch.int64).contiguous()
    lse = torch.empty(M, dtype=torch.float32, device=logits.device)
    rows = torch.empty(M, dtype=torch.float32, device=logits.device)
    check(lib().capdec_ce_finalize(ptr(_req(partials, "ce_finalize.partials")), partials.shape[1], None if z16 else ptr(logits),
                                   0 if z16 else logits.stride(0), ptr(targets), ptr(lse), ptr(rows), M, V, float(eps), stream()),
          "capdec_ce_finalize")
    return lse, rows


def ce_dlogits(logits, lse, targets, eps, grad_scale):
    """logits -> dlogits in place (the pass form of the gradient transform the fp32 loaders do on the fly)."""
    _mat(logits, "ce_dlogits.logits")
    M, V = logits.shape
    check(lib().capdec_ce_dlogits(ptr(logits), logits.stride(0), ptr(_req(lse, "ce_dlogits.lse")),
                                  ptr(_req(targets, "ce_dlogits.targets", torch.int64).contiguous()), M, V, float(eps),
                                  float(grad_scale), stream()), "capdec_ce_dlogits")
    return logits


def ce_dlogits_bf16(logits, lse, targets, eps, grad_scale):
    """The label-smoothed cross-entropy gradient of a logits block as a bf16 matrix (the operand form of the
    bf16-storage products); fp32 logits are left as they are, bf16 logits (a ``bf16_matrix`` the vocabulary product
    wrote) are REPLACED by their gradient."""
    rows, V = logits.shape
    if _req(lse, "ce_dlogits_bf16.lse").numel() != rows or _req(targets, "ce_dlogits_bf16.targets", torch.int64).numel() != rows \
            or not (lse.is_contiguous() and targets.is_contiguous()):
        raise _capi.CapdecError("ce_dlogits_bf16: lse / targets must be contiguous with one entry per logits row")
    if logits.dtype == torch.bfloat16:
        _bf16_operand(logits, "ce_dlogits_bf16.logits")
        if logits.stride(0) < _bf16_pitch(V):
            raise _capi.CapdecError("ce_dlogits_bf16: bf16 logits need the row pitch of ops.bf16_matrix")
        check(lib().capdec_ce_dlogits_bf16_inplace(ptr(logits), logits.stride(0), ptr(lse), ptr(targets), rows, V, float(eps),
                                                    float(grad_scale), stream()), "capdec_ce_dlogits_bf16_inplace")
        return _zero_padded(logits)
    _mat(logits, "ce_dlogits_bf16.logits")
    ldd = _bf16_pitch(V)
    dst = torch.empty((rows, ldd), dtype=torch.bfloat16, device=logits.device)
    check(lib().capdec_ce_dlogits_bf16(ptr(logits), logits.stride(0), ptr(lse), ptr(targets), rows, V, float(eps),
                                        float(grad_scale), ptr(dst), ldd, stream()), "capdec_ce_dlogits_bf16")
    return _zero_padded(dst[:, :V])


def reduce_sum(x, scale=1.0):
    x = _req(x, "sum.x").contiguous()
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    l = lib()
    ws = _capi.workspace(l.capdec_reduce_workspace_bytes(), x.device)
    check(l.capdec_sum(ptr(x), x.numel(), float(scale), ptr(out), ptr(ws), ws.numel(), stream()), "capdec_sum")
    return out


def sumsq(g, out=None):
    if not g.is_contiguous():
        raise _capi.CapdecError("sumsq: tensor must be contiguous")
    out = torch.empty(1, dtype=torch.float32, device=g.device) if out is None else out
    l = lib()
    ws = _capi.workspace(l.capdec_reduce_workspace_bytes(), g.device)
    check(l.capdec_sumsq(ptr(_req(g, "sumsq.g")), g.numel(), ptr(out), ptr(ws), ws.numel(), stream()), "capdec_sumsq")
    return out


def clip_scale(g, gnorm_sq, max_norm):
    check(lib().capdec_clip_scale(ptr(g), g.numel(), ptr(gnorm_sq), float(max_norm), stream()), "capdec_clip_scale")


def adamw(p, g, m, v, step, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, gnorm_sq=None, max_norm=1.0):
    for t in (p, g, m, v):
        if not t.is_contiguous() or t.numel() != p.numel():
            raise _capi.CapdecError("adamw: flat contiguous blocks of equal size required")
    check(lib().capdec_adamw(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), float(lr), float(betas[0]), float(betas[1]),
                             float(eps), float(weight_decay), int(step), ptr(gnorm_sq), float(max_norm), stream()),
          "capdec_adamw")


def pe_add(x, pe, div, mod, t0=0, drop_p=0.0, seed=0):
    x = _req(x, "pe_add.x").contiguous()
    d = x.shape[-1]
    R = x.numel() // d
    out = torch.empty_like(x)
    check(lib().capdec_pe_add(ptr(x), ptr(pe), ptr(out), R, d, int(div), int(mod), int(t0), float(drop_p), int(seed),
                              stream()), "capdec_pe_add")
    return out


def act_dropout_bwd(dy, pre=None, drop_p=0.0, seed=0):
    dy = _req(dy, "act_dropout_bwd.dy").contiguous()
    out = torch.empty_like(dy)
    check(lib().capdec_act_dropout_bwd(ptr(dy), ptr(pre), ptr(out), dy.numel(), float(drop_p), int(seed), stream()),
          "capdec_act_dropout_bwd")
    return out
