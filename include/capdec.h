/* capdec.h -- C ABI of libcapdec.so: the MI355X (gfx950) caption-decoder hot path.
 *
 * The reference (EricMoin/ImageCaption) has NO native / FFI interface: its hot path is Python on
 * torch.nn.  This header is therefore the boundary the build defines
 * underneath the reference's nn.Module surface; each entry point cites the reference-side
 * computation it replaces ("ref:" paths are relative to /root/reference, "torch/" means the
 * installed torch package, which is where the reference's arithmetic lives).
 *
 * Conventions
 *   - every pointer is a BORROWED DEVICE pointer; nothing is allocated, freed or synchronised
 *     across this ABI; scratch space is passed in as (workspace, workspace_bytes);
 *   - all matrices are row-major; "ld*" are row strides in ELEMENTS;
 *   - every call enqueues work on `stream` (a hipStream_t; NULL = the default stream) and returns
 *     0 (CAPDEC_OK) or a negative error code; capdec_last_error() describes the last failure of
 *     the calling thread;
 *   - dtype is CAPDEC_F32 unless an entry point says otherwise;
 *   - dropout is a counter-based mask keep(i) = hash(i, seed) >= p*2^32 that forward and backward
 *     regenerate from (seed, element index); p == 0 disables it (eval mode).
 */
#ifndef CAPDEC_H_
#define CAPDEC_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* capdec_stream_t; /* hipStream_t */

enum { CAPDEC_OK = 0, CAPDEC_ERR_ARG = -1, CAPDEC_ERR_LAUNCH = -2, CAPDEC_ERR_UNSUPPORTED = -3, CAPDEC_ERR_WORKSPACE = -4 };
/* compute type of capdec_gemm (operands and results are fp32 in memory in the first three):
 *   CAPDEC_F32   exact fp32 MFMA (v_mfma_f32_32x32x2_f32)
 *   CAPDEC_BF16  operands rounded to bf16, fp32 accumulate (mixed precision)
 *   CAPDEC_F32X3 fp32 operands split exactly into 3 bf16 terms, 6 bf16 MFMA partial products, fp32
 *                accumulate: fp32-level accuracy at 2.7x the fp32 matrix rate (csrc/gemm_f32x3.hip)
 *   CAPDEC_BF16S bf16 STORAGE: A and B point to bf16 matrices (made by capdec_cast_bf16 /
 *                capdec_ce_dlogits_bf16), either transpose; 16-byte aligned, lda / ldb multiples of 8 that cover
 *                the row rounded up to 8; bf16 MFMA, fp32 accumulate, fp32 epilogue and result
 *                (csrc/gemm_bf16s.hip).  ce_lse is not available (use capdec_ce_dlogits_bf16). */
enum { CAPDEC_F32 = 0, CAPDEC_BF16 = 1, CAPDEC_F32X3 = 2, CAPDEC_BF16S = 3 };
enum { CAPDEC_ACT_NONE = 0, CAPDEC_ACT_GELU = 1, CAPDEC_ACT_GELU_BWD = 2 };

int capdec_abi_version(void);
const char* capdec_last_error(void);

/* ---- GEMM with fused epilogue ---------------------------------------------------------------
 * C[M,N] = epilogue( opA(A)[M,K] * opB(B)[K,N] ),  opX = X or X^T (row-major storage).
 *   transA = 0: A stored [M,K] (lda >= K);  transA = 1: A stored [K,M] (lda >= M)
 *   transB = 0: B stored [K,N] (ldb >= N);  transB = 1: B stored [N,K] (ldb >= K)
 * epilogue, per element (m,n), in this order:
 *   v = acc + bias[n]; if (pre) pre[m,n] = v; GELU: v = gelu_erf(v); GELU_BWD: v *= gelu_erf'(aux[m,n]);
 *   dropout: v = keep(m*N+n) ? v/(1-p) : 0; v += addend[m,n]; C[m,n] = v.
 * colsum (transA = 1 only): colsum[m] = sum_k opA(A)[m,k]  (the bias gradient of a weight-gradient GEMM).
 * Replaces: torch/nn/functional.py linear() as called from ref models.py:37-44, 68-73 and
 * torch/nn/modules/transformer.py:1158-1199 (in/out projections, FFN), plus their autograd
 * backward (addmm / mm), bias+GELU+dropout+residual fused instead of separate passes.
 * F32 uses v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains).
 */
typedef struct capdec_epilogue {
    const void* bias;   /* [N] or NULL */
    const void* addend; /* [M,N], row stride ld_addend, or NULL */
    void* pre;          /* [M,N], row stride ldc, or NULL */
    const void* aux;    /* [M,N], row stride ldc; required for ACT_GELU_BWD */
    void* colsum;       /* [M] or NULL (transA = 1 only) */
    int ld_addend;
    int act;
    float drop_p;
    uint64_t drop_seed;
    /* label-smoothed cross-entropy fused into the vocabulary projection (ref train.py:41-50; all NULL = off):
     *   forward  (logits = h W^T + b, transA = 0, bias only): ce_partials [M, capdec_gemm_ce_tiles(M,N,K,dtype)]
     *            float4 receives, per row and column tile, (max, sum exp(z - max), sum z, t) of the logits the
     *            same epilogue writes to C; capdec_ce_finalize turns them into lse / loss rows.  With ce_targets set
     *            (forward), t is the fp32 logit of the row's target if its column lies in the tile, else 0 -- which
     *            is what lets the logits themselves leave as bf16 (c16 together with ce_partials REQUIRES
     *            ce_targets): the loss is then still the fp32 accumulators' (capdec_ce_finalize with logits = NULL).
     *   backward (the two products that consume dlogits: dW = dlogits^T h with transA = 1, dh = dlogits W with
     *            transA = 0; transB = 0, dtype CAPDEC_F32): with ce_lse set, A holds the LOGITS and the A loaders
     *            compute dlogits[r,v] = ce_grad_scale * (exp(z - ce_lse[r]) - ce_eps/V - (1 - ce_eps)*[v == ce_targets[r]])
     *            on their way into LDS (colsum, the bias gradient, sees the transformed values): the [B*T, V]
     *            block is written once by the forward product and only read afterwards. */
    void* ce_partials;
    const void* ce_lse;
    const int64_t* ce_targets;
    float ce_eps;
    float ce_grad_scale;
    /* bf16 result: if c16 != NULL the epilogue's result is written there as bf16 (round to nearest even, row stride
     * ldc16 in bf16 elements) INSTEAD of to C (which may then be NULL; ldc still strides pre / aux) -- for results whose
     * only readers take bf16 (the cross-attention K | V block of the bf16 storage path: capdec_attn_fwd_kv16; the
     * logits of the bf16 storage path, whose only reader is capdec_ce_dlogits_bf16_inplace). */
    void* c16;
    int ldc16;
    /* CAPDEC_BF16S operands whose reduction index is the contiguous one (A with transA = 0, B with transB = 1): a row
     * pitch that covers K rounded up to 64 is taken to mean that the elements K .. pitch-1 behind every row are ZERO
     * (capdec_cast_bf16 / capdec_ce_dlogits_bf16 write those zeros), and the LDS-DMA kernels -- which cannot mask a
     * load -- then run the reduction tail through them.  A caller whose pitch is that wide for another reason (a
     * column slice of a wider matrix, an uninitialised tail) sets bit 0 (A) / bit 1 (B): the product then takes the
     * register-staged kernel, which masks.  Irrelevant when K is a multiple of 64. */
    int bf16_unpadded;
} capdec_epilogue;

/* Split-K (chosen by the built-in cost model for small outputs with a long K, e.g. weight gradients) is combined
 * INSIDE the launch: every K-slice writes its partial tile to the workspace, takes a ticket, and the last-arriving
 * slice of a tile sums all slices in slice order (bitwise reproducible) and runs the epilogue -- no second
 * launch.  Workspace contract: capdec_gemm_workspace_bytes() bytes, 16-byte aligned; its first
 * CAPDEC_GEMM_TICKET_BYTES hold the tile tickets, which must be ZERO the first time a workspace is passed and are
 * left zero by every call (so: zero a workspace once when allocating it, give capdec_gemm a workspace no other
 * entry point writes to, and never share one workspace between GEMMs that may run concurrently on two streams). */
/* Few-row products (M <= 64, x[M,K] W[N,K]^T, fp32: the decode steps) run on a kernel of their own
 * (csrc/gemm_skinny.hip) that cuts reductions longer than 512 into K slices combined the same way; their workspace
 * need is part of capdec_gemm_workspace_bytes as well. */
#define CAPDEC_GEMM_TICKET_BYTES 16384
size_t capdec_gemm_workspace_bytes(int M, int N, int K);
/* Tuning hook (process-global, not thread-safe): force block tile bm x bn in {64,128}^2 (CAPDEC_BF16S: 64 x 64,
 * 128 x 128, 256 x 128 or 256 x 256; other compute types run a forced 256-row tile as 128 x 128) and a split-K
 * factor (and k-tile bk in {0 = default, 16, 32}) for every following capdec_gemm call; bm = 0
 * restores the built-in cost model.  CAPDEC_BF16S reads bk as a kernel choice: 128 x 128 with bk = 16 is the
 * register-staged kernel, 0 / 32 the LDS-DMA kernel with one / two image pairs; 256 x 256 with bk = 32 is the
 * eight-phase kernel (csrc/gemm_bf16s_p8.hip), with 0 the two-group "pair" kernel.
 * Environment (read per call, an A/B hook): CAPDEC_NO_VEC_EPI=1 makes every product store through the element form of
 * the fused epilogue instead of the vector form (same bits, slower; the eight-phase kernel then yields to the pair kernel). */
int capdec_gemm_tune(int bm, int bn, int split, int bk);
/* Tuned plans: record the (tile, split, k-tile) a caller MEASURED to be fastest for one product -- transposes,
 * shape and compute type -- on the device it runs on (the Python layer times the candidates once per shape,
 * outside any timed region: ops.tune_gemm_plans).  A recorded plan replaces the cost model's choice in every
 * later capdec_gemm / capdec_gemm_workspace_bytes / capdec_gemm_ce_tiles call of this process; bm = 0 erases one
 * entry, capdec_gemm_plan_clear all.  A plan changes speed and the rounding order of the K sum, nothing else. */
int capdec_gemm_plan_set(int transA, int transB, int M, int N, int K, int dtype, int bm, int bn, int split, int bk);
void capdec_gemm_plan_clear(void);
int capdec_gemm(int transA, int transB, int M, int N, int K, const void* A, int lda, const void* B, int ldb,
                void* C, int ldc, const capdec_epilogue* ep, int dtype, void* workspace, size_t workspace_bytes,
                capdec_stream_t stream);

/* capdec_gemm_grouped: up to CAPDEC_GEMM_GROUP_MAX INDEPENDENT exact-fp32 products of one layout (equal transA /
 * transB) in ONE launch, each with capdec_gemm's arguments and epilogue (bias, activation, dropout, residual, pre, aux,
 * colsum; no cross-entropy fusion, no bf16 result), operands 16-byte aligned with leading dimensions that are
 * multiples of 4.  For sets of small products that are ready at the same time -- the weight gradients of one decoder
 * layer (dW = dY^T X of torch/nn/modules/transformer.py:1143-1199's seven linear maps, 64 - 256 output tiles each): one
 * by one they need split-K with its slab round trips to occupy 256 CUs; together they are one chip-filling unsplit
 * grid.  bm x bn in {64,128}^2 is the block tile (0, 0 = 64 x 64).  Every product's K sum runs in one piece: results
 * are bit-identical to capdec_gemm under a forced unsplit plan of the same tile, deterministic, no workspace. */
#define CAPDEC_GEMM_GROUP_MAX 8
typedef struct capdec_gemm_desc {
    int transA, transB, M, N, K;
    const void* A; int lda;
    const void* B; int ldb;
    void* C; int ldc;
    const capdec_epilogue* ep; /* may be NULL */
} capdec_gemm_desc;
int capdec_gemm_grouped(const capdec_gemm_desc* descs, int n, int bm, int bn, capdec_stream_t stream);

/* ---- bf16 copies for CAPDEC_BF16S products ----------------------------------------------------------------
 * capdec_cast_bf16: dst[r][c] = bf16(src[r][c]) (round to nearest even) for an fp32 matrix [R, C] with row stride
 * lds; dst has row stride ldd (bf16 elements, a multiple of 8, >= C) and its padding columns C .. ldd-1 are
 * written as zeros.  One copy per tensor serves every product that reads it: capdec_gemm reads bf16 operands in
 * either orientation.  Replaces the .to(bfloat16) casts torch.autocast inserts around F.linear
 * (torch/amp/autocast_mode.py) if the reference's train.py:255-330 loop is run under bf16 autocast (BASELINE.json
 * configs[2]); the reference itself trains in fp32.
 * capdec_ce_dlogits_bf16: the label-smoothed cross-entropy gradient (ref train.py:41-50 backward) of a logits
 * block [M, V] written directly as its bf16 copy: dst[m][v] = bf16(grad_scale * (exp(z - lse[m]) - eps/V -
 * (1 - eps) * [v == targets[m]])), padding columns zero; lse from capdec_ce_finalize. */
int capdec_cast_bf16(const void* src, int lds, void* dst, int ldd, int R, int C, capdec_stream_t stream);
/* capdec_cast_bf16_multi: capdec_cast_bf16 of up to CAPDEC_CAST_MULTI_MAX matrices in ONE launch -- every weight
 * operand of a training step (the casts autocast would make per F.linear call, see above), whose addresses do not
 * change from step to step.  capdec_cast_bf16_multi_plan checks the matrices (src, dst, lds, ldd, R, C as for
 * capdec_cast_bf16; the other fields are filled in here), writes the table to table_dev (n * sizeof(capdec_cast_desc)
 * bytes of device memory, a blocking copy: call it once, not per step) and returns the launch size in *total_blocks;
 * capdec_cast_bf16_multi(table_dev, n, total_blocks) then makes all the copies, bit-identical to n single calls. */
#define CAPDEC_CAST_MULTI_MAX 256
typedef struct capdec_cast_desc {
    const void* src; /* fp32 [R, C], row stride lds */
    void* dst;       /* bf16 [R, ldd] */
    int lds, ldd, R, C;
    int vec, first_block, blocks, reserved; /* filled in by capdec_cast_bf16_multi_plan */
} capdec_cast_desc;
int capdec_cast_bf16_multi_plan(capdec_cast_desc* descs, int n, void* table_dev, int* total_blocks);
/* capdec_copy_f32_multi: the same table mechanism for plain fp32 copies (dst fp32 [R, ldd], ldd >= C, nothing is padded):
 * ONE launch gathers matrices that live apart into one block -- the cross-attention K | V projection weights and biases
 * of all decoder layers (rows d..3d of each layer's multihead_attn.in_proj_weight / _bias, ref models.py:53-65 through
 * torch/nn/modules/activation.py) stacked into the [L*2d, d] operand of the batched K | V product, in place of a
 * torch.cat per step. */
int capdec_copy_f32_multi_plan(capdec_cast_desc* descs, int n, void* table_dev, int* total_blocks);
int capdec_copy_f32_multi(const void* table_dev, int n, int total_blocks, capdec_stream_t stream);
/* tgt_in = captions[:, :-1], tgt_out = captions[:, 1:] (ref train.py:29-30) as two dense int64 [B, T1 - 1] blocks in one
 * launch; captions [B, T1] with row stride ld. */
int capdec_shift_captions(const int64_t* captions, int ld, int B, int T1, int64_t* tgt_in, int64_t* tgt_out,
                          capdec_stream_t stream);
int capdec_cast_bf16_multi(const void* table_dev, int n, int total_blocks, capdec_stream_t stream);
int capdec_ce_dlogits_bf16(const void* logits, int ldl, const void* lse, const int64_t* targets, int M, int V,
                           float eps, float grad_scale, void* dst, int ldd, capdec_stream_t stream);
/* The same gradient for a logits block that is ALREADY bf16 (written by the vocabulary product through
 * capdec_epilogue.c16, partials and target logits through ce_partials / ce_targets), in place: z16 [M, ld] bf16, ld a
 * multiple of 8 that covers V, 16-byte aligned; z16[m][v] <- bf16(grad_scale * (exp(float(z16[m][v]) - lse[m]) - eps/V -
 * (1 - eps) * [v == targets[m]])), padding columns V .. ld-1 zero.  Half the bytes of the fp32 logits on both sides of
 * the loss (what torch.autocast's bf16 F.linear output gives the reference's loss, train.py:41-50, under bf16 autocast);
 * the loss value itself comes from the fp32 partials. */
int capdec_ce_dlogits_bf16_inplace(void* z16, int ld, const void* lse, const int64_t* targets, int M, int V, float eps,
                                   float grad_scale, capdec_stream_t stream);

/* ---- LayerNorm (+ optional GELU + dropout) ----------------------------------------------------
 * fwd: y = dropout(act(LN(x))), LN eps as given; mean/rstd [M] saved for backward (may be NULL).
 * bwd: dx = LN'(dy_eff), dy_eff = dy * keep/(1-p) * act'(LN(x)); dgamma/dbeta [d] OVERWRITTEN.
 *      If dx_drop != NULL also writes dx_drop[i] = keep2(i) ? dx[i]/(1-p2) : 0 with (p2, seed2): the
 *      gradient entering the dropout that fed this norm's residual branch
 *      (torch/nn/modules/transformer.py:1144-1153 dropout1/2/3).
 * Replaces nn.LayerNorm at ref models.py:39, 43, 64 and torch/nn/modules/transformer.py:1076-1078.
 * d must be a multiple of 4 and <= 2048.
 */
size_t capdec_layernorm_bwd_workspace_bytes(int M, int d);
int capdec_layernorm_fwd(const void* x, const void* gamma, const void* beta, void* y, void* mean, void* rstd,
                         int M, int d, float eps, int act, float drop_p, uint64_t drop_seed, capdec_stream_t stream);
/* y = LN(gamma2, beta2)( add[row / W, :] + LN(gamma1, beta1)(x) ) in one launch: the two norms around a cross-attention
 * branch whose output is a per-image constant (memory of one position, decode steps; capdec_decode_step uses it).
 * Bit-identical to capdec_layernorm_fwd, a broadcast add and capdec_layernorm_fwd issued one after the other. */
int capdec_layernorm2_fwd(const void* x, const void* gamma1, const void* beta1, const void* add, int W, const void* gamma2,
                          const void* beta2, void* y, int M, int d, float eps, capdec_stream_t stream);
int capdec_layernorm_bwd(const void* dy, const void* x, const void* gamma, const void* beta, const void* mean,
                         const void* rstd, void* dx, void* dx_drop, void* dgamma, void* dbeta, int M, int d,
                         int act, float drop_p, uint64_t drop_seed, float drop2_p, uint64_t drop2_seed,
                         void* workspace, size_t workspace_bytes, capdec_stream_t stream);

/* The two halves of capdec_layernorm_bwd as separate calls: _dx writes dx (+ dx_drop) and the column partials
 * (capdec_layernorm_bwd_workspace_bytes(M, d) bytes, caller-owned), _dparams turns those partials into dgamma /
 * dbeta later -- on another stream if the caller likes: they are weight gradients, off the backward critical path. */
int capdec_layernorm_bwd_dx(const void* dy, const void* x, const void* gamma, const void* beta, const void* mean,
                            const void* rstd, void* dx, void* dx_drop, void* partials, size_t partials_bytes, int M,
                            int d, int act, float drop_p, uint64_t drop_seed, float drop2_p, uint64_t drop2_seed,
                            capdec_stream_t stream);
int capdec_layernorm_bwd_dparams(const void* partials, void* dgamma, void* dbeta, int M, int d, capdec_stream_t stream);

/* The same kernels with a bf16 output, for the CAPDEC_BF16S storage path (every LayerNorm result of the decoder, and
 * every dx_drop, is the operand of a matrix product that reads bf16):
 *   _fwd_y16   also writes y as a bf16 matrix y16 (row stride ldy16 in bf16 elements, a multiple of 8, >= d; padding
 *              columns zero: capdec_cast_bf16's layout) next to the fp32 y the residual path needs;
 *   _bwd_dx16  is capdec_layernorm_bwd_dx with dx_drop written as bf16 ONLY (dx_drop16, row stride lddd; with
 *              drop2_p == 0 it is the bf16 copy of dx).  Follow it with capdec_layernorm_bwd_dparams as usual.
 * Values are the fp32 results rounded to nearest even -- what capdec_cast_bf16 would make of them, minus the pass. */
int capdec_layernorm_fwd_y16(const void* x, const void* gamma, const void* beta, void* y, void* y16, int ldy16, void* mean,
                             void* rstd, int M, int d, float eps, int act, float drop_p, uint64_t drop_seed,
                             capdec_stream_t stream);
int capdec_layernorm_bwd_dx16(const void* dy, const void* x, const void* gamma, const void* beta, const void* mean,
                              const void* rstd, void* dx, void* dx_drop16, int lddd, void* partials, size_t partials_bytes,
                              int M, int d, int act, float drop_p, uint64_t drop_seed, float drop2_p, uint64_t drop2_seed,
                              capdec_stream_t stream);

/* capdec_layernorm_bwd_noalloc: capdec_layernorm_bwd for callers that CANNOT allocate or bring a workspace --
 * code under HIP graph capture, tight serving / fine-tuning loops, harnesses that forbid hipMalloc.  It has NO
 * counterpart in the reference (which never manages memory itself); same arguments minus the workspace, same
 * kernels, same bits.  The per-workgroup column partials (dgamma | dbeta, ln_blocks(M) x 2 x d floats) go to a
 * library-owned static device array of CAPDEC_LN_NOALLOC_SCRATCH_BYTES per GPU (held until the process ends;
 * every C1-C3 shape fits: M any, d <= 1024).  Shapes that need more are rejected with CAPDEC_ERR_WORKSPACE.
 * The call allocates nothing (no hipMalloc / hipFree; the array's per-device address is looked up on the
 * first call on a device).  Every call on one GPU shares that GPU's array (it is fully overwritten before it is
 * read, nothing has to be cleared): calls that overlap in time on the same GPU -- two streams, two threads --
 * race; serialise them on one stream. */
#define CAPDEC_LN_NOALLOC_SCRATCH_BYTES (4u << 20)
size_t capdec_layernorm_bwd_noalloc_max_bytes(void);
int capdec_layernorm_bwd_noalloc(const void* dy, const void* x, const void* gamma, const void* beta, const void* mean,
                                 const void* rstd, void* dx, void* dx_drop, void* dgamma, void* dbeta, int M, int d,
                                 int act, float drop_p, uint64_t drop_seed, float drop2_p, uint64_t drop2_seed,
                                 capdec_stream_t stream);

/* ---- token embedding + sinusoidal positions -----------------------------------------------------
 * fwd: out[b,t,:] = dropout(E[ids[b,t],:] + pe[t0+t,:]);  ids int64 [B,T]; pe [>=t0+T, d].
 * bwd: dE[v,:] = sum_{(b,t): ids=v} keep/(1-p) * dout[b,t,:]   (dE fully OVERWRITTEN, deterministic order).
 * Replaces ref models.py:109-114 (Embedding, transpose, PositionalEncoding.forward models.py:17-19).
 */
int capdec_embed_fwd(const int64_t* ids, const void* E, const void* pe, void* out, int B, int T, int d, int V,
                     int t0, float drop_p, uint64_t drop_seed, capdec_stream_t stream);
/* Token ids outside [0, V) are clamped by the kernels (an out-of-bounds guard) and counted: *count receives the
 * number of clamped ids that capdec_embed_fwd launches on the current device have seen since the last reset (this
 * call synchronises the device).  The reference raises an index error in that case; so does the Python layer
 * (CaptionTrainer.check_ids / ops.check_ids). */
int capdec_embed_bad_ids(unsigned int* count, int reset);
size_t capdec_embed_bwd_workspace_bytes(int B, int T, int d, int V);
int capdec_embed_bwd(const int64_t* ids, const void* dout, void* dE, int B, int T, int d, int V, float drop_p,
                     uint64_t drop_seed, void* workspace, size_t workspace_bytes, capdec_stream_t stream);
/* capdec_embed_bwd = plan + apply.  The plan (the stable sort of the rows by token id, kept in the int head of the
 * workspace) depends on ids only: run it early, off the critical path; apply needs the SAME workspace untouched. */
int capdec_embed_bwd_plan(const int64_t* ids, int B, int T, int d, int V, void* workspace, size_t workspace_bytes,
                          capdec_stream_t stream);
int capdec_embed_bwd_apply(const int64_t* ids, const void* dout, void* dE, int B, int T, int d, int V, float drop_p,
                           uint64_t drop_seed, void* workspace, size_t workspace_bytes, capdec_stream_t stream);

/* ---- fused attention (scores, softmax, context) --------------------------------------------------
 * o[b,tq,h,:] = sum_tk dropout(softmax_tk(scale * q[b,tq,h,:].k[b,tk,h,:] + causal)) v[b,tk,h,:]
 * Element (b,t,h,j) of q/k/v/o lives at ptr[(b*T + t)*ld + h*dh + j]: pass slices of the packed
 * projection buffers directly.  lse [B,H,Tq] = log-sum-exp of the scaled scores (saved for bwd).
 * causal: key tk is visible to query tq iff tk <= tq + (Tk - Tq).  dh in {32, 64, 128}.
 * bwd: dq, dk, dv from do, recomputing probabilities from lse (flash-style, no [Tq,Tk] tensor);
 *      delta [B,H,Tq] is scratch it fills (rowsum(do*o)).
 * Tk == 1 (the reference's memory of one ViT vector, ref models.py:104): the softmax is exactly 1, so both
 *      directions are streaming kernels: o = dropmask * v; dq = dk = 0 exactly (as torch computes them), dv =
 *      sum_t dropmask * do.  There dq may be NULL (not written), and lse / delta are not needed by bwd
 *      (lse, if given to fwd, still receives the scaled score; delta is left untouched).
 * Replaces torch/nn/functional.py:6386-6390, 6613-6631 (scaled_dot_product_attention inside
 * multi_head_attention_forward) for both the causal self-attention and the cross-attention over
 * the image memory (torch/nn/modules/transformer.py:1158-1194).
 */
int capdec_attn_fwd(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* o, int ldo,
                    void* lse, int B, int H, int Tq, int Tk, int dh, int causal, float scale, float drop_p,
                    uint64_t drop_seed, capdec_stream_t stream);
int capdec_attn_bwd(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const void* o, int ldo,
                    const void* dout, int lddo, const void* lse, void* delta, void* dq, int lddq, void* dk, int lddk,
                    void* dv, int lddv, int B, int H, int Tq, int Tk, int dh, int causal, float scale, float drop_p,
                    uint64_t drop_seed, capdec_stream_t stream);
/* Kernel choice of capdec_attn_fwd / capdec_attn_bwd (process-global, not thread-safe; a tuning / test hook like
 * capdec_gemm_tune): 0 = default (the persistent DMA-ring kernels wherever they cover the call: head dims 64 / 128,
 * Tk > 1, fp32 K / V), 1 = the one-item-per-wave kernels everywhere.  Both forms give bit-identical results. */
int capdec_attn_tune(int mode);
/* Attention with K and V given as bf16 matrices (ldk / ldv in bf16 elements, multiples of 8; Tk > 1): they are
 * widened to fp32 on their way into LDS and everything else is capdec_attn_fwd / capdec_attn_bwd_dkv16 (dK / dV come
 * back as bf16 too).  For the bf16 storage path, whose K | V block exists only as bf16 (capdec_epilogue.c16): the
 * attention kernels read K and V three times per layer and step, so this halves their largest stream. */
int capdec_attn_fwd_kv16(const void* q, int ldq, const void* k16, int ldk, const void* v16, int ldv, void* o, int ldo,
                         void* lse, int B, int H, int Tq, int Tk, int dh, int causal, float scale, float drop_p,
                         uint64_t drop_seed, capdec_stream_t stream);
int capdec_attn_bwd_kv16(const void* q, int ldq, const void* k16, int ldk, const void* v16, int ldv, const void* o, int ldo,
                         const void* dout, int lddo, const void* lse, void* delta, void* dq, int lddq, void* dk16,
                         int lddk, void* dv16, int lddv, int B, int H, int Tq, int Tk, int dh, int causal, float scale,
                         float drop_p, uint64_t drop_seed, capdec_stream_t stream);
/* capdec_attn_bwd with dK / dV written as bf16 matrices (lddk / lddv in bf16 elements, multiples of 8; Tk > 1): for
 * callers whose only consumers of dK / dV are CAPDEC_BF16S products (the cross-attention K/V projection's weight
 * and input gradients) -- saves the fp32 round trip and the cast pass of the step's largest gradient block.
 * Same computation as capdec_attn_bwd; the stored values are its fp32 results rounded to nearest even. */
int capdec_attn_bwd_dkv16(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const void* o, int ldo,
                          const void* dout, int lddo, const void* lse, void* delta, void* dq, int lddq, void* dk16,
                          int lddk, void* dv16, int lddv, int B, int H, int Tq, int Tk, int dh, int causal, float scale,
                          float drop_p, uint64_t drop_seed, capdec_stream_t stream);
/* The same attention with EVERY matrix bf16 (q, k, v, o, dout, dq, dk, dv; all ld in bf16 elements, multiples of 8;
 * each matrix under 2^31 elements; dh in {64, 128}; any Tk >= 1), lse / delta fp32: bf16 matrix products with fp32
 * accumulation (v_mfma_f32_32x32x16_bf16), scores / softmax / statistics in fp32, probabilities rounded to bf16 for
 * the context product.  For the CAPDEC_BF16S storage path, where the projections around attention read and write
 * bf16 anyway (capdec_epilogue.c16): the walked K / V tiles stream global -> LDS by DMA, half the bytes of the fp32
 * kernels and none of their conversions.  Same element indexing, causal rule, dropout element index (so the same
 * masks for one seed) and replaced reference lines as capdec_attn_fwd / capdec_attn_bwd; lse may be NULL in fwd. */
int capdec_attn_fwd_bf16(const void* q16, int ldq, const void* k16, int ldk, const void* v16, int ldv, void* o16, int ldo,
                         void* lse, int B, int H, int Tq, int Tk, int dh, int causal, float scale, float drop_p,
                         uint64_t drop_seed, capdec_stream_t stream);
int capdec_attn_bwd_bf16(const void* q16, int ldq, const void* k16, int ldk, const void* v16, int ldv, const void* o16,
                         int ldo, const void* do16, int lddo, const void* lse, void* delta, void* dq16, int lddq,
                         void* dk16, int lddk, void* dv16, int lddv, int B, int H, int Tq, int Tk, int dh, int causal,
                         float scale, float drop_p, uint64_t drop_seed, capdec_stream_t stream);

/* ---- label-smoothed softmax cross-entropy, forward + backward in one pass -------------------------
 * Per row m of logits [M,V] (row stride ldl) with target y = targets[m] (int64):
 *   loss_rows[m] = lse(z) - (1-eps)*z_y - (eps/V)*sum_v z_v
 *   dlogits[m,v] = grad_scale * (softmax(z)_v - (1-eps)*[v==y] - eps/V)     (may alias logits; NULL = skip)
 * Replaces ref train.py:41-50 (dense one-hot, log_softmax, product, mean over ALL B*T rows, PAD
 * included) and its autograd backward; the mean is capdec_mean() over loss_rows.
 */
int capdec_softmax_ce(const void* logits, int ldl, const int64_t* targets, void* loss_rows, void* dlogits, int lddl,
                      int M, int V, float eps, float grad_scale, capdec_stream_t stream);
/* The fused form of the same loss (capdec_epilogue.ce_*): the vocabulary product's epilogue leaves per-tile softmax
 * partials, capdec_ce_finalize combines them per row -- lse[m] and loss_rows[m] as above (z_y is read from the
 * fp32 logits; with logits = NULL it is the sum of the partials' fourth components, i.e. the product must have run
 * with ce_targets) -- and the backward products read the logits through the gradient transform.  capdec_gemm_ce_tiles: the
 * number of column tiles (second dimension of the partials) capdec_gemm will use for this shape and dtype.
 * capdec_ce_dlogits: the same transform as a pass in place, for compute types whose loaders do not carry it
 * (CAPDEC_BF16 / CAPDEC_F32X3) and for callers that want dlogits materialised. */
int capdec_gemm_ce_tiles(int M, int N, int K, int dtype);
int capdec_ce_finalize(const void* partials, int tiles, const void* logits, int ldl, const int64_t* targets, void* lse,
                       void* loss_rows, int M, int V, float eps, capdec_stream_t stream);
int capdec_ce_dlogits(void* logits, int ldl, const void* lse, const int64_t* targets, int M, int V, float eps,
                      float grad_scale, capdec_stream_t stream);
/* out[0] = scale * sum(x[0..n)) with a fixed (deterministic) reduction tree; workspace of
 * capdec_reduce_workspace_bytes() bytes. */
size_t capdec_reduce_workspace_bytes(void);
int capdec_sum(const void* x, size_t n, float scale, void* out, void* workspace, size_t workspace_bytes,
               capdec_stream_t stream);

/* ---- gradient clipping + AdamW over a flat parameter block ---------------------------------------
 * capdec_sumsq: out[0] = sum(g^2) (deterministic).  capdec_adamw: one fused pass
 *   coef = min(1, max_norm / (sqrt(*gnorm_sq) + 1e-6))  (1 if gnorm_sq == NULL)   ref train.py:56
 *   g = coef*g; p *= 1 - lr*wd; m = b1*m + (1-b1)*g; v = b2*v + (1-b2)*g*g;
 *   p -= lr/(1-b1^step) * m / (sqrt(v)/sqrt(1-b2^step) + eps)                     ref main.py:95
 * (torch.optim.AdamW single-tensor formula).  No host synchronisation: the norm stays on device.
 */
int capdec_sumsq(const void* g, size_t n, void* out, void* workspace, size_t workspace_bytes, capdec_stream_t stream);
/* g *= min(1, max_norm/(sqrt(*gnorm_sq)+1e-6)) in place, for callers that keep torch.optim (ref train.py:56-59). */
int capdec_clip_scale(void* g, size_t n, const void* gnorm_sq, float max_norm, capdec_stream_t stream);
int capdec_adamw(void* p, const void* g, void* m, void* v, size_t n, float lr, float beta1, float beta2, float eps,
                 float weight_decay, int step, const void* gnorm_sq, float max_norm, capdec_stream_t stream);

/* ---- cross-attention branch over a memory of ONE position (S = 1, ref models.py:104) ------------------
 * Softmax over one key is 1: head h's context is v[b, h-slice], kept or dropped as a whole by the
 * attention-probability dropout, so the branch  addend + dropout_out(out_proj(context) + bias)
 * (torch/nn/modules/transformer.py:1177-1187 with F.multi_head_attention_forward, functional.py:6613-6642)
 * needs no [B*T, d] x [d, d] product:
 *   U[b*H+h, :] = Wo[:, h-slice] v[b, h-slice]                         (the caller's tiny GEMM, B*H rows)
 *   capdec_cross1_fwd: out[r,:] = addend[r,:] + drop_out(bias + sum_h keep_h(r)/(1-p_attn) * U[(r/T)*H+h, :])
 *   capdec_cross1_bwd: dU[b*H+h,:] = sum_t keep_h(b,t)/(1-p_attn) * g[b*T+t,:];  dbias_part[b,:] = sum_t g[b*T+t,:]
 * with g the gradient at the branch output AFTER the output dropout's mask.  Element indices of both
 * dropouts are those of capdec_attn_fwd (Tk = 1) and of capdec_gemm's epilogue; all rows dense (ld = d);
 * d % 64 == 0; H in {1,2,4,8,16} for the backward.  Fixed summation orders (bitwise reproducible). */
int capdec_cross1_fwd(const void* U, const void* bias, const void* addend, void* out, int B, int T, int H, int d,
                      float attn_p, uint64_t attn_seed, float out_p, uint64_t out_seed, capdec_stream_t stream);
int capdec_cross1_bwd(const void* g, void* dU, void* dbias_part, int B, int T, int H, int d, float attn_p,
                      uint64_t attn_seed, capdec_stream_t stream);
/* The two reshapes around the B*H-row products of that branch (same reference lines):
 *   capdec_cross1_spread: vblk[b*H+h, j] = v[b*ldv + j] inside head h's slice, 0 elsewhere -- the operand of the U product;
 *   capdec_cross1_gather: dk[b,:] = 0 (exactly, as torch computes it: the softmax over one key has no gradient),
 *                         dv[b,j] = dvblk[b*H + head(j), j], dbias[j] = sum_b dbias_part[b,j] (image order: reproducible).
 * vblk / dvblk / dbias_part dense (ld = d); v, dk, dv row-strided (ld % 4 == 0); head dim % 4 == 0. */
int capdec_cross1_spread(const void* v, int ldv, void* vblk, int B, int H, int d, capdec_stream_t stream);
int capdec_cross1_gather(const void* dvblk, const void* dbias_part, void* dk, int lddk, void* dv, int lddv, void* dbias,
                         int B, int H, int d, capdec_stream_t stream);

/* ---- KV-cached decode step: greedy / beam search (NEW capability: the reference only samples,
 * ref train.py:338-368, and re-runs the whole prefix every step, train.py:312-332) ----------------
 * capdec_decode_attn: single-query attention of every row r (R = images * W beams) against
 *   - self (anc != NULL): keys j = 0..t of the cache; position j of row r lives in cache row
 *     (slot*Tmax + j), slot = (j == t ? r : anc[r*Tmax + j]); k / v point at the K and V halves of the
 *     cache, row stride ldkv;
 *   - cross (anc == NULL): keys j = 0..S-1 of image r / W: row ((r/W)*S + j) of k / v.  With W > 1 (and R a
 *     multiple of W) the W beams of an image are served by one wave that reads the image's K / V once; a
 *     row's result has the same bits whatever W is.
 * capdec_greedy_select: seq[r, t] = next_tok[r] = done[r] ? PAD : argmax_v logits[r, v] (ties -> lowest id);
 *   done[r] |= (token == END); gap[r] = min(gap[r], top1 - top2) on live rows (gap may be NULL).
 * capdec_beam_select (the beam top-k / reorder kernel), per image n with W <= 8 beams: rank the W*V
 *   candidates score[w] + log_softmax(logits[w])[v] by (score desc, w*V+v asc); a finished beam
 *   proposes only (PAD, score unchanged).  Writes the W winners' score / finished flag / next token
 *   and REORDERS the hypothesis state by rewriting the int tables only: seq_out[row, 0..t] and the
 *   ancestry table anc_out[row, 0..t] (the K/V cache itself is never moved).  gap[n] = min margin
 *   between the last kept and first rejected candidate (may be NULL).
 */
int capdec_decode_attn(const void* q, int ldq, const void* k, const void* v, int ldkv, void* out, int ldo,
                       const int* anc, int R, int H, int dh, int Tmax, int t, int W, int S, float scale,
                       capdec_stream_t stream);
/* capdec_decode_step: ONE decode step (position t) of the whole decoder stack for R rows, enqueued by a single
 * call: per layer, q and k|v projections (k|v land in the layer's cache at position t), capdec_decode_attn over
 * the cache, out-projection + residual, LayerNorm, the cross-attention branch (S > 1: q projection,
 * capdec_decode_attn over cross_kv, out-projection + residual; S == 1: the per-image constant cross_out[r / W]
 * + residual), LayerNorm, FFN (GELU) + residual, LayerNorm -- the post-norm layer of
 * torch/nn/modules/transformer.py:1144-1153 as ref models.py:116-122 runs it, one token at a time.  x [R, d] holds
 * the layer input (embedding + positional encoding) on entry and the last layer's output on return (the
 * stack's final LayerNorm and the output head stay with the caller).  Exactly the launches, in the order, that
 * a caller issuing these entry points one by one would make (imagecaption_amd/decode.py _step): bit-identical
 * results; what it removes is the caller's per-launch host cost, which bounds small-batch decode (the reference's
 * sampler, ref train.py:289-412, at its own batch sizes).  Tmax / W are the cache geometry of capdec_decode_attn;
 * every pointer of a layer is a device pointer, all matrices dense (row stride = their width), fp32.
 * dtype: CAPDEC_F32 / CAPDEC_BF16 / CAPDEC_F32X3 (arithmetic of the products on those fp32 matrices); CAPDEC_BF16S is
 * refused with CAPDEC_ERR_ARG (its products read bf16 copies, which this call does not hold).
 * workspace: capdec_decode_step_workspace_bytes(R, d, dff) bytes, 16-byte aligned; gemm_workspace: a capdec_gemm
 * workspace (ticket words zeroed once), capdec_gemm_workspace_bytes of the largest product. */
typedef struct capdec_decode_layer {
    const void *w_in, *b_in;     /* self_attn.in_proj_weight [3d, d], in_proj_bias [3d] */
    const void *w_out, *b_out;   /* self_attn.out_proj */
    const void *ln1_w, *ln1_b;
    const void *wc_in, *bc_in;   /* multihead_attn.in_proj (only its q block is used; S > 1) */
    const void *wc_out, *bc_out; /* multihead_attn.out_proj (S > 1) */
    const void *ln2_w, *ln2_b;
    const void *w1, *b1;         /* linear1 [dff, d] */
    const void *w2, *b2;         /* linear2 [d, dff] */
    const void *ln3_w, *ln3_b;
    void* cache;                 /* [rows, Tmax, 2d] self-attention K | V cache of this layer */
    const void* cross_kv;        /* S > 1: [images * S, 2d] cross-attention K | V of the image memory */
    const void* cross_out;       /* S == 1: [R / W, d] out_proj(v) + bias of the one memory position */
} capdec_decode_layer;
size_t capdec_decode_step_workspace_bytes(int R, int d, int dff);
int capdec_decode_step(const capdec_decode_layer* layers, int L, void* x, const int* anc, int R, int H, int d, int dff,
                       int Tmax, int t, int W, int S, int dtype, void* workspace, size_t workspace_bytes,
                       void* gemm_workspace, size_t gemm_workspace_bytes, capdec_stream_t stream);
/* ---- one decoder layer of a TRAINING step per call ------------------------------------------------------------------
 * capdec_train_layer_fwd / _bwd enqueue, from ONE call each, every launch of a post-norm decoder layer
 * (torch/nn/modules/transformer.py:1143-1199 as built at ref models.py:53-65, teacher-forced as ref train.py:24-61 runs
 * it) and of its autograd backward: packed in-projection, causal self-attention, out-projection + dropout + residual,
 * LayerNorm, cross-attention over the image memory (S > 1 positions), LayerNorm, FFN (GELU), LayerNorm; backward in
 * reverse, with the layer's seven weight gradients and three LayerNorm parameter gradients handed to `lane_stream`
 * (ordered after their producers by events; the critical path never waits for them -- the CALLER joins the lane before
 * it reads the gradients).  Exactly the entry points above, with the arguments and in the per-stream order a caller
 * issuing them one by one would use (imagecaption_amd/engine.py layer_fwd / layer_bwd): bit-identical results.  What it
 * removes is the caller's per-launch host cost, which bounds the bf16-storage step.
 *   geometry: B images x T tokens, memory of S > 1 positions, H heads, model width d, FFN width dff; `dtype` is the
 *     compute type of every product (CAPDEC_BF16S: d, dff multiples of 64, head dim 64 or 128 -- the all-bf16
 *     attention kernels); drop_p the dropout probability of all six sites, whose seeds are
 *     splitmix64(dropout_key ^ (16 * layer + site) * 0xD6E8FEB86659FD93) (engine.DropoutPlan); group_wgrads: with
 *     CAPDEC_F32, the weight gradients go out as one capdec_gemm_grouped launch of tile group_bm x group_bn.
 *   layer: fp32 parameters and their gradients (dense, the reference's state_dict shapes; gradients are OVERWRITTEN),
 *     with CAPDEC_BF16S also zero-padded bf16 copies of the six weight matrices (wcq16: rows 0..d-1 of
 *     multihead_attn.in_proj_weight, the query block), refreshed by the caller before the step; `arena`:
 *     capdec_train_layer_arena_bytes(geom) bytes, 256-byte aligned, owned by this layer from its forward to the end of
 *     its backward's lane work.  capdec_train_layer_slot gives offset / size / row stride / element type of a tensor
 *     inside it (slot ids: CAPDEC_TL_*): the layer output x3 (+ its bf16 copy), the input gradient dx.
 *   fwd: x [B*T, d] fp32 (+ x16, its bf16 copy, with CAPDEC_BF16S); kv [B*S, 2d] = this layer's K | V projection of
 *     the memory (row stride ldkv; bf16 with CAPDEC_BF16S) from the caller's batched product.
 *   bwd: dy [B*T, d] fp32, the x / x16 / kv of the forward call, mem [B*S, d] (+ mem16), dkv [B*S, 2d] receiving
 *     dK | dV (fp32, or bf16 with CAPDEC_BF16S; the caller's batched d(memory) product reads it).  Each stream needs a
 *     capdec_gemm workspace of its own.  lane_stream == stream runs everything on one stream. */
typedef struct capdec_train_geom {
    int B, T, S, H, d, dff, causal, dtype;
    float drop_p;
    int group_wgrads, group_bm, group_bn;
    int wgrad_flush_mask; /* bit i: hand the queued weight gradients to the lane after the i-th one of the layer's backward
                             (order: linear2, linear1, cross out_proj, cross q, cross k|v, self out_proj, self in_proj; the
                             last one always hands over).  0 = the library's default (after linear1 and after cross k|v) */
} capdec_train_geom;
typedef struct capdec_train_layer {
    const void *w_in, *b_in, *w_out, *b_out, *ln1_w, *ln1_b, *wc_in, *bc_in, *wc_out, *bc_out, *ln2_w, *ln2_b, *w1, *b1, *w2,
        *b2, *ln3_w, *ln3_b;
    void *g_w_in, *g_b_in, *g_w_out, *g_b_out, *g_ln1_w, *g_ln1_b, *g_wc_in, *g_bc_in, *g_wc_out, *g_bc_out, *g_ln2_w,
        *g_ln2_b, *g_w1, *g_b1, *g_w2, *g_b2, *g_ln3_w, *g_ln3_b;
    const void *w_in16, *w_out16, *wcq16, *wc_out16, *w1_16, *w2_16; /* CAPDEC_BF16S only */
    int ld_w_in16, ld_w_out16, ld_wcq16, ld_wc_out16, ld_w1_16, ld_w2_16;
    void* arena;
    size_t arena_bytes;
} capdec_train_layer;
enum { CAPDEC_TL_X3 = 19, CAPDEC_TL_X3_16 = 20, CAPDEC_TL_DX = 37, CAPDEC_TL_SLOTS = 41 };
size_t capdec_train_layer_arena_bytes(const capdec_train_geom* geom);
int capdec_train_layer_slot(const capdec_train_geom* geom, int slot, size_t* offset, size_t* bytes, int* ld, int* is_bf16);
int capdec_train_layer_fwd(const capdec_train_layer* layer, const capdec_train_geom* geom, int layer_index,
                           uint64_t dropout_key, const void* x, const void* x16, int ldx16, const void* kv, int ldkv,
                           void* gemm_workspace, size_t gemm_workspace_bytes, capdec_stream_t stream);
int capdec_train_layer_bwd(const capdec_train_layer* layer, const capdec_train_geom* geom, int layer_index,
                           uint64_t dropout_key, const void* dy, const void* x, const void* x16, int ldx16, const void* mem,
                           const void* mem16, int ldmem16, const void* kv, int ldkv, void* dkv, int lddkv,
                           void* gemm_workspace, size_t gemm_workspace_bytes, void* lane_gemm_workspace,
                           size_t lane_gemm_workspace_bytes, capdec_stream_t stream, capdec_stream_t lane_stream);

int capdec_greedy_select(const void* logits, int ldl, int* done, int* seq, int64_t* next_tok, void* gap, int R, int V,
                         int Tmax, int t, capdec_stream_t stream);
int capdec_beam_select(const void* logits, int ldl, const void* score_in, void* score_out, const int* fin_in,
                       int* fin_out, const int* anc_in, int* anc_out, const int* seq_in, int* seq_out,
                       int64_t* next_tok, void* gap, int N, int W, int V, int Tmax, int t, capdec_stream_t stream);

/* capdec_sample_select: one step of the reference's sampling decode (ref train.py:335-383) for R rows,
 * without the reference's per-sample host syncs: step i writes seq[r, i+1] = next_tok[r] = END for the
 * last two steps (i >= max_len-3), otherwise a draw from the top-5 of logits/temperature after the
 * period (id 4) / END (id 2) biases of train.py:343-360, and updates words_since_period / sentences
 * (train.py:374-380).  The draw replaces torch.multinomial by inverse CDF on the counter-based uniform
 * (mix32(mix32(r + lo32(seed)) ^ hi32(seed) ^ i*0x9E3779B9) >> 8) / 2^24: same distribution.  cnt_end[i] /
 * cnt_max[i] count the rows that emitted END / reached max_sentences in step i (zero both arrays,
 * [max_len] ints, before step 0); once a step has all R rows in either, later calls do nothing -- the
 * reference's break at train.py:383. */
int capdec_sample_select(const void* logits, int ldl, int* words_since_period, int* sentences, int* seq,
                         int64_t* next_tok, int* cnt_end, int* cnt_max, int R, int V, int Tmax, int i, int max_len,
                         int max_sentences, int min_words, float temperature, uint64_t seed, capdec_stream_t stream);

/* ---- peak probes (measurement only, no reference counterpart): a register-resident MFMA loop
 * (dtype CAPDEC_F32: v_mfma_f32_32x32x2_f32, CAPDEC_BF16: v_mfma_f32_32x32x16_bf16; `blocks` workgroups of
 * 4 waves, 4*iters MFMAs per wave; *flops receives the FLOPs issued) and a streaming copy of `bytes`. */
int capdec_probe_mfma(int dtype, int blocks, int iters, void* out, double* flops, capdec_stream_t stream);
int capdec_probe_copy(const void* src, void* dst, size_t bytes, capdec_stream_t stream);

/* ---- elementwise helpers for the sub-module-at-a-time call path ---------------------------------
 * capdec_pe_add: out[r,:] = dropout(x[r,:] + pe[t0 + (r/div)%mod, :]) over x viewed as [R,d]
 *   (seq-first [T,B,d]: div=B, mod=T -- PositionalEncoding.forward, ref models.py:17-19).
 * capdec_act_dropout_bwd: out[i] = dy[i] * keep(i)/(1-p) * (pre ? gelu_erf'(pre[i]) : 1): the
 *   gradient through Dropout(GELU(.)) of ref models.py:40-41, 70-71 when a Linear is called alone.
 */
int capdec_pe_add(const void* x, const void* pe, void* out, int R, int d, int div, int mod, int t0, float drop_p,
                  uint64_t drop_seed, capdec_stream_t stream);
int capdec_act_dropout_bwd(const void* dy, const void* pre, void* out, size_t n, float drop_p, uint64_t drop_seed,
                           capdec_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CAPDEC_H_ */
