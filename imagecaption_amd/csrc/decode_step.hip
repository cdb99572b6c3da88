// One decode step of the whole decoder stack behind ONE C-ABI call (capdec_decode_step): the per-token sequence of
// ~10 launches per layer that decode.py's _step issues op by op -- q / k|v projections (the latter straight into
// the cache), cached self-attention, out-projection + residual, LayerNorm, the cross-attention branch, LayerNorm,
// FFN, LayerNorm -- enqueued from C.  Same kernels, same plans, same order, same buffers' contents: bit-identical
// to the op-by-op path; what changes is the host cost per launch (a HIP launch instead of a Python -> ctypes round
// trip), which is what bounds small-batch decode (the reference's own sampler at B = 32: ~65 launches per token).
#include <math.h>
#include <string.h>

#include "common.h"
#include "gemm_common.h"

using namespace capdec;

static size_t step_floats(int R, int d, int dff) { return (size_t)R * ((size_t)7 * d + dff); }

extern "C" size_t capdec_decode_step_workspace_bytes(int R, int d, int dff) {
    if (R <= 0 || d <= 0 || dff <= 0) return 0;
    return step_floats(R, d, dff) * sizeof(float);
}

extern "C" int capdec_decode_step(const capdec_decode_layer* layers, int L, void* x, const int* anc, int R, int H, int d,
                                  int dff, int Tmax, int t, int W, int S, int dtype, void* workspace,
                                  size_t workspace_bytes, void* gemm_workspace, size_t gemm_workspace_bytes,
                                  capdec_stream_t stream) {
    CAPDEC_REQUIRE(layers && L > 0 && x && anc, "capdec_decode_step: null pointer");
    // every matrix of this call is fp32 (activations, weights, cache): the bf16-STORAGE type would make capdec_gemm read
    // those very pointers as bf16 matrices.  Callers in that mode keep the op-by-op path, whose products cast their
    // operands (decode.py CaptionDecoder._step).
    CAPDEC_REQUIRE(dtype == CAPDEC_F32 || dtype == CAPDEC_BF16 || dtype == CAPDEC_F32X3,
                   "capdec_decode_step: dtype %d -- all operands of this call are fp32 matrices; CAPDEC_BF16S (bf16 "
                   "storage) is not accepted here", dtype);
    CAPDEC_REQUIRE(R > 0 && H > 0 && d > 0 && d % H == 0 && d % 4 == 0 && dff > 0 && Tmax > 0 && t >= 0 && t < Tmax && W > 0 &&
                       S > 0, "capdec_decode_step: bad geometry R=%d H=%d d=%d dff=%d Tmax=%d t=%d W=%d S=%d", R, H, d, dff,
                   Tmax, t, W, S);
    if (!workspace || workspace_bytes < step_floats(R, d, dff) * sizeof(float) || !aligned16(workspace)) {
        set_error("capdec_decode_step: needs %zu workspace bytes (16-byte aligned), got %zu",
                  step_floats(R, d, dff) * sizeof(float), workspace_bytes);
        return CAPDEC_ERR_WORKSPACE;
    }
    float* q = (float*)workspace;
    float* a = q + (size_t)R * d;
    float* s1 = a + (size_t)R * d;
    float* x1 = s1 + (size_t)R * d;
    float* s2 = x1 + (size_t)R * d;
    float* x2 = s2 + (size_t)R * d;
    float* s3 = x2 + (size_t)R * d;
    float* h = s3 + (size_t)R * d;
    float* xf = (float*)x;
    const int dh = d / H;
    const float scale = 1.0f / sqrtf((float)dh);
    const float eps = 1e-5f;
    hipStream_t st = (hipStream_t)stream;

    auto linear = [&](const float* in, int K, const void* w, const void* b, int N, float* out, int ldc, const float* addend,
                      int act) -> int {
        capdec_epilogue ep;
        memset(&ep, 0, sizeof(ep));
        ep.bias = b;
        ep.addend = addend;
        ep.ld_addend = addend ? d : 0;
        ep.act = act;
        return capdec_gemm(0, 1, R, N, K, in, K, w, K, out, ldc, &ep, dtype, gemm_workspace, gemm_workspace_bytes, stream);
    };
#define STEP(call) do { const int rc_ = (call); if (rc_) return rc_; } while (0)
    for (int li = 0; li < L; ++li) {
        const capdec_decode_layer& P = layers[li];
        CAPDEC_REQUIRE(P.w_in && P.b_in && P.w_out && P.b_out && P.ln1_w && P.ln1_b && P.ln2_w && P.ln2_b && P.w1 && P.b1 &&
                           P.w2 && P.b2 && P.ln3_w && P.ln3_b && P.cache && (S == 1 ? P.cross_out != nullptr
                                                                                   : (P.cross_kv && P.wc_in && P.bc_in &&
                                                                                      P.wc_out && P.bc_out)),
                       "capdec_decode_step: layer %d has a null pointer", li);
        const float* w_in = (const float*)P.w_in;
        const float* b_in = (const float*)P.b_in;
        float* cache = (float*)P.cache;
        // K | V of the new token go straight into the cache: position t of every row (row stride Tmax * 2d)
        float* kv_t = cache + (size_t)t * 2 * d;
        GemmParams qkv;
        memset(&qkv, 0, sizeof(qkv));
        qkv.A = xf; qkv.B = w_in; qkv.C = q; qkv.M = R; qkv.N = 3 * d; qkv.K = d; qkv.lda = d; qkv.ldb = d; qkv.ldc = d;
        qkv.bias = b_in; qkv.drop = make_dropout(0.f, 0);
        qkv.C2 = kv_t; qkv.ldc2 = Tmax * 2 * d; qkv.n_split = d;
        // (a plan pinned for either of the two products sends the op-by-op path to the tiled kernel: follow it, the
        // contract is bit-identity with that path)
        if (gemm_skinny_ok(qkv, true, true, dtype) && !gemm_plan_forced(0, 1, R, d, d, dtype) &&
            !gemm_plan_forced(0, 1, R, 2 * d, d, dtype)) {
            // few rows: q and k | v as ONE product over the packed in-projection, its columns split between the query
            // buffer and the cache (each output element is computed exactly as by the two separate products)
            STEP(launch_gemm_skinny(qkv, gemm_workspace, gemm_workspace_bytes, st));
            STEP(check_launch("capdec_decode_step"));
        } else {
            STEP(linear(xf, d, w_in, b_in, d, q, d, nullptr, CAPDEC_ACT_NONE));
            STEP(linear(xf, d, w_in + (size_t)d * d, b_in + d, 2 * d, kv_t, Tmax * 2 * d, nullptr, CAPDEC_ACT_NONE));
        }
        STEP(capdec_decode_attn(q, d, cache, cache + d, 2 * d, a, d, anc, R, H, dh, Tmax, t, W, S, scale, stream));
        STEP(linear(a, d, P.w_out, P.b_out, d, s1, d, xf, CAPDEC_ACT_NONE));
        if (S == 1) {
            // LayerNorm, + the per-image constant of the one-key cross-attention branch, LayerNorm: one launch
            STEP(capdec_layernorm2_fwd(s1, P.ln1_w, P.ln1_b, P.cross_out, W, P.ln2_w, P.ln2_b, x2, R, d, eps, stream));
        } else {
            STEP(capdec_layernorm_fwd(s1, P.ln1_w, P.ln1_b, x1, nullptr, nullptr, R, d, eps, CAPDEC_ACT_NONE, 0.f, 0, stream));
            const float* wc = (const float*)P.wc_in;
            STEP(linear(x1, d, wc, P.bc_in, d, q, d, nullptr, CAPDEC_ACT_NONE));
            STEP(capdec_decode_attn(q, d, P.cross_kv, (const float*)P.cross_kv + d, 2 * d, a, d, nullptr, R, H, dh, Tmax, t, W,
                                    S, scale, stream));
            STEP(linear(a, d, P.wc_out, P.bc_out, d, s2, d, x1, CAPDEC_ACT_NONE));
            STEP(capdec_layernorm_fwd(s2, P.ln2_w, P.ln2_b, x2, nullptr, nullptr, R, d, eps, CAPDEC_ACT_NONE, 0.f, 0, stream));
        }
        STEP(linear(x2, d, P.w1, P.b1, dff, h, dff, nullptr, CAPDEC_ACT_GELU));
        {
            capdec_epilogue ep;
            memset(&ep, 0, sizeof(ep));
            ep.bias = P.b2;
            ep.addend = x2;
            ep.ld_addend = d;
            STEP(capdec_gemm(0, 1, R, d, dff, h, dff, P.w2, dff, s3, d, &ep, dtype, gemm_workspace, gemm_workspace_bytes,
                             stream));
        }
        STEP(capdec_layernorm_fwd(s3, P.ln3_w, P.ln3_b, xf, nullptr, nullptr, R, d, eps, CAPDEC_ACT_NONE, 0.f, 0, stream));
    }
#undef STEP
    return CAPDEC_OK;
}
