// Label-smoothed softmax cross-entropy (forward + backward in one pass over the logits),
// deterministic reductions, global-norm clipping and AdamW over a flat parameter block.
// All HBM-bound streaming kernels.  See include/capdec.h.
#include "common.h"

namespace capdec {

__device__ __forceinline__ float block_sum_256(float v, float* sh) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
__device__ __forceinline__ float block_max_256(float v, float* sh) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}

// One 256-thread block per row.  Pass 1: max and sum(z).  Pass 2: sum(exp(z - max)).
// Pass 3 (optional): dlogits.  The row (<= 120 KB) stays in L2 between passes.
// dlogits may alias logits (in-place mode), so neither pointer is __restrict__; everything the loss row needs
// from the logits (z_y included) is read BEFORE the last block-wide barrier, i.e. before any thread can store a
// gradient over it.
__global__ __launch_bounds__(256) void softmax_ce_kernel(const float* logits, int ldl,
                                                          const int64_t* __restrict__ targets,
                                                          float* __restrict__ loss_rows, float* dlogits, int lddl, int V,
                                                          float eps, float grad_scale) {
    __shared__ float sh[4];
    const int m = blockIdx.x;
    const float* z = logits + (size_t)m * ldl;
    float mx = -INFINITY, sz = 0.f;
    for (int v = threadIdx.x; v < V; v += 256) {
        const float x = z[v];
        mx = fmaxf(mx, x);
        sz += x;
    }
    mx = block_max_256(mx, sh);
    sz = block_sum_256(sz, sh);
    float se = 0.f;
    for (int v = threadIdx.x; v < V; v += 256) se += __expf(z[v] - mx);
    int64_t y = targets[m];
    y = y < 0 ? 0 : (y >= V ? V - 1 : y);
    const float zy = z[y];
    se = block_sum_256(se, sh);  // barrier: every read above is ordered before every store below
    const float lse = mx + __logf(se);
    if (threadIdx.x == 0 && loss_rows) loss_rows[m] = lse - (1.f - eps) * zy - (eps / (float)V) * sz;
    if (dlogits) {
        float* dz = dlogits + (size_t)m * lddl;
        const float u = eps / (float)V;
        for (int v = threadIdx.x; v < V; v += 256) {
            float g = __expf(z[v] - lse) - u;
            if (v == (int)y) g -= (1.f - eps);
            dz[v] = g * grad_scale;
        }
    }
}

// ---- fused form: the vocabulary product's epilogue left per-tile partials (gemm_common.h: ce_tile_partials) ----
// One wave per row: combine (max, sum exp, sum z) over the row's column tiles in a fixed order.
__global__ __launch_bounds__(256) void ce_finalize_kernel(const float4* __restrict__ part, int tiles,
                                                           const float* __restrict__ logits, int ldl,
                                                           const int64_t* __restrict__ targets, float* __restrict__ lse_out,
                                                           float* __restrict__ loss_rows, int M, int V, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float4* pr = part + (size_t)row * tiles;
    float mx = -INFINITY;
    for (int t = lane; t < tiles; t += 64) mx = fmaxf(mx, pr[t].x);
    mx = wave_max(mx);
    float se = 0.f, sz = 0.f, zy = 0.f;
    for (int t = lane; t < tiles; t += 64) {
        const float4 q = pr[t];
        se += q.y * __expf(q.x - mx);
        sz += q.z;
        zy += q.w;  // the target logit in exactly one tile, zero elsewhere (products run with ce_targets)
    }
    se = wave_sum(se);
    sz = wave_sum(sz);
    zy = wave_sum(zy);
    if (lane == 0) {
        const float lse = mx + __logf(se);
        int64_t y = targets[row];
        y = y < 0 ? 0 : (y >= V ? V - 1 : y);
        if (lse_out) lse_out[row] = lse;
        if (logits) zy = logits[(size_t)row * ldl + y];
        if (loss_rows) loss_rows[row] = lse - (1.f - eps) * zy - (eps / (float)V) * sz;
    }
}

// dlogits in place from the logits and the row lse (one read, one write)
__global__ __launch_bounds__(256) void ce_dlogits_kernel(float* z, int ldl, const float* __restrict__ lse_in,
                                                          const int64_t* __restrict__ targets, int V, float eps,
                                                          float grad_scale) {
    const int m = blockIdx.x;
    float* zr = z + (size_t)m * ldl;
    const float lse = lse_in[m], u = eps / (float)V;
    int64_t y = targets[m];
    y = y < 0 ? 0 : (y >= V ? V - 1 : y);
    for (int v = threadIdx.x; v < V; v += 256) {
        float g = __expf(zr[v] - lse) - u;
        if (v == (int)y) g -= (1.f - eps);
        zr[v] = g * grad_scale;
    }
}

// ---- deterministic reductions: 1024 fixed partials, then one block ---------------------------
constexpr int RED_PARTS = 1024;

template <bool SQ>
__global__ __launch_bounds__(256) void reduce_stage1_kernel(const float* __restrict__ x, size_t n, float* __restrict__ part) {
    __shared__ float sh[4];
    float s = 0.f;
    const size_t stride = (size_t)gridDim.x * 256;
    // 16 bytes per lane while whole float4s remain (x is 16-byte aligned: checked by the caller), then the tail;
    // which elements a thread owns depends on n and the grid only, so the sum is reproducible
    const size_t n4 = (reinterpret_cast<uintptr_t>(x) & 15u) == 0 ? n >> 2 : 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        s += SQ ? (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w) : (v.x + v.y) + (v.z + v.w);
    }
    for (size_t i = (n4 << 2) + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const float v = x[i];
        s += SQ ? v * v : v;
    }
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void reduce_stage2_kernel(const float* __restrict__ part, int nparts, float scale,
                                                             float* __restrict__ out) {
    __shared__ float sh[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) s += part[i];
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) out[0] = s * scale;
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                     float* __restrict__ v, size_t n, float lr, float b1, float b2,
                                                     float eps, float wd, float bc1, float bc2_sqrt,
                                                     const float* __restrict__ gnorm_sq, float max_norm) {
    float coef = 1.f;
    if (gnorm_sq) {
        const float c = max_norm / (sqrtf(gnorm_sq[0]) + 1e-6f);
        coef = c < 1.f ? c : 1.f;
    }
    const float step_size = lr / bc1;
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const float gi = g[i] * coef;
        float pi = p[i] * (1.f - lr * wd);
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        pi -= step_size * mi / (sqrtf(vi) / bc2_sqrt + eps);
        p[i] = pi;
    }
}

// in-place scale of a flat block by the clip coefficient (for callers that keep torch's optimizer)
__global__ __launch_bounds__(256) void clip_scale_kernel(float* __restrict__ g, size_t n, const float* __restrict__ gnorm_sq,
                                                          float max_norm) {
    const float c = max_norm / (sqrtf(gnorm_sq[0]) + 1e-6f);
    if (c >= 1.f) return;
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) g[i] *= c;
}

static int stream_blocks(size_t n) {
    size_t b = (n + 255) / 256;
    return (int)(b > 2048 ? 2048 : (b ? b : 1));
}

}  // namespace capdec

using namespace capdec;

extern "C" int capdec_softmax_ce(const void* logits, int ldl, const int64_t* targets, void* loss_rows, void* dlogits,
                                 int lddl, int M, int V, float eps, float grad_scale, capdec_stream_t stream) {
    CAPDEC_REQUIRE(logits && targets, "capdec_softmax_ce: null pointer");
    CAPDEC_REQUIRE(M > 0 && V > 0 && ldl >= V && (!dlogits || lddl >= V), "capdec_softmax_ce: bad shape M=%d V=%d", M, V);
    hipLaunchKernelGGL(softmax_ce_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, (const float*)logits, ldl, targets,
                       (float*)loss_rows, (float*)dlogits, lddl, V, eps, grad_scale);
    return check_launch("capdec_softmax_ce");
}

extern "C" int capdec_ce_finalize(const void* partials, int tiles, const void* logits, int ldl, const int64_t* targets,
                                  void* lse, void* loss_rows, int M, int V, float eps, capdec_stream_t stream) {
    CAPDEC_REQUIRE(partials && targets && (lse || loss_rows), "capdec_ce_finalize: null pointer");
    CAPDEC_REQUIRE(M > 0 && V > 0 && tiles > 0 && (!logits || ldl >= V) && aligned16(partials), "capdec_ce_finalize: bad shape M=%d V=%d "
                   "tiles=%d (partials must be 16-byte aligned)", M, V, tiles);
    hipLaunchKernelGGL(ce_finalize_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const float4*)partials,
                       tiles, (const float*)logits, ldl, targets, (float*)lse, (float*)loss_rows, M, V, eps);
    return check_launch("capdec_ce_finalize");
}

extern "C" int capdec_ce_dlogits(void* logits, int ldl, const void* lse, const int64_t* targets, int M, int V, float eps,
                                 float grad_scale, capdec_stream_t stream) {
    CAPDEC_REQUIRE(logits && lse && targets && M > 0 && V > 0 && ldl >= V, "capdec_ce_dlogits: bad arguments");
    hipLaunchKernelGGL(ce_dlogits_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, (float*)logits, ldl, (const float*)lse,
                       targets, V, eps, grad_scale);
    return check_launch("capdec_ce_dlogits");
}

extern "C" size_t capdec_reduce_workspace_bytes(void) { return RED_PARTS * sizeof(float); }

static int reduce_impl(bool sq, const void* x, size_t n, float scale, void* out, void* ws, size_t ws_bytes,
                       capdec_stream_t stream, const char* what) {
    CAPDEC_REQUIRE(x && out, "%s: null pointer", what);
    if (!ws || ws_bytes < RED_PARTS * sizeof(float)) {
        set_error("%s: needs %zu workspace bytes, got %zu", what, RED_PARTS * sizeof(float), ws_bytes);
        return CAPDEC_ERR_WORKSPACE;
    }
    size_t b = (n / 4 + 255) / 256;
    const int nb = (int)(b > RED_PARTS ? RED_PARTS : (b ? b : 1));
    hipStream_t st = (hipStream_t)stream;
    if (sq) hipLaunchKernelGGL((reduce_stage1_kernel<true>), dim3(nb), dim3(256), 0, st, (const float*)x, n, (float*)ws);
    else hipLaunchKernelGGL((reduce_stage1_kernel<false>), dim3(nb), dim3(256), 0, st, (const float*)x, n, (float*)ws);
    hipLaunchKernelGGL(reduce_stage2_kernel, dim3(1), dim3(256), 0, st, (const float*)ws, nb, scale, (float*)out);
    return check_launch(what);
}

extern "C" int capdec_sum(const void* x, size_t n, float scale, void* out, void* workspace, size_t workspace_bytes,
                          capdec_stream_t stream) {
    return reduce_impl(false, x, n, scale, out, workspace, workspace_bytes, stream, "capdec_sum");
}
extern "C" int capdec_sumsq(const void* g, size_t n, void* out, void* workspace, size_t workspace_bytes,
                            capdec_stream_t stream) {
    return reduce_impl(true, g, n, 1.0f, out, workspace, workspace_bytes, stream, "capdec_sumsq");
}

extern "C" int capdec_adamw(void* p, const void* g, void* m, void* v, size_t n, float lr, float beta1, float beta2,
                            float eps, float weight_decay, int step, const void* gnorm_sq, float max_norm,
                            capdec_stream_t stream) {
    CAPDEC_REQUIRE(p && g && m && v && n > 0 && step >= 1, "capdec_adamw: bad arguments");
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2_sqrt = sqrtf(1.f - powf(beta2, (float)step));
    hipLaunchKernelGGL(adamw_kernel, dim3(stream_blocks(n)), dim3(256), 0, (hipStream_t)stream, (float*)p,
                       (const float*)g, (float*)m, (float*)v, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2_sqrt,
                       (const float*)gnorm_sq, max_norm);
    return check_launch("capdec_adamw");
}

extern "C" int capdec_clip_scale(void* g, size_t n, const void* gnorm_sq, float max_norm, capdec_stream_t stream) {
    CAPDEC_REQUIRE(g && gnorm_sq && n > 0, "capdec_clip_scale: bad arguments");
    hipLaunchKernelGGL(clip_scale_kernel, dim3(stream_blocks(n)), dim3(256), 0, (hipStream_t)stream, (float*)g, n,
                       (const float*)gnorm_sq, max_norm);
    return check_launch("capdec_clip_scale");
}
