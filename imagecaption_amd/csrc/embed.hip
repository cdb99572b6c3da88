// Token embedding + sinusoidal position add (+ dropout), and its deterministic backward.
// See include/capdec.h (capdec_embed_fwd / capdec_embed_bwd).
#include "common.h"

namespace capdec {

// Token ids outside [0, V) are clamped so that nothing is dereferenced out of bounds -- a guard, not semantics:
// the reference raises there (nn.Embedding / scatter index errors, e.g. a checkpoint paired with another
// vocabulary, ref main.py:113-137), so every clamped id is also COUNTED here and capdec_embed_bad_ids() hands the
// count to the caller, who raises (the kernels themselves never synchronise).
static __device__ unsigned int g_bad_ids;

// one wave per (b,t) row; float4 per lane
__global__ __launch_bounds__(256) void embed_fwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ E,
                                                         const float* __restrict__ pe, float* __restrict__ out, int BT,
                                                         int T, int d, int V, int t0, Dropout drop) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= BT) return;
    int64_t id = ids[row];
    if ((id < 0 || id >= V) && lane == 0) atomicAdd(&g_bad_ids, 1u);
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);
    const int t = t0 + row % T;
    const float4* e4 = reinterpret_cast<const float4*>(E + (size_t)id * d);
    const float4* p4 = reinterpret_cast<const float4*>(pe + (size_t)t * d);
    float4* o4 = reinterpret_cast<float4*>(out + (size_t)row * d);
    for (int q = lane; q < (d >> 2); q += 64) {
        const float4 a = e4[q], b = p4[q];
        float o[4] = {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w};
        if (drop.on()) {
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = drop.apply(o[j], (uint32_t)row * (uint32_t)d + (uint32_t)(q * 4 + j));
        }
        o4[q] = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// ---- backward: deterministic scatter-add ------------------------------------------------------
// Rows are grouped by token with a stable counting sort (rank = number of earlier rows with the
// same id), so dE[v] is the sum of its rows in row order whatever the launch geometry.  Work is
// spread over fixed 32-row chunks of the SORTED list (heavy tokens such as <PAD>, 80 % of the
// reference's positions, would otherwise serialise one workgroup): each wave reduces one chunk
// into per-(token, chunk) partial rows, then one workgroup per token adds that token's partials
// in chunk order.  No float atomics: the result is bitwise reproducible.
__device__ __forceinline__ int clamp_id(int64_t id, int V) { return (int)(id < 0 ? 0 : (id >= V ? V - 1 : id)); }

// rank[m] = number of rows before m with the same id.  Block (tb, tm), tb <= tm, adds what the 256 rows of tile tb
// contribute to the rows of tile tm (integer atomics: order-independent); the (tiles x tiles)/2 grid keeps every CU
// busy, where one block per row tile walking all earlier tiles left the launch on a handful of CUs.
__global__ __launch_bounds__(256) void embed_rank_kernel(const int64_t* __restrict__ ids, int* __restrict__ count,
                                                          int* __restrict__ rank, int BT, int V) {
    __shared__ int tile[256];
    const int tb = blockIdx.x, tm = blockIdx.y;
    if (tb > tm) return;
    const int m = tm * 256 + threadIdx.x;
    const int my = m < BT ? clamp_id(ids[m], V) : -1;
    const int j = tb * 256 + threadIdx.x;
    tile[threadIdx.x] = j < BT ? clamp_id(ids[j], V) : -2;
    __syncthreads();
    const int lim = tb < tm ? 256 : (int)threadIdx.x;  // only rows before m
    int r = 0;
#pragma unroll 8
    for (int k = 0; k < 256; ++k) r += (k < lim && tile[k] == my) ? 1 : 0;
    if (m < BT) {
        if (r) atomicAdd(&rank[m], r);
        if (tb == tm) atomicAdd(&count[my], 1);
    }
}

__global__ __launch_bounds__(1024) void embed_scan_kernel(const int* __restrict__ count, int* __restrict__ offset, int V) {
    __shared__ int sums[1024];
    const int per = (V + 1023) / 1024;
    const int lo = threadIdx.x * per, hi = min(V, lo + per);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += count[i];
    sums[threadIdx.x] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int v = threadIdx.x >= o ? sums[threadIdx.x - o] : 0;
        __syncthreads();
        sums[threadIdx.x] += v;
        __syncthreads();
    }
    int run = sums[threadIdx.x] - s;
    for (int i = lo; i < hi; ++i) {
        offset[i] = run;
        run += count[i];
    }
    if (threadIdx.x == 1023) offset[V] = sums[1023];
}

__global__ __launch_bounds__(256) void embed_order_kernel(const int64_t* __restrict__ ids, const int* __restrict__ offset,
                                                           const int* __restrict__ rank, int* __restrict__ order, int BT,
                                                           int V) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m < BT) order[offset[clamp_id(ids[m], V)] + rank[m]] = m;
}

// Chunk of the sorted list reduced by one wave.  Small chunks and one-wave workgroups: the launch has
// (BT / CHUNK) x (d / 256) independent waves (640 at the bench shape) instead of a few dozen workgroups
// walking their rows one after the other on a handful of CUs.
constexpr int EMB_CHUNK = 16;

// one wave per (EMB_CHUNK-position chunk of the sorted list, 256-column group); part[j] (row j of [BT,d]) receives
// the sum of the run of equal tokens that STARTS (within this chunk) at sorted position j.
__global__ __launch_bounds__(64) void embed_partial_kernel(const int64_t* __restrict__ ids, const int* __restrict__ order,
                                                            const float* __restrict__ dout, float* __restrict__ part,
                                                            int BT, int d, int V, Dropout drop) {
    const int lane = threadIdx.x;
    const int chunk = blockIdx.x;
    const int j0 = chunk * EMB_CHUNK;
    if (j0 >= BT) return;
    const int n = min(EMB_CHUNK, BT - j0);
    int my_row = 0, my_id = -1;
    if (lane < n) {
        my_row = order[j0 + lane];
        my_id = clamp_id(ids[my_row], V);
    }
    const int nq = d >> 2;
    {
        const int q = blockIdx.y * 64 + lane;  // this wave's 256 columns
        const bool act = q < nq;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        int head = 0;
        for (int g = 0; g < n; g += 4) {
            float4 v[4];
            int rows[4], vid[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                rows[u] = __shfl(my_row, min(g + u, n - 1), 64);
                vid[u] = __shfl(my_id, min(g + u, n - 1), 64);
                v[u] = (act && g + u < n) ? reinterpret_cast<const float4*>(dout + (size_t)rows[u] * d)[q]
                                          : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = g + u;
                if (j < n) {
                    float a[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
                    if (drop.on()) {
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            a[c] = drop.apply(a[c], (uint32_t)rows[u] * (uint32_t)d + (uint32_t)(q * 4 + c));
                    }
                    acc.x += a[0]; acc.y += a[1]; acc.z += a[2]; acc.w += a[3];
                    const int nxt = __shfl(my_id, min(j + 1, n - 1), 64);
                    if (j == n - 1 || nxt != vid[u]) {  // wave-uniform: end of a run
                        if (act) reinterpret_cast<float4*>(part + (size_t)(j0 + head) * d)[q] = acc;
                        acc = make_float4(0.f, 0.f, 0.f, 0.f);
                        head = j + 1;
                    }
                }
            }
        }
    }
}

// one workgroup per token: dE[v] = part[a] + part[next multiple of EMB_CHUNK after a] + ... (< b), in order;
// the list is cut into 4 contiguous shares (one per wave), combined in wave order.
__global__ __launch_bounds__(256) void embed_combine_kernel(const int* __restrict__ offset, const float* __restrict__ part,
                                                             float* __restrict__ dE, int d) {
    // 4 x d floats, sized at launch: a fixed 32 KiB block kept this (tiny) kernel from sharing a CU with the
    // weight-gradient products that run beside it on the other lane
    extern __shared__ __attribute__((aligned(16))) float sh_dyn[];
    const int v = blockIdx.x;
    const int a = offset[v], b = offset[v + 1];
    float* dst = dE + (size_t)v * d;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nq = d >> 2;
    if (a == b) {
        for (int q = threadIdx.x; q < nq; q += 256) reinterpret_cast<float4*>(dst)[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    const int first_b = (a / EMB_CHUNK + 1) * EMB_CHUNK;                       // first chunk boundary after a
    const int nh = 1 + (first_b < b ? (b - 1 - first_b) / EMB_CHUNK + 1 : 0);  // heads: a, first_b, first_b+CHUNK, ...
    const int share = (nh + 3) / 4;
    const int h0 = wave * share, h1 = min(nh, h0 + share);
    for (int q = lane; q < nq; q += 64) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        // eight partial rows are requested at a time and then added in head order (the order of the sum is the
        // same as a one-by-one walk; a heavy token such as <PAD> has a hundred heads per wave)
        for (int hb = h0; hb < h1; hb += 8) {
            float4 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int h = hb + u;
                const int pos = h == 0 ? a : first_b + (h - 1) * EMB_CHUNK;
                t[u] = h < h1 ? reinterpret_cast<const float4*>(part + (size_t)pos * d)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (hb + u < h1) {
                    acc.x += t[u].x; acc.y += t[u].y; acc.z += t[u].z; acc.w += t[u].w;
                }
            }
        }
        reinterpret_cast<float4*>(&sh_dyn[wave * d])[q] = acc;
    }
    __syncthreads();
    for (int q = threadIdx.x; q < nq; q += 256) {
        const float4 p0 = reinterpret_cast<const float4*>(&sh_dyn[0])[q], p1 = reinterpret_cast<const float4*>(&sh_dyn[d])[q];
        const float4 p2 = reinterpret_cast<const float4*>(&sh_dyn[2 * d])[q], p3 = reinterpret_cast<const float4*>(&sh_dyn[3 * d])[q];
        reinterpret_cast<float4*>(dst)[q] = make_float4(((p0.x + p1.x) + p2.x) + p3.x, ((p0.y + p1.y) + p2.y) + p3.y,
                                                        ((p0.z + p1.z) + p2.z) + p3.z, ((p0.w + p1.w) + p2.w) + p3.w);
    }
}

static size_t embed_ws_ints(int BT, int V) { return ((size_t)2 * V + 1 + 2 * (size_t)BT + 3) / 4 * 4; }

}  // namespace capdec

using namespace capdec;

extern "C" int capdec_embed_fwd(const int64_t* ids, const void* E, const void* pe, void* out, int B, int T, int d, int V,
                                int t0, float drop_p, uint64_t drop_seed, capdec_stream_t stream) {
    CAPDEC_REQUIRE(ids && E && pe && out, "capdec_embed_fwd: null pointer");
    CAPDEC_REQUIRE(B > 0 && T > 0 && V > 0 && d > 0 && d % 4 == 0 && t0 >= 0, "capdec_embed_fwd: bad shape");
    CAPDEC_REQUIRE(aligned16(E) && aligned16(pe) && aligned16(out), "capdec_embed_fwd: unaligned");
    CAPDEC_REQUIRE((uint64_t)B * T * d < (1ull << 32), "capdec_embed_fwd: B*T*d exceeds 2^32");
    const int BT = B * T;
    hipLaunchKernelGGL(embed_fwd_kernel, dim3((BT + 3) / 4), dim3(256), 0, (hipStream_t)stream, ids, (const float*)E,
                       (const float*)pe, (float*)out, BT, T, d, V, t0, make_dropout(drop_p, drop_seed));
    return check_launch("capdec_embed_fwd");
}

namespace capdec {
__global__ __launch_bounds__(256) void embed_zero_kernel(int* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0;
}
}  // namespace capdec

extern "C" int capdec_embed_bad_ids(unsigned int* count, int reset) {
    CAPDEC_REQUIRE(count, "capdec_embed_bad_ids: null pointer");
    if (hipMemcpyFromSymbol(count, HIP_SYMBOL(g_bad_ids), sizeof(unsigned int)) != hipSuccess) {
        set_error("capdec_embed_bad_ids: hipMemcpyFromSymbol failed");
        return CAPDEC_ERR_LAUNCH;
    }
    if (reset && *count) {
        const unsigned int zero = 0;
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_bad_ids), &zero, sizeof(zero)) != hipSuccess) return CAPDEC_ERR_LAUNCH;
    }
    return CAPDEC_OK;
}

extern "C" size_t capdec_embed_bwd_workspace_bytes(int B, int T, int d, int V) {
    if (B <= 0 || T <= 0 || d <= 0 || V <= 0) return 0;
    return (embed_ws_ints(B * T, V) + (size_t)B * T * d) * sizeof(float);
}

// The sort (count / offset / rank / order, the int head of the workspace) depends on the token ids only, so a
// trainer can run capdec_embed_bwd_plan early -- off the critical path, e.g. beside the forward pass -- and only
// capdec_embed_bwd_apply (partial sums + combine) at the end of backward.  capdec_embed_bwd = plan + apply.
static int embed_check(const char* what, const int64_t* ids, int B, int T, int d, int V, const void* workspace,
                       size_t workspace_bytes) {
    CAPDEC_REQUIRE(ids, "%s: null ids", what);
    CAPDEC_REQUIRE(B > 0 && T > 0 && V > 0 && d > 0 && d % 4 == 0 && d <= 2048, "%s: bad shape", what);
    CAPDEC_REQUIRE((uint64_t)B * T * d < (1ull << 32), "%s: B*T*d exceeds 2^32", what);
    CAPDEC_REQUIRE(aligned16(workspace), "%s: unaligned workspace", what);
    const size_t need = capdec_embed_bwd_workspace_bytes(B, T, d, V);
    if (!workspace || workspace_bytes < need) {
        set_error("%s: needs %zu workspace bytes, got %zu", what, need, workspace_bytes);
        return CAPDEC_ERR_WORKSPACE;
    }
    return CAPDEC_OK;
}

extern "C" int capdec_embed_bwd_plan(const int64_t* ids, int B, int T, int d, int V, void* workspace,
                                     size_t workspace_bytes, capdec_stream_t stream) {
    if (int rc = embed_check("capdec_embed_bwd_plan", ids, B, T, d, V, workspace, workspace_bytes)) return rc;
    const int BT = B * T;
    int* count = (int*)workspace;
    int* offset = count + V;
    int* rank = offset + V + 1;
    int* order = rank + BT;
    hipStream_t st = (hipStream_t)stream;
    // count [V], offset [V+1] and rank [BT] are contiguous: one fill zeroes the two accumulated arrays (a kernel of this
    // library rather than hipMemsetAsync: the step's trace then holds nothing but capdec kernels)
    {
        const size_t n = (size_t)2 * V + 1 + BT;
        size_t zb = (n + 255) / 256;
        if (zb > 1024) zb = 1024;
        hipLaunchKernelGGL(embed_zero_kernel, dim3((unsigned)zb), dim3(256), 0, st, count, n);
    }
    const int nb = (BT + 255) / 256;
    hipLaunchKernelGGL(embed_rank_kernel, dim3(nb, nb), dim3(256), 0, st, ids, count, rank, BT, V);
    hipLaunchKernelGGL(embed_scan_kernel, dim3(1), dim3(1024), 0, st, (const int*)count, offset, V);
    hipLaunchKernelGGL(embed_order_kernel, dim3(nb), dim3(256), 0, st, ids, (const int*)offset, (const int*)rank, order, BT, V);
    return check_launch("capdec_embed_bwd_plan");
}

extern "C" int capdec_embed_bwd_apply(const int64_t* ids, const void* dout, void* dE, int B, int T, int d, int V,
                                      float drop_p, uint64_t drop_seed, void* workspace, size_t workspace_bytes,
                                      capdec_stream_t stream) {
    if (int rc = embed_check("capdec_embed_bwd_apply", ids, B, T, d, V, workspace, workspace_bytes)) return rc;
    CAPDEC_REQUIRE(dout && dE && aligned16(dout) && aligned16(dE), "capdec_embed_bwd_apply: null or unaligned operand");
    const int BT = B * T;
    const int* offset = (const int*)workspace + V;
    const int* order = offset + V + 1 + BT;
    float* part = (float*)workspace + embed_ws_ints(BT, V);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(embed_partial_kernel, dim3((BT + EMB_CHUNK - 1) / EMB_CHUNK, (d / 4 + 63) / 64), dim3(64), 0, st, ids,
                       order, (const float*)dout, part, BT, d, V, make_dropout(drop_p, drop_seed));
    hipLaunchKernelGGL(embed_combine_kernel, dim3(V), dim3(256), 4 * (size_t)d * sizeof(float), st, offset, (const float*)part,
                       (float*)dE, d);
    return check_launch("capdec_embed_bwd_apply");
}

extern "C" int capdec_embed_bwd(const int64_t* ids, const void* dout, void* dE, int B, int T, int d, int V, float drop_p,
                                uint64_t drop_seed, void* workspace, size_t workspace_bytes, capdec_stream_t stream) {
    if (int rc = capdec_embed_bwd_plan(ids, B, T, d, V, workspace, workspace_bytes, stream)) return rc;
    return capdec_embed_bwd_apply(ids, dout, dE, B, T, d, V, drop_p, drop_seed, workspace, workspace_bytes, stream);
}
