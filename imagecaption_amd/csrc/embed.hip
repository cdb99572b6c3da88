// Token embedding + sinusoidal position add (+ dropout), and its deterministic backward.
// See include/capdec.h (capdec_embed_fwd / capdec_embed_bwd).
#include "common.h"

namespace capdec {

// one wave per (b,t) row; float4 per lane; ids outside [0,V) are clamped (never dereferenced OOB).
__global__ __launch_bounds__(256) void embed_fwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ E,
                                                         const float* __restrict__ pe, float* __restrict__ out, int BT,
                                                         int T, int d, int V, int t0, Dropout drop) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= BT) return;
    int64_t id = ids[row];
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

// One block per vocabulary row v: scan all ids in index order, add matching rows.  The scan
// order is fixed, so the sum is bitwise reproducible; rows of absent tokens are written as zeros
// (no memset needed).  A presence bitmap built by the first kernel lets absent rows skip the scan.
__global__ __launch_bounds__(256) void embed_present_kernel(const int64_t* __restrict__ ids, int* __restrict__ present,
                                                             int BT, int V) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < BT) {
        int64_t id = ids[i];
        id = id < 0 ? 0 : (id >= V ? V - 1 : id);
        present[id] = 1;
    }
}

__global__ __launch_bounds__(256) void embed_bwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ dout,
                                                         float* __restrict__ dE, const int* __restrict__ present, int BT,
                                                         int d, int V, Dropout drop) {
    const int v = blockIdx.x;
    float* dst = dE + (size_t)v * d;
    if (!present[v]) {
        for (int c = threadIdx.x; c < d; c += 256) dst[c] = 0.f;
        return;
    }
    // columns handled by this thread: c = threadIdx.x + 256*i, i < 8  (d <= 2048)
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    for (int base = 0; base < BT; base += 256) {
        const int i = base + threadIdx.x;
        int64_t id = -1;
        if (i < BT) {
            id = ids[i];
            id = id < 0 ? 0 : (id >= V ? V - 1 : id);
        }
        // every wave needs the matches of all 4 waves of this chunk, in index order
        __shared__ unsigned long long masks[4];
        const unsigned long long m = __ballot(id == v);
        if ((threadIdx.x & 63) == 0) masks[threadIdx.x >> 6] = m;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            unsigned long long mm = masks[w];
            while (mm) {
                const int b = __ffsll((long long)mm) - 1;
                mm &= mm - 1;
                const int row = base + w * 64 + b;
                const float* src = dout + (size_t)row * d;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int c = threadIdx.x + 256 * k;
                    if (c < d) {
                        float g = src[c];
                        if (drop.on()) g = drop.apply(g, (uint32_t)row * (uint32_t)d + (uint32_t)c);
                        acc[k] += g;
                    }
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = threadIdx.x + 256 * k;
        if (c < d) dst[c] = acc[k];
    }
}

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

extern "C" size_t capdec_embed_bwd_workspace_bytes(int V) { return V > 0 ? (size_t)V * sizeof(int) : 0; }

extern "C" int capdec_embed_bwd(const int64_t* ids, const void* dout, void* dE, int B, int T, int d, int V, float drop_p,
                                uint64_t drop_seed, void* workspace, size_t workspace_bytes, capdec_stream_t stream) {
    CAPDEC_REQUIRE(ids && dout && dE, "capdec_embed_bwd: null pointer");
    CAPDEC_REQUIRE(B > 0 && T > 0 && V > 0 && d > 0 && d <= 2048, "capdec_embed_bwd: bad shape");
    CAPDEC_REQUIRE((uint64_t)B * T * d < (1ull << 32), "capdec_embed_bwd: B*T*d exceeds 2^32");
    if (!workspace || workspace_bytes < (size_t)V * sizeof(int)) {
        set_error("capdec_embed_bwd: needs %zu workspace bytes, got %zu", (size_t)V * sizeof(int), workspace_bytes);
        return CAPDEC_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const int BT = B * T;
    if (hipMemsetAsync(workspace, 0, (size_t)V * sizeof(int), st) != hipSuccess) {
        set_error("capdec_embed_bwd: memset failed");
        return CAPDEC_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(embed_present_kernel, dim3((BT + 255) / 256), dim3(256), 0, st, ids, (int*)workspace, BT, V);
    hipLaunchKernelGGL(embed_bwd_kernel, dim3(V), dim3(256), 0, st, ids, (const float*)dout, (float*)dE,
                       (const int*)workspace, BT, d, V, make_dropout(drop_p, drop_seed));
    return check_launch("capdec_embed_bwd");
}
