// Skinny fp32 product for decode steps: C[M, N] = epilogue(A[M, K] * B[N, K]^T) with M of a few dozen rows (one
// token per caption: the reference's sampler at its own batch sizes, ref train.py:289-412; greedy / beam steps).
// The tiled kernel of gemm_f32.hip needs 64-row tiles and in-launch split-K to find any parallelism in such a
// shape (8 column tiles x K slices, tickets, slab round trips: 13.6 us for 32 x 512 x 512); here one 1024-thread
// workgroup owns a 32 x 32 output block and its SIXTEEN WAVES SPLIT K among themselves (combined through 67.6 KB of
// LDS in wave order: deterministic), so a 512-wide layer is 16 independent workgroups with no global hand-off at all.
//
// Exact fp32 (v_mfma_f32_32x32x2_f32).  The MFMA sums over k in whatever order the lanes present it, so lane
// (l31, half) of wave w walks the CONTIGUOUS k-run  w * K/16 + half * K/32 .. + K/32  of its row -- 16-byte loads
// straight from global memory in operand layout, no LDS staging, no wasted elements.
#include "common.h"
#include "gemm_common.h"

namespace capdec {

constexpr int SK_WAVES = 16;
constexpr int SK_NT = SK_WAVES * 64, SK_EPT = 1024 / SK_NT;  // threads; outputs per thread
constexpr int SK_CHUNK = 4;  // float4 loads per operand in flight per lane (32 k values)

__global__ __launch_bounds__(SK_NT) void gemm_skinny_f32_kernel(GemmParams p) {
    __shared__ float red[SK_WAVES][32][33];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
    // gridDim.z workgroups share one output block, each with its own K slice (long reductions: a lone workgroup would
    // be bound by what ONE CU can pull in); combined below through the workspace, deterministically
    const int S = gridDim.z, kslice = p.K / S;
    const int krun = kslice / (2 * SK_WAVES);              // k values per lane (a multiple of 4: checked by the host)
    const int kbeg = blockIdx.z * kslice + wave * (kslice / SK_WAVES) + half * krun;
    const bool arow = m0 + l31 < p.M, brow = n0 + l31 < p.N;
    const float4* ap = reinterpret_cast<const float4*>(p.A + (size_t)min(m0 + l31, p.M - 1) * p.lda + kbeg);
    const float4* bp = reinterpret_cast<const float4*>(p.B + (size_t)min(n0 + l31, p.N - 1) * p.ldb + kbeg);
    // what the epilogue will need goes on its way now, under the operand loads (a dependent round trip after the
    // reduction would cost this latency-bound kernel another microsecond or two)
    const int en = n0 + (tid & 31);
    const float bias = (p.bias && en < p.N) ? p.bias[en] : 0.f;
    float addv[SK_EPT], auxv[SK_EPT];
#pragma unroll
    for (int e = 0; e < SK_EPT; ++e) {
        const int m = m0 + (tid >> 5) + (SK_NT / 32) * e;
        const bool ok = m < p.M && en < p.N;
        addv[e] = (p.addend && ok) ? p.addend[(size_t)m * p.ld_addend + en] : 0.f;
        auxv[e] = (p.act == CAPDEC_ACT_GELU_BWD && ok) ? p.aux[(size_t)m * p.ldc + en] : 0.f;
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int nq = krun / 4;  // float4 per lane
    // two register sets: the loads of chunk c + 1 are in flight under the MFMA chain of chunk c
    float4 av[2][SK_CHUNK], bv[2][SK_CHUNK];
    auto fetch = [&](int q0, float4 (&a)[SK_CHUNK], float4 (&b)[SK_CHUNK]) {
#pragma unroll
        for (int i = 0; i < SK_CHUNK; ++i) {
            const bool in = q0 + i < nq;
            a[i] = (in && arow) ? ap[q0 + i] : make_float4(0.f, 0.f, 0.f, 0.f);
            b[i] = (in && brow) ? bp[q0 + i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto chain = [&](const float4 (&a)[SK_CHUNK], const float4 (&b)[SK_CHUNK]) {
#pragma unroll
        for (int i = 0; i < SK_CHUNK; ++i) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[i].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[i].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[i].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[i].w, acc, 0, 0, 0);
        }
    };
    fetch(0, av[0], bv[0]);
    for (int q0 = 0; q0 < nq; q0 += 2 * SK_CHUNK) {
        if (q0 + SK_CHUNK < nq) fetch(q0 + SK_CHUNK, av[1], bv[1]);
        chain(av[0], bv[0]);
        if (q0 + SK_CHUNK < nq) {
            if (q0 + 2 * SK_CHUNK < nq) fetch(q0 + 2 * SK_CHUNK, av[0], bv[0]);
            chain(av[1], bv[1]);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][(r & 3) + 8 * (r >> 2) + 4 * half][l31] = acc[r];
    __syncthreads();
    // 1024 outputs, SK_NT threads: thread -> column tid & 31, row(s) (tid >> 5) + (SK_NT / 32) * e
    const int n = en;
    float vsum[SK_EPT];
#pragma unroll
    for (int e = 0; e < SK_EPT; ++e) {
        const int row = (tid >> 5) + (SK_NT / 32) * e;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < SK_WAVES; ++w) v += red[w][row][tid & 31];
        vsum[e] = v;
    }
    if (S > 1) {
        // the recipe of splitk_combine (gemm_common.h): write-through partials, drain, one ticket per workgroup; the
        // last arriver acquires, sums ALL slices in slice order (rounding independent of arrival order) and goes on
        const int tile = blockIdx.y * gridDim.x + blockIdx.x;
        float* slab0 = p.ws + (size_t)tile * S * 1024;
        float* mine = slab0 + (size_t)blockIdx.z * 1024;
#pragma unroll
        for (int e = 0; e < SK_EPT; ++e) __hip_atomic_store(&mine[e * SK_NT + tid], vsum[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // also: every thread is done reading `red`, whose first word now carries the ticket
        int* flag = reinterpret_cast<int*>(&red[0][0][0]);
        if (tid == 0) *flag = __hip_atomic_fetch_add(&p.counters[tile], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (*flag != S - 1) return;
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(&p.counters[tile], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < SK_EPT; ++e) {
            float v = 0.f;
            for (int z = 0; z < S; ++z) v += slab0[(size_t)z * 1024 + e * SK_NT + tid];
            vsum[e] = v;
        }
    }
#pragma unroll
    for (int e = 0; e < SK_EPT; ++e) {
        const int row = (tid >> 5) + (SK_NT / 32) * e, m = m0 + row;
        if (m < p.M && n < p.N) {
            float v = vsum[e];
            v += bias;
            const size_t o = (size_t)m * p.ldc + n;
            if (p.pre) p.pre[o] = v;
            if (p.act == CAPDEC_ACT_GELU) v = gelu_erf(v);
            else if (p.act == CAPDEC_ACT_GELU_BWD) v *= gelu_erf_grad(auxv[e]);
            if (p.drop.on()) v = p.drop.apply(v, (uint32_t)m * (uint32_t)p.N + (uint32_t)n);
            if (p.addend) v += addv[e];
            if (p.C2 && n >= p.n_split) p.C2[(size_t)m * p.ldc2 + (n - p.n_split)] = v;
            else p.C[o] = v;
        }
    }
}

int gemm_skinny_slices(int K);

// x[M, K] * W[N, K]^T with few rows: exact fp32, K a multiple of 128 (16 waves x 2 lane halves x one float4),
// 16-byte aligned operands with leading dimensions that are multiples of 4; no bias-gradient sums / CE partials;
// a K-sliced launch takes one ticket per 32 x 32 block, so more blocks than tickets go to the tiled kernel instead
bool gemm_skinny_ok(const GemmParams& p, bool AK, bool BKM, int dtype) {
    if (!(dtype == CAPDEC_F32 && AK && BKM && p.M <= 64 && p.K % (8 * SK_WAVES) == 0 && p.K >= 128 && p.lda % 4 == 0 &&
          p.ldb % 4 == 0 && aligned16(p.A) && aligned16(p.B) && !p.colsum && !p.ce_part && !p.ce_lse && !p.c16))
        return false;
    return gemm_skinny_slices(p.K) <= 1 || (long)cdiv(p.N, 32) * cdiv(p.M, 32) <= GEMM_SPLITK_MAX_TILES;
}

// K slices per output block: a function of K ALONE (512 of K per workgroup, at most 8 slices), so that a product's
// bits do not depend on how many columns it was launched with (capdec_decode_step merges q and k | v into one launch)
int gemm_skinny_slices(int K) {
    int s = K / 512;
    if (s > 8) s = 8;
    while (s > 1 && (K % s || (K / s) % (8 * SK_WAVES))) --s;
    return s < 1 ? 1 : s;
}

size_t gemm_skinny_workspace_bytes(int M, int N, int K) {
    const int s = gemm_skinny_slices(K);
    if (s <= 1) return 0;
    return GEMM_COUNTER_BYTES + (size_t)cdiv(N, 32) * cdiv(M, 32) * s * 1024 * sizeof(float);
}

int launch_gemm_skinny(GemmParams p, void* workspace, size_t workspace_bytes, hipStream_t st) {
    const int s = gemm_skinny_slices(p.K);
    if (s > 1) {
        const size_t need = gemm_skinny_workspace_bytes(p.M, p.N, p.K);
        if (!workspace || workspace_bytes < need || !aligned16(workspace)) {
            set_error("capdec_gemm: the few-row product needs %zu workspace bytes (16-byte aligned) for its K slices, got %zu",
                      need, workspace_bytes);
            return CAPDEC_ERR_WORKSPACE;
        }
        p.counters = (int*)workspace;
        p.ws = (float*)((char*)workspace + GEMM_COUNTER_BYTES);
    }
    hipLaunchKernelGGL(gemm_skinny_f32_kernel, dim3(cdiv(p.N, 32), cdiv(p.M, 32), s), dim3(SK_NT), 0, st, p);
    return CAPDEC_OK;
}

}  // namespace capdec
