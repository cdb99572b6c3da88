// Shared between gemm_f32.hip and gemm_bf16.hip: kernel parameters, the fused epilogue, plans.
#pragma once
#include "common.h"

namespace capdec {

struct GemmParams {
    const float* A;
    const float* B;
    float* C;
    int M, N, K, lda, ldb, ldc;
    const float* bias;
    const float* addend;
    float* pre;
    const float* aux;
    float* colsum;
    int ld_addend, act;
    Dropout drop;
    int ksplit;        // K elements per z-slice (multiple of the k-tile)
    // split-K (gridDim.z = S > 1), combined inside the launch by each tile's last-arriving slice (splitk_combine):
    float* ws;         // [tile][S] slabs of BM*BN floats in the accumulator-register layout (float4 per thread)
    float* ws_colsum;  // [S][m_pad] bias-gradient partials
    int* counters;     // [tiles] arrival tickets; zero before the launch, reset to zero by the last arriver
    int m_pad;         // tiles_m * BM
    int tiles_n;
    // label-smoothed cross-entropy fused into the vocabulary projection (ref train.py:41-50):
    float4* ce_part;        // forward: per (row, column tile) softmax partials (max, sum exp(z - max), sum z, target logit | 0: with ce_y)
    const float* ce_lse;    // backward: A holds the LOGITS; the A loaders turn them into dlogits on their way to LDS
    const int64_t* ce_y;    //   dlogits[row, col] = gs * (exp(z - lse[row]) - eps/V - (1 - eps) * [col == y[row]])
    float ce_gs, ce_u, ce_w;  // grad scale, eps / V, 1 - eps
    __bf16* c16;            // if set: the result goes THERE as bf16 (row stride ldc16), not to C
    int ldc16;
    int unpadded;           // bf16 storage: bit 0 / 1 = A's / B's row padding behind K is NOT known to be zero
    // gemm_skinny.hip only (library-internal, capdec_decode_step): columns n >= n_split go to C2[m, n - n_split]
    // (row stride ldc2) -- one product writes the query block and, through the cache's row stride, the K | V block
    float* C2;
    int ldc2, n_split;
    int vec_epi;            // the result can leave through the vector form of the epilogue (epilogue_vec_ok; set by capdec_gemm)
};

// capdec_gemm_grouped: the products of one launch, passed by value (kernel arguments)
struct GemmGroupParams {
    int n;
    int first[CAPDEC_GEMM_GROUP_MAX];   // first workgroup of product i (a multiple of 8)
    int tiles[CAPDEC_GEMM_GROUP_MAX];   // its tile count; the workgroups between first + tiles and the next first are idle
    GemmParams p[CAPDEC_GEMM_GROUP_MAX];
};

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GEMM_SPLITK_MAX_TILES = 4096;                       // arrival counters at the head of the workspace
constexpr size_t GEMM_COUNTER_BYTES = GEMM_SPLITK_MAX_TILES * 4;  // 16 KiB

// In-launch split-K combine, deterministic.  Every slice stores its partial accumulators to its slab WRITE-THROUGH
// (sc1 stores, 16 B per lane, 4 KiB contiguous per store instruction: they leave the XCD's L2, so no release fence
// and no L2 write-back is needed), drains them, and one lane takes a ticket; the slice that draws the last ticket
// acquires (one agent-scope L1 invalidate), re-reads ALL S slabs (its own included) in slice order z = 0..S-1 --
// so the sum's rounding does not depend on which slice happened to arrive last -- and returns true: that
// workgroup alone runs the epilogue.  Placement-independent (correct wherever a tile's slices run); the ticket
// word is reset by the last arriver, so it is zero again at the next launch (the caller zeroes it once, at
// allocation).  `lds_flag`: one int of the kernel's own (now idle) LDS array.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int TM, int TN, int NT = 256>
__device__ __forceinline__ bool splitk_combine(const GemmParams& p, f32x16 (&acc)[TM][TN], float& csum, bool do_colsum,
                                               int tile_id, int m0, int bm, int tid, int* lds_flag) {
    const int S = gridDim.z, z = blockIdx.z;
    constexpr int PER = TM * TN * 4;  // float4 per thread per slab (NT threads per workgroup)
    float4* slab0 = reinterpret_cast<float4*>(p.ws) + (size_t)tile_id * S * (PER * NT);
    float4* mine = slab0 + (size_t)z * (PER * NT);
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(mine, 0, PER * NT * 16, 0x00020000);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                u32x4 v;
                v.x = __float_as_uint(acc[i][j][4 * q]);
                v.y = __float_as_uint(acc[i][j][4 * q + 1]);
                v.z = __float_as_uint(acc[i][j][4 * q + 2]);
                v.w = __float_as_uint(acc[i][j][4 * q + 3]);
                __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, (((i * TN + j) * 4 + q) * NT + tid) * 16, 0, 16 /* sc1 */);
            }
    if (do_colsum && tid < bm)
        __hip_atomic_store(&p.ws_colsum[(size_t)z * p.m_pad + m0 + tid], csum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its write-through stores ...
    __syncthreads();                                   // ... before one lane signals for all of them
    if (tid == 0)
        *lds_flag = __hip_atomic_fetch_add(&p.counters[tile_id], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (*lds_flag != S - 1) return false;
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(&p.counters[tile_id], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // about 8 float4 loads in flight per thread: the combine must fit in the registers of the main loop, or it
    // would cost every workgroup of the launch (not only the last arrivers) its occupancy
    constexpr int UNR = PER >= 8 ? 1 : 8 / PER;
#pragma unroll UNR
    for (int s = 0; s < S; ++s) {
        const float4* sl = slab0 + (size_t)s * (PER * NT);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = sl[((i * TN + j) * 4 + q) * NT + tid];
                    acc[i][j][4 * q] += v.x; acc[i][j][4 * q + 1] += v.y;
                    acc[i][j][4 * q + 2] += v.z; acc[i][j][4 * q + 3] += v.w;
                }
    }
    if (do_colsum && tid < bm) {
        float t = 0.f;
        for (int s = 0; s < S; ++s) t += p.ws_colsum[(size_t)s * p.m_pad + m0 + tid];
        csum = t;
    }
    return true;
}

// The fused epilogue over a wave's accumulators (32 x 32 MFMA layout: lane (l31, half) holds column l31 and rows
// (r & 3) + 8 (r >> 2) + 4 half of each 32 x 32 block), written so that no load ever waits behind a store: on gfx9
// loads and stores share one in-order counter, and an element-by-element "load bias / residual, add, store" loop
// makes every load wait for the previous element's store to be acknowledged (~64 serialized round trips per lane
// and tile -- the shape this epilogue had before).  Here the bias is read once per column, and the residual /
// GELU-input values of the NEXT half block (8 elements) are requested before the current half block is stored.
// Same arithmetic, in the same order, as before: v = acc + bias; pre = v; activation; dropout; + addend.
template <int TM, int TN>
__device__ __forceinline__ void epilogue_store(const GemmParams& p, const f32x16 (&acc)[TM][TN], int mbase, int nbase,
                                               int l31, int half) {
    float bj[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = nbase + j * 32 + l31;
        bj[j] = (p.bias && n < p.N) ? p.bias[n] : 0.f;
    }
    // one staged stream: the residual if there is one, else the GELU input (both at once -- no caller of this
    // library does that -- reads the GELU input in place, the slow way)
    const bool has_add = p.addend != nullptr, gbwd = p.act == CAPDEC_ACT_GELU_BWD;
    const float* stg = has_add ? p.addend : p.aux;
    const int ld_stg = has_add ? p.ld_addend : p.ldc;
    const bool staged = has_add || gbwd;
    float st[2][8];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = nbase + j * 32 + l31;
#pragma unroll
            for (int h8 = 0; h8 < 2; ++h8) {
                // request the next half block (8 elements) before this one is stored
                const bool first = i == 0 && j == 0 && h8 == 0, last = i == TM - 1 && j == TN - 1 && h8 == 1;
                const int cur = h8, nxt = h8 ^ 1;
                if (first) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int m = mbase + (e & 3) + 8 * (e >> 2) + 4 * half;
                        st[0][e] = (staged && m < p.M && n < p.N) ? stg[(size_t)m * ld_stg + n] : 0.f;
                    }
                }
                if (!last) {
                    const int i2 = h8 ? (j == TN - 1 ? i + 1 : i) : i, j2 = h8 ? (j == TN - 1 ? 0 : j + 1) : j;
                    const int n2 = nbase + j2 * 32 + l31;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int r = nxt * 8 + e;
                        const int m = mbase + i2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                        st[nxt][e] = (staged && m < p.M && n2 < p.N) ? stg[(size_t)m * ld_stg + n2] : 0.f;
                    }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int r = h8 * 8 + e;
                    const int m = mbase + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (m < p.M && n < p.N) {
                        const size_t o = (size_t)m * p.ldc + n;
                        float v = acc[i][j][r] + bj[j];
                        if (p.pre) p.pre[o] = v;
                        if (p.act == CAPDEC_ACT_GELU) v = gelu_erf(v);
                        else if (gbwd) v *= gelu_erf_grad(has_add ? p.aux[o] : st[cur][e]);
                        if (p.drop.on()) v = p.drop.apply(v, (uint32_t)m * (uint32_t)p.N + (uint32_t)n);
                        if (has_add) {  // never contracted with the activation's last product (the vector form agrees bit for bit)
#pragma clang fp contract(off)
                            v = v + st[cur][e];
                        }
                        if (p.c16) p.c16[(size_t)m * p.ldc16 + n] = (__bf16)v;
                        else p.C[o] = v;
                    }
                }
            }
        }
}

// ---- the same epilogue in VECTOR form ---------------------------------------------------------------------------
// The element form above is TM * TN * 16 unrolled element bodies (bounds test, option tests, 64-bit address, one
// 4-byte store each): ~25 000 instructions of straight-line code for a 128 x 64 wave block.  Kernels with several
// workgroups per CU hide it under their neighbours' products; the 256 x 256 block (one workgroup per CU) spent 39 of
// its 129 us in it on a K = 1024 product.  Here the accumulators of a wave's (32 TM) x 64 block go through a
// wave-private LDS region, 32 PI rows at a time, so that a lane then owns four CONSECUTIVE columns of a row: 16-byte
// loads and stores, a compact loop, and the residual / GELU-input vectors of the next row group are requested before this
// group's stores (loads never wait behind stores).  Same arithmetic per
// element in the same order: v = acc + bias; pre = v; activation; dropout; + addend.
// Needs N, ldc (ld_addend, ldc16) multiples of 4 and 16-byte aligned bases (8-byte for c16): epilogue_vec_ok.
// `wl`: 32 PI x 32 TN floats of LDS that only this wave touches; unpadded -- the writes put a half-wave on 32 consecutive
// columns of a row, the 16-byte reads are served in the 16-lane groups of ds_read_b128, which cover 16 distinct
// slots of the 256-byte bank row.
static inline bool epilogue_vec_ok(const GemmParams& p) {
    auto al = [](const void* q, uintptr_t a) { return (reinterpret_cast<uintptr_t>(q) & (a - 1)) == 0; };
    if (p.N % 4 || p.ldc % 4) return false;
    if (p.c16 ? (p.ldc16 % 4 || !al(p.c16, 8)) : !al(p.C, 16)) return false;
    if (p.bias && !al(p.bias, 16)) return false;
    if (p.pre && !al(p.pre, 16)) return false;
    if (p.addend && (p.ld_addend % 4 || !al(p.addend, 16))) return false;
    if (p.act == CAPDEC_ACT_GELU_BWD && !al(p.aux, 16)) return false;
    return true;
}

// Geometry of a wave's block for the vector epilogue: block row i (32 rows) starts at tile row
// mbase + (i / PI) * pass_rows + (i % PI) * 32, block column j at nbase + j * col_stride.  The usual block is contiguous
// (pass_rows = 32 PI, col_stride = 32); the eight-phase kernel deals rows and columns to its waves in two halves.
template <int TM, int TN, int PI>
__device__ __forceinline__ void epilogue_store_vec(const GemmParams& p, const f32x16 (&acc)[TM][TN], int mbase, int nbase,
                                                   int lane, float* wl, int pass_rows = 32 * PI, int col_stride = 32) {
    static_assert(TN == 1 || TN == 2, "a wave block is 32 or 64 columns wide");
    constexpr int W = 32 * TN, LPR = W / 4, RPI = 64 / LPR;  // lanes per row, rows per wave-wide 16-byte access
    constexpr int NIT = PI * 32 / RPI;                       // row groups per pass
    // the options are wave-uniform run-time branches inside ONE loop body (one body per kernel keeps the code and the
    // build small); every path stores exactly one result vector per row group, so the counted wait in front of the
    // prefetched residual / GELU-input vectors never has to cover this group's stores
    const bool PRE = p.pre != nullptr, DROP = p.drop.on(), ADD = p.addend != nullptr, C16 = p.c16 != nullptr;
    const bool GELU = p.act == CAPDEC_ACT_GELU, GBWD = p.act == CAPDEC_ACT_GELU_BWD;
    const int l31 = lane & 31, half = lane >> 5, c4 = (lane % LPR) * 4, rsub = lane / LPR;
    const int n = nbase + (c4 & 31) + (c4 >> 5) * col_stride;
    const bool n_ok = n < p.N;  // N % 4 == 0: all four columns or none
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias && n_ok) b4 = *reinterpret_cast<const float4*>(p.bias + n);
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int ps = 0; ps < TM / PI; ++ps) {
#pragma unroll
        for (int ii = 0; ii < PI; ++ii)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    wl[(ii * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * W + j * 32 + l31] = acc[ps * PI + ii][j][r];
        const int m_first = mbase + ps * pass_rows + rsub;
        if (!ADD && !GBWD) {
            // no operand to fetch: a loop of its own WITHOUT the loop-carried loads below -- with them the compiler closes every turn
            // with s_waitcnt vmcnt(0), which also waits for the turn's STORES (a round trip of ~1000 cycles each: 31 000 cycles for
            // the 32 row groups of a 256 x 256 tile, against ~4 000 for 256 KB of stores in flight; tools/micro/store_rate.hip)
#pragma unroll 1
            for (int it = 0; it < NIT; ++it) {
                const int m = m_first + it * RPI;
                const float4 v = *reinterpret_cast<const float4*>(&wl[(it * RPI + rsub) * W + c4]);
                if (m < p.M && n_ok) {
                    const size_t o = (size_t)m * p.ldc + n;
                    float e[4] = {v.x + b4.x, v.y + b4.y, v.z + b4.z, v.w + b4.w};
                    if (PRE) *reinterpret_cast<float4*>(p.pre + o) = make_float4(e[0], e[1], e[2], e[3]);
                    if (GELU) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) e[c] = gelu_erf(e[c]);
                    }
                    if (DROP) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) e[c] = p.drop.apply(e[c], (uint32_t)m * (uint32_t)p.N + (uint32_t)(n + c));
                    }
                    if (C16) {
                        typedef __bf16 bf16x4e __attribute__((ext_vector_type(4)));
                        const bf16x4e h = {(__bf16)e[0], (__bf16)e[1], (__bf16)e[2], (__bf16)e[3]};
                        *reinterpret_cast<bf16x4e*>(p.c16 + (size_t)m * p.ldc16 + n) = h;
                    } else {
                        *reinterpret_cast<float4*>(p.C + o) = make_float4(e[0], e[1], e[2], e[3]);
                    }
                }
            }
            continue;
        }
        // A residual (addend) or GELU-input (aux) vector per row group.  Memory operations retire in order, so a wait for a
        // fetched vector also waits for every store issued before the fetch: fetched one row group ahead of its use, each turn
        // would wait for the previous turn's stores (~1000 cycles each).  So the vectors of G row groups are requested together,
        // then those G groups are computed and stored with nothing to wait for: one store round trip per G groups.
        constexpr int G = TN == 1 ? 2 : (NIT < 4 ? NIT : 4);  // (32-column wave blocks belong to the register-capped fp32 tiles: two at a time)
        static_assert(NIT % G == 0, "row groups per pass: a multiple of the fetch batch");
#pragma unroll 1
        for (int it0 = 0; it0 < NIT; it0 += G) {
            typedef float f32x4v __attribute__((ext_vector_type(4)));
            f32x4v nadd[G], naux[G];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int mm = m_first + (it0 + g) * RPI;
                const bool ok = mm < p.M && n_ok;
                nadd[g] = naux[g] = f32x4v{0.f, 0.f, 0.f, 0.f};
                if (ADD && ok) nadd[g] = *reinterpret_cast<const f32x4v*>(p.addend + (size_t)mm * p.ld_addend + n);
                if (GBWD && ok) naux[g] = *reinterpret_cast<const f32x4v*>(p.aux + (size_t)mm * p.ldc + n);
            }
            // all of them are waited for HERE, on every path (a row group wholly outside the matrix skips its body: its vector
            // would stay pending and every later reuse of its registers would draw a full wait, stores included)
#pragma unroll
            for (int g = 0; g < G; ++g) asm volatile("" : "+v"(nadd[g]), "+v"(naux[g]));
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int it = it0 + g, m = m_first + it * RPI;
                const f32x4v add = nadd[g], aux = naux[g];
                const float4 v = *reinterpret_cast<const float4*>(&wl[(it * RPI + rsub) * W + c4]);
                if (m < p.M && n_ok) {
                    const size_t o = (size_t)m * p.ldc + n;
                    float e[4] = {v.x + b4.x, v.y + b4.y, v.z + b4.z, v.w + b4.w};
                    if (PRE) *reinterpret_cast<float4*>(p.pre + o) = make_float4(e[0], e[1], e[2], e[3]);
                    const float ax[4] = {aux.x, aux.y, aux.z, aux.w}, ad[4] = {add.x, add.y, add.z, add.w};
                    if (GELU) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) e[c] = gelu_erf(e[c]);
                    } else if (GBWD) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) e[c] *= gelu_erf_grad(ax[c]);
                    }
                    if (DROP) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) e[c] = p.drop.apply(e[c], (uint32_t)m * (uint32_t)p.N + (uint32_t)(n + c));
                    }
                    if (ADD) {  // never contracted with the product in front of it (the element form agrees bit for bit)
#pragma clang fp contract(off)
#pragma unroll
                        for (int c = 0; c < 4; ++c) e[c] = e[c] + ad[c];
                    }
                    if (C16) {
                        typedef __bf16 bf16x4e __attribute__((ext_vector_type(4)));
                        const bf16x4e h = {(__bf16)e[0], (__bf16)e[1], (__bf16)e[2], (__bf16)e[3]};
                        *reinterpret_cast<bf16x4e*>(p.c16 + (size_t)m * p.ldc16 + n) = h;
                    } else {
                        *reinterpret_cast<float4*>(p.C + o) = make_float4(e[0], e[1], e[2], e[3]);
                    }
                }
            }
        }
    }
}

// Butterfly reductions over aligned groups of L = 8 / 16 lanes as DPP moves (quad permutes, then the half-row and row
// mirrors: VALU only), in place of ds_bpermute round trips through the LDS crossbar: every lane ends with the group's
// result, bit-identical to the xor butterfly (after the quad steps a quad's lanes hold equal bits, so the partner a mirror
// picks inside the other quad / half does not matter).
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
template <int L>
__device__ __forceinline__ float group_max(float v) {
    static_assert(L == 8 || L == 16, "lane groups of 8 or 16");
    v = fmaxf(v, dpp_move<0xB1>(v));   // quad_perm [1,0,3,2]
    v = fmaxf(v, dpp_move<0x4E>(v));   // quad_perm [2,3,0,1]
    v = fmaxf(v, dpp_move<0x141>(v));  // row_half_mirror
    if (L == 16) v = fmaxf(v, dpp_move<0x140>(v));  // row_mirror
    return v;
}
template <int L>
__device__ __forceinline__ float group_sum(float v) {
    static_assert(L == 8 || L == 16, "lane groups of 8 or 16");
    v += dpp_move<0xB1>(v);
    v += dpp_move<0x4E>(v);
    v += dpp_move<0x141>(v);
    if (L == 16) v += dpp_move<0x140>(v);
    return v;
}

// ---- fused cross-entropy pieces shared by the three GEMM kernels (same accumulator layout) ---------------------
// Backward: four consecutive vocabulary columns col0.. of logits row `row` -> their loss gradients.  `nvalid`
// components (and only if row_ok) are inside the matrix; the rest stay the zero padding the tile expects.
__device__ __forceinline__ float4 ce_dlogits4(const GemmParams& p, float4 z, float lse, int y, int col0, int nvalid,
                                              bool row_ok) {
    float r[4] = {z.x, z.y, z.z, z.w};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float g = __expf(r[c] - lse) - p.ce_u;
        if (col0 + c == y) g -= p.ce_w;
        r[c] = (row_ok && c < nvalid) ? g * p.ce_gs : 0.f;
    }
    return make_float4(r[0], r[1], r[2], r[3]);
}
__device__ __forceinline__ int ce_target(const GemmParams& p, int row, int V) {
    const int64_t y = p.ce_y[row];
    return y < 0 ? 0 : (y >= V ? V - 1 : (int)y);  // out-of-range ids are clamped, as capdec_softmax_ce does
}

// the logits product's result vector: fp32 to C, or bf16 to c16 (the bf16-storage mode keeps its logits in bf16)
__device__ __forceinline__ void ce_store4(const GemmParams& p, int m, int n, const float (&e)[4], bool c16) {
    if (c16) {
        typedef __bf16 bf16x4e __attribute__((ext_vector_type(4)));
        const bf16x4e h = {(__bf16)e[0], (__bf16)e[1], (__bf16)e[2], (__bf16)e[3]};
        *reinterpret_cast<bf16x4e*>(p.c16 + (size_t)m * p.ldc16 + n) = h;
    } else {
        *reinterpret_cast<float4*>(p.C + (size_t)m * p.ldc + n) = make_float4(e[0], e[1], e[2], e[3]);
    }
}

// Forward: per row of this workgroup's BM x BN logits tile (acc + bias), the softmax partials over the tile's
// columns: (max, sum exp(z - max), sum z).  32 rows at a time go through LDS as [row][BN + 1] (written by the two
// waves that own them in the accumulator layout), then 8 threads per row reduce BN / 8 columns each.  `lds`: the
// kernel's own (idle) LDS array, >= 32 * (BN + 1) floats.
// MAP = 1: the eight-phase kernel's wave blocks (gemm_bf16s_p8.hip): 2 x 4 waves, block row i of wave row wr = tile rows
// 128 (i / 2) + 64 wr + 32 (i % 2), block column j of wave column wc = tile columns 128 j + 32 wc.
template <int TM, int TN, int BM, int BN, int WN = 2, int NT = 256, int MAP = 0>
__device__ __forceinline__ void ce_tile_partials(const GemmParams& p, const f32x16 (&acc)[TM][TN], float* lds, int m0,
                                                 int n0, int tile_n, int tid) {
    // WN waves side by side along N (each 32 * TN columns), NT / 64 / WN wave rows (each 32 * TM rows); NT / 32
    // threads reduce one row
    constexpr int LDR = BN + 1, TPR = NT / 32, CPT = BN / TPR, WM = NT / 64 / WN;
    const int lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int wr = wave / WN, wc = wave % WN;
#pragma unroll
    for (int c = 0; c < WM * TM; ++c) {
        // chunk c = tile rows 32 c .. 32 c + 31: owned by wave row cw as its block row ci
        const int cw = MAP == 1 ? (c >> 1) & 1 : c / TM, ci = MAP == 1 ? (c >> 2) * 2 + (c & 1) : c % TM;
        __syncthreads();
        if (wr == cw) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = MAP == 1 ? j * 128 + wc * 32 + l31 : wc * (32 * TN) + j * 32 + l31;
                const int n = n0 + col;
                const float b = (p.bias && n < p.N) ? p.bias[n] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) lds[((r & 3) + 8 * (r >> 2) + 4 * half) * LDR + col] = acc[ci][j][r] + b;
            }
        }
        __syncthreads();
        const int row = tid / TPR, part = tid % TPR;
        const int m = MAP == 1 ? m0 + c * 32 + row : m0 + cw * (BM / WM) + ci * 32 + row;
        const int ycol = (p.ce_y && m < p.M) ? ce_target(p, m, p.N) - n0 : -1;  // the row's target column inside this tile, if any
        float v[CPT];
        float mx = -INFINITY, sz = 0.f, zy = 0.f;
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const int col = k * TPR + part;  // interleaved: the TPR threads of a row read consecutive words (no bank conflicts)
            const bool ok = n0 + col < p.N;
            v[k] = ok ? lds[row * LDR + col] : -INFINITY;
            mx = fmaxf(mx, v[k]);
            sz += ok ? v[k] : 0.f;
            if (col == ycol) zy = v[k];
        }
        mx = group_max<TPR>(mx);
        float se = 0.f;
#pragma unroll
        for (int k = 0; k < CPT; ++k) se += __expf(v[k] - mx);  // excluded columns: exp(-inf) = 0 (mx is finite: column 0 of the tile is in)
        se = group_sum<TPR>(se);
        sz = group_sum<TPR>(sz);
        if (p.ce_y) zy = group_sum<TPR>(zy);  // one thread of the row holds it, the others zero: exact
        if (part == 0 && m < p.M) p.ce_part[(size_t)m * p.tiles_n + tile_n] = make_float4(mx, se, sz, zy);
    }
}

// The logits product's epilogue in vector form WITH the softmax partials of the fused cross-entropy (the vector
// counterpart of ce_tile_partials + epilogue_store for the usual wave geometry: 2 x 2 waves, contiguous wave blocks).
// logits = acc + bias are stored as fp32 by the vector loop; on the way a row of the wave block (LPR lanes x 4 columns)
// gives (max, sum exp(z - max), sum z) over the wave's columns by log2(LPR) shuffle steps each, left in LDS per
// (tile row, wave column); after ONE barrier a thread per tile row combines the two wave columns in a fixed order.
// Deterministic; rounding differs from ce_tile_partials in the last bits (another summation order), which the finalize
// kernel's sum over the tiles absorbs.  `part`: BM x 2 float4 of LDS outside the waves' private regions.
template <int TM, int TN, int BM>
__device__ __forceinline__ void epilogue_store_vec_ce(const GemmParams& p, const f32x16 (&acc)[TM][TN], int m0, int n0, int wrow,
                                                      int wcol, int wc, int lane, float* wl, float4* part, int tile_n, int tid) {
    static_assert(TN == 1 || TN == 2, "a wave block is 32 or 64 columns wide");
    constexpr int W = 32 * TN, LPR = W / 4, RPI = 64 / LPR, NIT = 32 / RPI;
    const int l31 = lane & 31, half = lane >> 5, c4 = (lane % LPR) * 4, rsub = lane / LPR;
    const int n = n0 + wcol + c4;
    const bool n_ok = n < p.N;  // N % 4 == 0 (epilogue_vec_ok): all four columns or none
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias && n_ok) b4 = *reinterpret_cast<const float4*>(p.bias + n);
    // with the targets at hand the partial's fourth component carries the row's target logit (if its column is in this
    // tile): the tile's BM targets are staged behind the partials (workgroup-uniform branch, one more barrier)
    int* yt = reinterpret_cast<int*>(part + BM * 2);
    const bool ZY = p.ce_y != nullptr, C16 = p.c16 != nullptr;
    if (ZY) {
        if (tid < BM) yt[tid] = m0 + tid < p.M ? ce_target(p, m0 + tid, p.N) : -1;
        __syncthreads();
    }
#pragma unroll
    for (int ps = 0; ps < TM; ++ps) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) wl[((r & 3) + 8 * (r >> 2) + 4 * half) * W + j * 32 + l31] = acc[ps][j][r];
        const int row0 = wrow + ps * 32 + rsub;  // tile row of this lane's first row group
#pragma unroll 1
        for (int it = 0; it < NIT; ++it) {
            const int trow = row0 + it * RPI, m = m0 + trow;
            const float4 v = *reinterpret_cast<const float4*>(&wl[(it * RPI + rsub) * W + c4]);
            const float e[4] = {v.x + b4.x, v.y + b4.y, v.z + b4.z, v.w + b4.w};
            float mx = n_ok ? fmaxf(fmaxf(e[0], e[1]), fmaxf(e[2], e[3])) : -INFINITY;
            mx = group_max<LPR>(mx);
            const float mref = mx == -INFINITY ? 0.f : mx;  // a wave block wholly past column N: no exp(-inf + inf)
            float se = n_ok ? (__expf(e[0] - mref) + __expf(e[1] - mref)) + (__expf(e[2] - mref) + __expf(e[3] - mref)) : 0.f;
            float sz = n_ok ? (e[0] + e[1]) + (e[2] + e[3]) : 0.f;
            se = group_sum<LPR>(se);
            sz = group_sum<LPR>(sz);
            if ((lane % LPR) == 0) part[trow * 2 + wc] = make_float4(mx, se, sz, 0.f);
            if (ZY) {  // the one lane that holds the row's target column overwrites the zero (a wave's LDS writes land in order)
                const unsigned yr = (unsigned)(yt[trow] - n);  // 0..3 there (then n_ok: targets are < N)
                if (yr < 4u) reinterpret_cast<float*>(&part[trow * 2 + wc])[3] = yr == 0 ? e[0] : yr == 1 ? e[1] : yr == 2 ? e[2] : e[3];
            }
            if (m < p.M && n_ok) ce_store4(p, m, n, e, C16);
        }
    }
    __syncthreads();
    if (tid < BM && m0 + tid < p.M) {
        const float4 q0 = part[tid * 2], q1 = part[tid * 2 + 1];
        const float mx = fmaxf(q0.x, q1.x);  // finite: column 0 of the tile is inside the matrix
        const float se = q0.y * __expf(q0.x - mx) + q1.y * __expf(q1.x - mx);
        p.ce_part[(size_t)(m0 + tid) * p.tiles_n + tile_n] = make_float4(mx, se, q0.z + q1.z, q0.w + q1.w);
    }
}

struct GemmPlan {
    int bm, bn, bk, split, ksplit;
};
struct GemmForce {
    int bm, bn, split, bk;
};
extern GemmForce g_gemm_force;  // tuning hook (capdec_gemm_tune)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// workspace bytes of a split-K plan: ticket words, one slab per (tile, slice), bias-gradient partials
static inline size_t splitk_workspace_bytes(const GemmPlan& pl, int M, int N) {
    if (pl.split <= 1) return 0;
    const size_t tiles_m = cdiv(M, pl.bm), tiles = tiles_m * cdiv(N, pl.bn);
    return GEMM_COUNTER_BYTES + (tiles * pl.split * pl.bm * pl.bn + (size_t)pl.split * tiles_m * pl.bm) * sizeof(float);
}

GemmPlan plan_gemm_bf16(int M, int N, int K, const GemmForce& force);
void launch_gemm_bf16(const GemmParams& p, const GemmPlan& pl, bool AK, bool BKM, bool vec, hipStream_t st);
GemmPlan plan_gemm_f32x3(int M, int N, int K, const GemmForce& force);
int launch_gemm_f32x3(const GemmParams& p, const GemmPlan& pl, bool AK, bool BKM, bool vec, hipStream_t st);
GemmPlan plan_gemm_bf16s(int M, int N, int K, const GemmForce& force);
void launch_gemm_bf16s(const GemmParams& p, const GemmPlan& pl, bool AKC, bool BKC, hipStream_t st);  // A, B: bf16
bool gemm_skinny_ok(const GemmParams& p, bool AK, bool BKM, int dtype);  // few-row decode products (gemm_skinny.hip)
bool gemm_plan_forced(int ta, int tb, int M, int N, int K, int dtype);   // tuning hook or pinned plan applies (gemm_f32.hip)
int launch_gemm_skinny(GemmParams p, void* workspace, size_t workspace_bytes, hipStream_t st);  // 0 or CAPDEC_ERR_WORKSPACE
size_t gemm_skinny_workspace_bytes(int M, int N, int K);
bool gemm_bf16s_dma_ok(const GemmParams& p, bool AKC, bool BKC);  // operands fit the LDS-DMA kernels (gemm_bf16s_dma.hip)

}  // namespace capdec
