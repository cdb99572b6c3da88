// bf16-storage GEMM, 256 x 256 block, eight phases per pair of k-tiles ("p8" kernel; plan 256 x 256, k-tile field 32).
//
// Same block as the pair kernel of gemm_bf16s_dma.hip (eight waves as 2 x 4, 128 x 64 outputs each, 128 KB of LDS, one
// workgroup per CU, two wave groups one barrier apart so that on every SIMD one wave multiplies while the other reads),
// but staged and consumed in QUARTERS of a k-tile, which is what lets loads stay in flight across barriers:
//   * a k-tile (64 deep) is four half-tiles of 16 KB -- A rows 0-127 / 128-255 of the image, B likewise -- and eight
//     half-tile slots (two k-tiles) make up the LDS;
//   * the image rows are dealt so that every wave's FIRST 64 rows lie in A half 0 and its second 64 in A half 1 (its
//     first 32 columns in B half 0, the other 32 in B half 1): half-tile (local row 64 wr + x) <-> tile row
//     128 wr + 64 half + x.  A phase multiplies one 64 x 32 quadrant of the wave's output over the whole k-tile
//     (8 MFMAs 32 x 32 x 16), and each half-tile is read in exactly one phase of the four:
//         phase 0: read B half 0 (4 ds_read_b128) + A half 0 (8)   -> quadrant (0, 0)
//         phase 1: read B half 1 (4)                               -> quadrant (0, 1)
//         phase 2: read A half 1 (8)                               -> quadrant (1, 1)
//         phase 3: (B half 0 is still in registers)                -> quadrant (1, 0)
//   * so a slot is free one to two phases after it was read, and every phase re-fills one: phase p requests the
//     half-tile that is needed 7 phases later (need order B0, A0, B1, A1).  Seven half-tiles are in flight or landed
//     ahead of the reader; once per k-tile (phase 3) a wave waits with vmcnt(6): everything but its last three requests
//     (two DMA instructions each) has landed, which is exactly the next k-tile.  The counter is never drained in the loop.
// Order rules (the two wave groups run one barrier apart; barrier n is the n-th s_barrier every wave executes):
//   RAW  a half-tile is read one phase (two barriers) after the vmcnt wait that covers it: both groups have then waited.
//   WAR  B half 0's slot is re-requested in the phase after its reads, so those reads are retired (lgkmcnt(8): the four
//        B reads are issued first) BEFORE the barrier that ends their read phase; the other slots are re-requested two
//        phases after their reads, which are retired by the lgkmcnt(0) in front of the MFMAs.
// All LDS reads are inline assembly (an ordinary LDS read beside an LDS DMA in flight draws a compiler-inserted
// s_waitcnt vmcnt(0): see attention_f32_ring.hip), barriers are bare s_barrier.
// Covers k-contiguous A and B (the forward products: activations x weights^T); no bias-gradient sums.  Other
// layouts and products with column sums keep the pair kernel.
#include "common.h"
#include <type_traits>
#include <cstdlib>

#include "gemm_common.h"

namespace capdec {

typedef __bf16 bf16x8d __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#define CAPDEC_GLDS16(gptr, lptr) \
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(gptr), \
                                     (void __attribute__((address_space(3)))*)(lptr), 16, 0, 0)

constexpr int P8_BK = 64;
constexpr int P8_HALF = 128 * P8_BK * 2;  // bytes of a half-tile
constexpr int P8_B0 = 4 * P8_HALF;        // B half-tiles start here (A: [buf][half], then B: [buf][half])
#ifndef CAPDEC_P8_MGROUP
#define CAPDEC_P8_MGROUP 8
#endif

template <int OFF>
__device__ __forceinline__ i32x4 p8_read128(unsigned addr) {
    i32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}

// ABL (diagnostic builds only): 1 = no DMA after the prologue, 2 = no MFMA, 3 = no fragment reads, 4 = no epilogue store
template <bool SPLITK, int ABL = 0>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2)))
void gemm_bf16s_p8_kernel(GemmParams p) {
    constexpr int BM = 256, BN = 256, TM = 4, TN = 2, NT = 512, WN = 4;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[8 * P8_HALF];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int wr = wave / WN, wc = wave % WN;
    const int wrow = wr * 128, wcol = wc * 64;
    const bool late = wave >= WN;  // wave-uniform: the group that runs one barrier behind
    const int nwg = gridDim.x;
    int wg = blockIdx.x;
    {   // XCD-aware tile order: a contiguous run of tiles per XCD (blocks b and b + 8 share one)
        const int q = nwg >> 3, r = nwg & 7, xcd = wg & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wg >> 3);
    }
    int tile_m, tile_n;
    {
        const int tiles_m = nwg / p.tiles_n;
        const int per_group = CAPDEC_P8_MGROUP * p.tiles_n;
        const int gid = wg / per_group, first_m = gid * CAPDEC_P8_MGROUP;
        const int gsz = min(tiles_m - first_m, CAPDEC_P8_MGROUP);
        const int in_g = wg - gid * per_group;
        tile_m = first_m + in_g % gsz;
        tile_n = in_g / gsz;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kbeg = blockIdx.z * p.ksplit;
    const int kend = min(p.K, kbeg + p.ksplit);
    const int nk = (kend - kbeg + P8_BK - 1) / P8_BK;
    const __bf16* A16 = reinterpret_cast<const __bf16*>(p.A) + kbeg;
    const __bf16* B16 = reinterpret_cast<const __bf16*>(p.B) + kbeg;

    // DMA sources: this wave's two 1 KiB pieces (8 rows x 128 B) of each half-tile, k-tile 0.  Piece s of local row lr
    // goes to slot s ^ ((lr >> 1) & 7) of its 128-byte row (the swizzle is applied here, on the source, and in the reads).
    unsigned offa[2][2], offb[2][2];  // [half][piece]
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int lr = (wave * 2 + j) * 8 + (lane >> 3);
        const int src = (lane & 7) ^ ((lr >> 1) & 7);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int row = m0 + (lr >> 6) * 128 + h * 64 + (lr & 63);
            const int col = n0 + (lr >> 5) * 64 + h * 32 + (lr & 31);
            offa[h][j] = (unsigned)min(row, p.M - 1) * (unsigned)p.lda + src * 8;
            offb[h][j] = (unsigned)min(col, p.N - 1) * (unsigned)p.ldb + src * 8;
        }
    }
    unsigned char* dma_dst = smem + wave * 2048;  // wave-uniform
    const unsigned lds0 = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)smem;
    // fragment read addresses, one per 16-deep k-step: row * 128 + (((2 ks + half) ^ ((l31 >> 1) & 7)) * 16)
    unsigned ra[4], rb[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const unsigned sl = (unsigned)(((2 * ks + half) ^ ((l31 >> 1) & 7)) * 16);
        ra[ks] = lds0 + (wr * 64 + l31) * 128 + sl;
        rb[ks] = lds0 + P8_B0 + (wc * 32 + l31) * 128 + sl;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    i32x4 fa[2][4], fb0[4], fb1[4];  // A fragments of the current half (2 row blocks x 4 k-steps), B halves 0 / 1

#define P8_BAR() asm volatile("s_barrier" ::: "memory")
    // request half-tile (operand, half h) of the k-tile that starts at element k0 into buffer BUF
    auto stage_a = [&](auto buf_c, auto h_c, int k0) {
        constexpr int SLOT = (decltype(buf_c)::value * 2 + decltype(h_c)::value) * P8_HALF;
        constexpr int H = decltype(h_c)::value;
        if (ABL == 1 && k0 >= 2 * P8_BK) return;
        CAPDEC_GLDS16(A16 + k0 + offa[H][0], dma_dst + SLOT);
        CAPDEC_GLDS16(A16 + k0 + offa[H][1], dma_dst + SLOT + 1024);
    };
    auto stage_b = [&](auto buf_c, auto h_c, int k0) {
        constexpr int SLOT = P8_B0 + (decltype(buf_c)::value * 2 + decltype(h_c)::value) * P8_HALF;
        constexpr int H = decltype(h_c)::value;
        if (ABL == 1 && k0 >= 2 * P8_BK) return;
        CAPDEC_GLDS16(B16 + k0 + offb[H][0], dma_dst + SLOT);
        CAPDEC_GLDS16(B16 + k0 + offb[H][1], dma_dst + SLOT + 1024);
    };
    auto read_a = [&](auto buf_c, auto h_c) {
        constexpr int SLOT = (decltype(buf_c)::value * 2 + decltype(h_c)::value) * P8_HALF;
        if (ABL == 3) return;
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            fa[ii][0] = ii ? p8_read128<SLOT + 4096>(ra[0]) : p8_read128<SLOT>(ra[0]);
            fa[ii][1] = ii ? p8_read128<SLOT + 4096>(ra[1]) : p8_read128<SLOT>(ra[1]);
            fa[ii][2] = ii ? p8_read128<SLOT + 4096>(ra[2]) : p8_read128<SLOT>(ra[2]);
            fa[ii][3] = ii ? p8_read128<SLOT + 4096>(ra[3]) : p8_read128<SLOT>(ra[3]);
        }
    };
    auto read_b = [&](auto buf_c, auto h_c, i32x4 (&fb)[4]) {
        constexpr int SLOT = (decltype(buf_c)::value * 2 + decltype(h_c)::value) * P8_HALF;  // rb already holds P8_B0
        if (ABL == 3) return;
        fb[0] = p8_read128<SLOT>(rb[0]);
        fb[1] = p8_read128<SLOT>(rb[1]);
        fb[2] = p8_read128<SLOT>(rb[2]);
        fb[3] = p8_read128<SLOT>(rb[3]);
    };
    auto multiply = [&](auto mq_c, const i32x4 (&fb)[4], auto nq_c) {
        constexpr int MQ = decltype(mq_c)::value, NQ = decltype(nq_c)::value;
        __builtin_amdgcn_sched_barrier(0);
        if (ABL == 2) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) asm volatile("" ::"v"(fa[0][ks]), "v"(fa[1][ks]), "v"(fb[ks]));
            return;
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
                acc[MQ * 2 + ii][NQ] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                    __builtin_bit_cast(bf16x8d, fa[ii][ks]), __builtin_bit_cast(bf16x8d, fb[ks]), acc[MQ * 2 + ii][NQ], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto retire_a = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[0][2]), "+v"(fa[0][3]),
                     "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(fa[1][2]), "+v"(fa[1][3]));
    };
    auto retire_b = [&](i32x4 (&fb)[4]) {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fb[0]), "+v"(fb[1]), "+v"(fb[2]), "+v"(fb[3]));
    };
    // the four phases of k-tile t, which lives in buffer BUF
    auto k_tile = [&](auto buf_c, int t) {
        constexpr int BUF = decltype(buf_c)::value;
        using B_ = std::integral_constant<int, BUF>;
        using O_ = std::integral_constant<int, BUF ^ 1>;
        using H0 = std::integral_constant<int, 0>;
        using H1 = std::integral_constant<int, 1>;
        const int k1 = (t + 1) * P8_BK, k2 = (t + 2) * P8_BK;
        const bool has1 = t + 1 < nk, has2 = t + 2 < nk;
        // phase 0
        read_b(B_{}, H0{}, fb0);
        read_a(B_{}, H0{});
        if (has1) stage_a(O_{}, H1{}, k1);
        asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(fb0[0]), "+v"(fb0[1]), "+v"(fb0[2]), "+v"(fb0[3]));
        P8_BAR();
        retire_a();
        multiply(H0{}, fb0, H0{});
        P8_BAR();
        // phase 1
        read_b(B_{}, H1{}, fb1);
        if (has2) stage_b(B_{}, H0{}, k2);
        P8_BAR();
        retire_b(fb1);
        multiply(H0{}, fb1, H1{});
        P8_BAR();
        // phase 2
        read_a(B_{}, H1{});
        if (has2) stage_a(B_{}, H0{}, k2);
        P8_BAR();
        retire_a();
        multiply(H1{}, fb1, H1{});
        P8_BAR();
        // phase 3
        if (has2) {
            stage_b(B_{}, H1{}, k2);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // all but the last three requests: k-tile t + 1 has landed
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        P8_BAR();
        multiply(H1{}, fb0, H0{});
        P8_BAR();
    };

    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    if (nk > 0) {
        stage_b(I0{}, I0{}, 0);
        stage_a(I0{}, I0{}, 0);
        stage_b(I0{}, I1{}, 0);
        stage_a(I0{}, I1{}, 0);
    }
    if (nk > 1) {
        stage_b(I1{}, I0{}, P8_BK);
        stage_a(I1{}, I0{}, P8_BK);
        stage_b(I1{}, I1{}, P8_BK);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    P8_BAR();            // every wave's pieces of k-tile 0 have landed
    if (late) P8_BAR();  // the stagger
    for (int t = 0; t < nk; t += 2) {
        k_tile(I0{}, t);
        if (t + 1 < nk) k_tile(I1{}, t + 1);
    }
    if (!late) P8_BAR();  // the stagger, paid back
    P8_BAR();             // everyone is done with the images: the epilogue reuses the LDS
#undef P8_BAR

    float csum = 0.f;
    if constexpr (SPLITK) {
        if (gridDim.z > 1 && !splitk_combine<TM, TN, NT>(p, acc, csum, false, tile_m * p.tiles_n + tile_n, m0, BM, tid,
                                                          reinterpret_cast<int*>(smem)))
            return;
    }
    if (p.ce_part) {
        ce_tile_partials<TM, TN, BM, BN, WN, NT>(p, acc, reinterpret_cast<float*>(smem), m0, n0, tile_n, tid);
        __syncthreads();  // its LDS reads are done before the waves write their private regions below
    }
    if (ABL == 4) {
        if (acc[0][0][0] == 123.456f) p.C[0] = acc[1][1][3] + acc[2][0][5] + acc[3][1][7];
        return;
    }
    // one workgroup per CU: nothing else hides the epilogue, so it is the vector form (gemm_common.h)
    epilogue_store_vec<TM, TN, 2>(p, acc, m0 + wrow, n0 + wcol, lane, reinterpret_cast<float*>(smem + wave * 16384));
}

// the operands the p8 kernel takes (beyond gemm_bf16s_dma_ok): both k-contiguous, no bias-gradient sums, a result the
// vector epilogue can store
bool gemm_bf16s_p8_ok(const GemmParams& p, bool AKC, bool BKC) {
    return AKC && BKC && p.colsum == nullptr && p.vec_epi;
}

void launch_gemm_bf16s_p8(const GemmParams& p, dim3 grid, hipStream_t st) {
#ifdef CAPDEC_P8_DIAG
    static const int abl = getenv("CAPDEC_P8_ABL") ? atoi(getenv("CAPDEC_P8_ABL")) : 0;
    if (abl == 1) { hipLaunchKernelGGL((gemm_bf16s_p8_kernel<false, 1>), grid, dim3(512), 0, st, p); return; }
    if (abl == 2) { hipLaunchKernelGGL((gemm_bf16s_p8_kernel<false, 2>), grid, dim3(512), 0, st, p); return; }
    if (abl == 3) { hipLaunchKernelGGL((gemm_bf16s_p8_kernel<false, 3>), grid, dim3(512), 0, st, p); return; }
    if (abl == 4) { hipLaunchKernelGGL((gemm_bf16s_p8_kernel<false, 4>), grid, dim3(512), 0, st, p); return; }
#endif
    if (grid.z > 1) hipLaunchKernelGGL((gemm_bf16s_p8_kernel<true>), grid, dim3(512), 0, st, p);
    else hipLaunchKernelGGL((gemm_bf16s_p8_kernel<false>), grid, dim3(512), 0, st, p);
}

}  // namespace capdec
