// bf16-storage GEMM, LDS-DMA form (the 128 x 128 plan of CAPDEC_BF16S when the operands allow it).
//
// The register-staged kernel of gemm_bf16s.hip pays for every tile twice on the LDS side: the ds_write_b128 pass
// (13 cycles per wave-instruction of VGPR -> LDS transfer) next to the fragment reads, and 48 staging VGPRs that cap
// it at two waves per SIMD.  Here the tiles go global -> LDS directly (global_load_lds_dwordx4: no VGPRs, no
// ds_write), a workgroup holds 32 KB of LDS and ~100 VGPRs, and three to four workgroups per CU overlap each
// other's load / compute phases (one LDS image per workgroup, two barriers per k-tile).
//
// An LDS-DMA instruction writes 1 KiB lane-linear (wave base + 16 B x lane), so the LDS images are unpadded and the
// bank-conflict swizzles are applied on the SOURCE side (which 16-byte piece a lane fetches) and again in the reads:
//   k-contiguous operand ([row][ld]):  image [128 rows][64 k] of 128-byte rows, piece s of row r at slot
//       s ^ ((r >> 1) & 7); fragments by ds_read_b128 (the 16 lanes of a read group then cover 16 distinct slots
//       of the 256-byte bank row: conflict-free).
//   k-major operand ([k][ld]):  image [64 k][128 rows] of 256-byte rows, piece c of k-row r at slot
//       c ^ (((r & 3) << 2) | ((r >> 2) & 3)); fragments by two ds_read_b64_tr_b16 (the hardware transposing read:
//       a 16-lane group fetches 4 k-rows x 16 columns and each lane receives its column's 4 k values) -- so the
//       weight-gradient and input-gradient products need no transposed copy and no register transpose either.
// Two block shapes:
//   128 x 128, 4 waves (64 x 64 each): 32 KB per image pair, ~130 VGPRs, three workgroups per CU (one image pair) or
//       two (two pairs).  Every k-tile moves 32 KB through the CU's vector-memory path for 2.1 MFLOP: the measured
//       rate (0.31-0.37 of the matrix peak) sits at ~75 % of what a CU can pull out of L2 (66-73 GB/s, guide
//       section 'Indexed rows') -- the tile, not the schedule, is the limit.
//   256 x 256, 8 waves (128 x 64 each): twice the flops per byte moved; two image pairs (128 KB, one workgroup per CU);
//       a kernel of its own further down (gemm_bf16s_pair_kernel: two wave groups one phase apart).
// A DMA cannot mask: rows / columns past the matrix edge are clamped to the last valid one (they only feed output
// elements that are never stored); the reduction tail relies on the k-contiguous operands' zero padding (pitch >=
// K rounded up to 64, zero-filled by capdec_cast_bf16) -- the host side picks this kernel only when that holds
// (launch_gemm_bf16s).
#include "common.h"
#include <type_traits>

#include "gemm_common.h"

namespace capdec {

typedef __bf16 bf16x8d __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int DBK = 64;
#ifndef CAPDEC_DMA_MGROUP
#define CAPDEC_DMA_MGROUP 8
#endif
constexpr int DMA_MGROUP = CAPDEC_DMA_MGROUP;  // row tiles that share a sweep over the column tiles (L2 reuse of their A panel)

#define CAPDEC_GLDS16(gptr, lptr) \
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(gptr), \
                                     (void __attribute__((address_space(3)))*)(lptr), 16, 0, 0)

// per-lane source pointers of one operand's 4 DMA instructions (this wave's 4 KiB of the image), k-tile 0.
// An image of R rows is R / 8 pieces of 1 KiB: k-contiguous, piece c = rows 8c .. 8c + 7; k-major, the image is
// R / 128 sub-images [64 k][128 rows] of 16 pieces each, piece c = k-rows 4 (c % 16) .. + 3 of sub-image c / 16.
template <bool KC>
struct DmaSrc {
    // 32-bit element offsets from a wave-uniform base (the host side checks that the matrix spans < 2^31 elements):
    // half the address registers of per-lane 64-bit pointers -- the difference between three and four waves per SIMD
    unsigned off[4];   // k-contiguous: row * ld + piece (k-tile 0);  k-major: the (clamped) column
    int krow[4];       // k-major only: the k-row inside the tile

    __device__ __forceinline__ void init(int ld, int row0, int nrows, int wave, int lane) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = wave * 4 + j;  // 1 KiB piece of the image
            if constexpr (KC) {
                const int row = c * 8 + (lane >> 3);
                const int src = (lane & 7) ^ ((row >> 1) & 7);
                off[j] = (unsigned)min(row0 + row, nrows - 1) * (unsigned)ld + src * 8;
            } else {
                const int kr = (c & 15) * 4 + (lane >> 4);
                const int src = (lane & 15) ^ (((kr & 3) << 2) | ((kr >> 2) & 3));
                krow[j] = kr;
                off[j] = min(row0 + (c >> 4) * 128 + src * 8, ld - 8);
            }
        }
    }
    // source of piece j in the k-tile starting at k0 (k-major: rows past K are clamped to the last one -- the
    // partner's zero padding kills them)
    __device__ __forceinline__ const __bf16* src(const __bf16* g, int ld, int j, int k0, int K) const {
        if constexpr (KC) return g + k0 + off[j];
        else return g + ((unsigned)min(k0 + krow[j], K - 1) * (unsigned)ld + off[j]);
    }
};

// NBUF = 1: one LDS image pair (32 KB, three workgroups per CU), load -> barrier -> compute -> barrier per k-tile.
// NBUF = 2: two image pairs (64 KB, two workgroups per CU), the DMA of tile t + 1 is in flight while tile t is
//           computed, one barrier per k-tile.
// SPLITK: the kernel contains the in-launch split-K combine.  Its code (cold, but part of the same register
// allocation) costs ~20 VGPRs -- the difference between three and four resident workgroups per CU for the one-image
// 128 x 128 form -- so unsplit launches get an instantiation without it.
// The transposing read as inline assembly: the BUILTIN gets a compiler-inserted s_waitcnt vmcnt(0) whenever an LDS DMA is
// in flight (see attention_bf16.hip), which in the two-image form below stood between the request for k-tile kt + 1 and
// the first fragment read of k-tile kt -- the prefetch of every product with a k-major operand was fully serialised
// (found in round 3 by reading the ISA).  The callers wait with their own lgkmcnt tied to the results.
__device__ __forceinline__ s16x4 pair_read_tr16(unsigned addr) {  // inline assembly: see attention_bf16.hip
    s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}

template <int OFF>
__device__ __forceinline__ s16x4 pair_read_tr16_at(unsigned addr) {  // the same with an immediate byte offset
    s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}

// waves per SIMD of the one-image unsplit form: four until round 3 (the fourth workgroup hid the element-form epilogue); with the
// vector epilogue three and four measure alike (A/B within 0.5 %) and three needs no spills
#ifndef CAPDEC_DMA_WPE
#define CAPDEC_DMA_WPE 3
#endif
template <bool AKC, bool BKC, int NBUF, bool SPLITK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NBUF == 1 ? (SPLITK ? 3 : CAPDEC_DMA_WPE) : 2)))
void gemm_bf16s_dma_kernel(GemmParams p) {
    constexpr int BM = 128, BN = BM, TM = 2, TN = 2;
    constexpr int NT = 256, WN = BN / (32 * TN);  // waves along N; two wave rows
    constexpr int D_IMG = BM * DBK * 2;                       // bytes per operand image
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NBUF * 2 * D_IMG];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int wrow = (wave / WN) * (BM / 2), wcol = (wave % WN) * (32 * TN);
    const int nwg = gridDim.x;
    int wg = blockIdx.x;
    {   // XCD-aware tile order: a contiguous run of tiles per XCD (blocks b and b + 8 share one)
        const int q = nwg >> 3, r = nwg & 7, xcd = wg & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wg >> 3);
    }
    int tile_m, tile_n;
    {
        const int tiles_m = nwg / p.tiles_n;
        const int per_group = DMA_MGROUP * p.tiles_n;
        const int gid = wg / per_group, first_m = gid * DMA_MGROUP;
        const int gsz = min(tiles_m - first_m, DMA_MGROUP);
        const int in_g = wg - gid * per_group;
        tile_m = first_m + in_g % gsz;
        tile_n = in_g / gsz;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kbeg = blockIdx.z * p.ksplit;
    const int kend = min(p.K, kbeg + p.ksplit);
    const int nk = (kend - kbeg + DBK - 1) / DBK;
    const __bf16* A16 = reinterpret_cast<const __bf16*>(p.A);
    const __bf16* B16 = reinterpret_cast<const __bf16*>(p.B);

    DmaSrc<AKC> sa;
    DmaSrc<BKC> sb;
    sa.init(p.lda, m0, p.M, wave, lane);
    sb.init(p.ldb, n0, p.N, wave, lane);
    unsigned char* img_a = smem + wave * 4096;          // this wave's DMA destinations (wave-uniform)
    unsigned char* img_b = smem + D_IMG + wave * 4096;
    const unsigned lds0 = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)smem;

    // fragment read offsets (bytes inside an image)
    //   k-contiguous: row * 128 + (((ks * 2) ^ c) * 16), c = half ^ ((l31 >> 1) & 7)
    //   k-major (transposing read): lane 4q + p of 16-lane group g: k-row (g >> 1) * 8 + rd * 4 + q (+ ks * 16),
    //       piece ch = row / 8 + (g & 1) * 2 + (p >> 1), slot ch ^ ((q << 2) | (((g >> 1) * 2 + rd) & 3)), + 8 * (p & 1)
    const int g4 = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
    int offa[TM][2], offb[TN][2];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        if constexpr (AKC) {
            offa[i][0] = (wrow + i * 32 + l31) * 128;
            offa[i][1] = (half ^ ((l31 >> 1) & 7)) * 16;
        } else {
#pragma unroll
            for (int rd = 0; rd < 2; ++rd) {
                const int kr = (g4 >> 1) * 8 + rd * 4 + q4;
                const int r0 = wrow + i * 32;
                const int ch = (r0 & 127) / 8 + (g4 & 1) * 2 + (p4 >> 1);
                offa[i][rd] = (r0 >> 7) * 16384 + kr * 256 + ((ch ^ ((q4 << 2) | (((g4 >> 1) * 2 + rd) & 3))) * 16) + 8 * (p4 & 1);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        if constexpr (BKC) {
            offb[j][0] = (wcol + j * 32 + l31) * 128;
            offb[j][1] = (half ^ ((l31 >> 1) & 7)) * 16;
        } else {
#pragma unroll
            for (int rd = 0; rd < 2; ++rd) {
                const int kr = (g4 >> 1) * 8 + rd * 4 + q4;
                const int c0 = wcol + j * 32;
                const int ch = (c0 & 127) / 8 + (g4 & 1) * 2 + (p4 >> 1);
                offb[j][rd] = (c0 >> 7) * 16384 + kr * 256 + ((ch ^ ((q4 << 2) | (((g4 >> 1) * 2 + rd) & 3))) * 16) + 8 * (p4 & 1);
            }
        }
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const bool do_colsum = p.colsum != nullptr && tile_n == 0;  // colsum[m] = sum_k A(m, k): the bias gradient
    float csum = 0.f;
    float cs8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // k-major A: 8 columns x this thread's k-rows

    if constexpr (NBUF == 2) {
        if (nk > 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) CAPDEC_GLDS16(sa.src(A16, p.lda, j, kbeg, p.K), img_a + j * 1024);
#pragma unroll
            for (int j = 0; j < 4; ++j) CAPDEC_GLDS16(sb.src(B16, p.ldb, j, kbeg, p.K), img_b + j * 1024);
        }
    }
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = NBUF == 2 ? (kt & 1) : 0;
        if constexpr (NBUF == 1) {
            const int k0 = kbeg + kt * DBK;
#pragma unroll
            for (int j = 0; j < 4; ++j) CAPDEC_GLDS16(sa.src(A16, p.lda, j, k0, p.K), img_a + j * 1024);
#pragma unroll
            for (int j = 0; j < 4; ++j) CAPDEC_GLDS16(sb.src(B16, p.ldb, j, k0, p.K), img_b + j * 1024);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tile kt has landed (this wave's part) ...
            __syncthreads();                                   // ... everyone's; and everyone is done reading tile kt - 1
            if (kt + 1 < nk) {
                const int k1 = kbeg + (kt + 1) * DBK;
                const int nb = (buf ^ 1) * 2 * D_IMG;
#pragma unroll
                for (int j = 0; j < 4; ++j) CAPDEC_GLDS16(sa.src(A16, p.lda, j, k1, p.K), img_a + nb + j * 1024);
#pragma unroll
                for (int j = 0; j < 4; ++j) CAPDEC_GLDS16(sb.src(B16, p.ldb, j, k1, p.K), img_b + nb + j * 1024);
            }
        }
        const unsigned char* As = smem + buf * 2 * D_IMG;
        const unsigned char* Bs = As + D_IMG;
        const unsigned as0 = lds0 + buf * 2 * D_IMG, bs0 = as0 + D_IMG;
        // k-major operands: transposing reads in inline assembly, one 16-deep k-step ahead of the products that use them
        s16x4 ta[2][TM][2], tb[2][TN][2];
        auto issue_tr = [&](auto ks_c, int pb) {
            constexpr int KS = decltype(ks_c)::value;
            if constexpr (!AKC) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    ta[pb][i][0] = pair_read_tr16_at<KS * 4096>(as0 + offa[i][0]);
                    ta[pb][i][1] = pair_read_tr16_at<KS * 4096>(as0 + offa[i][1]);
                }
            }
            if constexpr (!BKC) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    tb[pb][j][0] = pair_read_tr16_at<KS * 4096>(bs0 + offb[j][0]);
                    tb[pb][j][1] = pair_read_tr16_at<KS * 4096>(bs0 + offb[j][1]);
                }
            }
        };
        auto retire_tr = [&](int pb) {  // every LDS read issued so far has returned
            static_assert(TM == 2 && TN == 2, "the operand lists below name 2 x 2 fragments");
            if constexpr (!AKC)
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ta[pb][0][0]), "+v"(ta[pb][0][1]), "+v"(ta[pb][1][0]), "+v"(ta[pb][1][1]));
            if constexpr (!BKC)
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(tb[pb][0][0]), "+v"(tb[pb][0][1]), "+v"(tb[pb][1][0]), "+v"(tb[pb][1][1]));
        };
        auto k_step = [&](auto ks_c) {
            constexpr int ks = decltype(ks_c)::value;
            bf16x8d af[TM], bfr[TN];
            constexpr int pb = ks & 1;
            if constexpr (!AKC || !BKC) {
                retire_tr(pb);
                if constexpr (ks + 1 < DBK / 16) issue_tr(std::integral_constant<int, ks + 1>{}, pb ^ 1);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if constexpr (AKC) {
                    af[i] = *reinterpret_cast<const bf16x8d*>(As + offa[i][0] + (offa[i][1] ^ (ks * 32)));
                } else {
                    const s16x4 lo = ta[pb][i][0], hi = ta[pb][i][1];
                    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    af[i] = __builtin_bit_cast(bf16x8d, v);
                }
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if constexpr (BKC) {
                    bfr[j] = *reinterpret_cast<const bf16x8d*>(Bs + offb[j][0] + (offb[j][1] ^ (ks * 32)));
                } else {
                    const s16x4 lo = tb[pb][j][0], hi = tb[pb][j][1];
                    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    bfr[j] = __builtin_bit_cast(bf16x8d, v);
                }
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        };
        static_assert(DBK == 64, "four 16-deep k-steps per k-tile");
        if constexpr (!AKC || !BKC) issue_tr(std::integral_constant<int, 0>{}, 0);
        k_step(std::integral_constant<int, 0>{});
        k_step(std::integral_constant<int, 1>{});
        k_step(std::integral_constant<int, 2>{});
        k_step(std::integral_constant<int, 3>{});
        if (do_colsum) {
            if constexpr (AKC) {  // row sums of the A image (pieces in swizzled order: a sum does not care)
                if (tid < BM) {
#pragma unroll
                    for (int s = 0; s < 8; ++s) {
                        const bf16x8d v = *reinterpret_cast<const bf16x8d*>(As + tid * 128 + s * 16);
#pragma unroll
                        for (int e = 0; e < 8; ++e) csum += (float)v[e];
                    }
                }
            } else {  // [64 k][128] sub-image t >> 8: thread t owns piece t & 15 (8 columns) of k-rows ((t & 255) >> 4) + 16 u
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int r = ((tid & 255) >> 4) + 16 * u;
                    const int slot = (tid & 15) ^ (((r & 3) << 2) | ((r >> 2) & 3));
                    const bf16x8d v = *reinterpret_cast<const bf16x8d*>(As + (tid >> 8) * 16384 + r * 256 + slot * 16);
#pragma unroll
                    for (int e = 0; e < 8; ++e) cs8[e] += (float)v[e];
                }
            }
        }
        if constexpr (NBUF == 1) __syncthreads();  // every wave is done with the images before the next tile's DMA lands
    }
    if constexpr (NBUF == 2) __syncthreads();  // the epilogue reuses the LDS

    if constexpr (!AKC) {
        if (do_colsum) {  // 16 threads hold partial sums of each column: combine through LDS in a fixed order
            float* red = reinterpret_cast<float*>(smem);  // [16][BM]
#pragma unroll
            for (int e = 0; e < 8; ++e) red[((tid & 255) >> 4) * BM + (tid >> 8) * 128 + (tid & 15) * 8 + e] = cs8[e];
            __syncthreads();
            if (tid < BM) {
#pragma unroll
                for (int u = 0; u < 16; ++u) csum += red[u * BM + tid];
            }
            __syncthreads();
        }
    }
    // split-K: only the last-arriving slice of this tile goes on, with the slice-ordered sum in acc / csum
    if constexpr (SPLITK) {
        if (gridDim.z > 1 && !splitk_combine<TM, TN, NT>(p, acc, csum, do_colsum, tile_m * p.tiles_n + tile_n, m0, BM, tid,
                                                          reinterpret_cast<int*>(smem)))
            return;
    }
    if (do_colsum && tid < BM && m0 + tid < p.M) p.colsum[m0 + tid] = csum;
    if (p.ce_part) ce_tile_partials<TM, TN, BM, BN, WN, NT>(p, acc, reinterpret_cast<float*>(smem), m0, n0, tile_n, tid);
    if (p.vec_epi) {  // the vector form of the epilogue (gemm_common.h): 32 rows at a time through 8 KB of LDS per wave
        __syncthreads();  // every wave is done with whatever used the LDS last
        epilogue_store_vec<TM, TN, 1>(p, acc, m0 + wrow, n0 + wcol, lane, reinterpret_cast<float*>(smem + wave * 8192));
        return;
    }
    epilogue_store<TM, TN>(p, acc, m0 + wrow, n0 + wcol, l31, half);
}

// ---- 256 x 256 block, two wave groups one phase apart ("pair" kernel) -------------------------------------------
// A first 256 x 256 form (the loop above with eight waves) kept its waves in step: all read fragments, then all
// multiply, and every k-tile drained the DMA counter at its barrier -- the LDS pipe and the matrix pipe took turns and
// the block ran at 0.85x of the 128 x 128 form.  Same images here (two 64-deep image pairs, 128 KB, one workgroup per
// CU; DmaSrc with 8 waves covers 256 rows), other loop:
//   * a k-tile is four phases -- read fragments of its first 32-deep half, multiply them, read the second half,
//     multiply -- each closed by a BARE s_barrier (__syncthreads() would drain vmcnt);
//   * the upper wave row (waves 4-7; each shares a SIMD with a wave of the lower row) runs ONE BARRIER BEHIND the
//     lower one, so on every SIMD one wave multiplies while the other reads;
//   * the next k-tile is requested as soon as its slot is free -- by the early group at the start of a k-tile, by the
//     late group one phase earlier -- and waited for (vmcnt(0): a wave has only that one request outstanding) four
//     phases later, just before the barrier after which the first wave may read it.
// Order rules: a wave retires its own fragment reads (lgkmcnt(0)) before the barrier that ends a read phase, so when
// the slot of k-tile t - 1 is refilled every read of it has returned; and every wave has waited for its pieces of
// k-tile t + 1 before the barrier that opens the first read phase of t + 1 in either group.
// A 32-deep stage ring (three stages in flight, 64-byte image rows) was tried first: 500 TFLOP/s on the products with
// two k-contiguous operands whatever its depth -- DMA pieces of 16 rows x 64 B cost the address path twice what
// 8 rows x 128 B do -- and on par with the 128 x 128 form where an operand is k-major.
template <bool AKC, bool BKC, bool SPLITK>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2)))
void gemm_bf16s_pair_kernel(GemmParams p) {
    constexpr int BM = 256, BN = 256, TM = 4, TN = 2, NT = 512, WN = 4;
    constexpr int D_IMG = BM * DBK * 2;  // 32 KiB per operand and k-tile
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * 2 * D_IMG];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int wrow = (wave / WN) * (BM / 2), wcol = (wave % WN) * (32 * TN);
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
        const int per_group = DMA_MGROUP * p.tiles_n;
        const int gid = wg / per_group, first_m = gid * DMA_MGROUP;
        const int gsz = min(tiles_m - first_m, DMA_MGROUP);
        const int in_g = wg - gid * per_group;
        tile_m = first_m + in_g % gsz;
        tile_n = in_g / gsz;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kbeg = blockIdx.z * p.ksplit;
    const int kend = min(p.K, kbeg + p.ksplit);
    const int nk = (kend - kbeg + DBK - 1) / DBK;
    const __bf16* A16 = reinterpret_cast<const __bf16*>(p.A);
    const __bf16* B16 = reinterpret_cast<const __bf16*>(p.B);

    DmaSrc<AKC> sa;
    DmaSrc<BKC> sb;
    sa.init(p.lda, m0, p.M, wave, lane);
    sb.init(p.ldb, n0, p.N, wave, lane);
    unsigned char* img_a = smem + wave * 4096;          // this wave's DMA destinations (wave-uniform)
    unsigned char* img_b = smem + D_IMG + wave * 4096;
    const unsigned lds0 = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)smem;

    const int g4 = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
    int offa[TM][2], offb[TN][2];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        if constexpr (AKC) {
            offa[i][0] = (wrow + i * 32 + l31) * 128;
            offa[i][1] = (half ^ ((l31 >> 1) & 7)) * 16;
        } else {
#pragma unroll
            for (int rd = 0; rd < 2; ++rd) {
                const int kr = (g4 >> 1) * 8 + rd * 4 + q4;
                const int r0 = wrow + i * 32;
                const int ch = (r0 & 127) / 8 + (g4 & 1) * 2 + (p4 >> 1);
                offa[i][rd] = (r0 >> 7) * 16384 + kr * 256 + ((ch ^ ((q4 << 2) | (((g4 >> 1) * 2 + rd) & 3))) * 16) + 8 * (p4 & 1);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        if constexpr (BKC) {
            offb[j][0] = (wcol + j * 32 + l31) * 128;
            offb[j][1] = (half ^ ((l31 >> 1) & 7)) * 16;
        } else {
#pragma unroll
            for (int rd = 0; rd < 2; ++rd) {
                const int kr = (g4 >> 1) * 8 + rd * 4 + q4;
                const int c0 = wcol + j * 32;
                const int ch = (c0 & 127) / 8 + (g4 & 1) * 2 + (p4 >> 1);
                offb[j][rd] = (c0 >> 7) * 16384 + kr * 256 + ((ch ^ ((q4 << 2) | (((g4 >> 1) * 2 + rd) & 3))) * 16) + 8 * (p4 & 1);
            }
        }
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const bool do_colsum = p.colsum != nullptr && tile_n == 0;  // colsum[m] = sum_k A(m, k): the bias gradient
    float csum = 0.f;
    float cs8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    auto request = [&](int kt) {  // this wave's pieces of k-tile kt -> slot kt & 1
        const int k0 = kbeg + kt * DBK, slot = (kt & 1) * 2 * D_IMG;
#pragma unroll
        for (int j = 0; j < 4; ++j) CAPDEC_GLDS16(sa.src(A16, p.lda, j, k0, p.K), img_a + slot + j * 1024);
#pragma unroll
        for (int j = 0; j < 4; ++j) CAPDEC_GLDS16(sb.src(B16, p.ldb, j, k0, p.K), img_b + slot + j * 1024);
    };
    // fragments of one 32-deep half (two 16-deep k-steps) of the k-tile in slot `buf`, then its products
    bf16x8d af[2][TM], bfr[2][TN];
    s16x4 ta[2][TM][2], tb[2][TN][2];
    auto read_half = [&](int buf, int h) {
        const unsigned char* As = smem + buf * 2 * D_IMG;
        const unsigned char* Bs = As + D_IMG;
        const unsigned as0 = lds0 + buf * 2 * D_IMG, bs0 = as0 + D_IMG;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int ks = 2 * h + u;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if constexpr (AKC) {
                    af[u][i] = *reinterpret_cast<const bf16x8d*>(As + offa[i][0] + (offa[i][1] ^ (ks * 32)));
                } else {
                    ta[u][i][0] = pair_read_tr16(as0 + offa[i][0] + ks * 4096);
                    ta[u][i][1] = pair_read_tr16(as0 + offa[i][1] + ks * 4096);
                }
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if constexpr (BKC) {
                    bfr[u][j] = *reinterpret_cast<const bf16x8d*>(Bs + offb[j][0] + (offb[j][1] ^ (ks * 32)));
                } else {
                    tb[u][j][0] = pair_read_tr16(bs0 + offb[j][0] + ks * 4096);
                    tb[u][j][1] = pair_read_tr16(bs0 + offb[j][1] + ks * 4096);
                }
            }
        }
        if (do_colsum && h == 0) {  // once per k-tile
            if constexpr (AKC) {  // row sums of the A image (pieces in swizzled order: a sum does not care)
                if (tid < BM) {
#pragma unroll
                    for (int s = 0; s < 8; ++s) {
                        const bf16x8d v = *reinterpret_cast<const bf16x8d*>(As + tid * 128 + s * 16);
#pragma unroll
                        for (int e = 0; e < 8; ++e) csum += (float)v[e];
                    }
                }
            } else {  // [64 k][128] sub-image t >> 8: thread t owns piece t & 15 (8 columns) of k-rows ((t & 255) >> 4) + 16 u
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int r = ((tid & 255) >> 4) + 16 * u;
                    const int slot = (tid & 15) ^ (((r & 3) << 2) | ((r >> 2) & 3));
                    const bf16x8d v = *reinterpret_cast<const bf16x8d*>(As + (tid >> 8) * 16384 + r * 256 + slot * 16);
#pragma unroll
                    for (int e = 0; e < 8; ++e) cs8[e] += (float)v[e];
                }
            }
        }
    };
    auto retire_reads = [&]() {  // every fragment read issued so far has returned (also the inline-assembly ones)
        if constexpr (!AKC)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ta[0][0][0]), "+v"(ta[0][0][1]), "+v"(ta[0][1][0]), "+v"(ta[0][1][1]),
                         "+v"(ta[0][2][0]), "+v"(ta[0][2][1]), "+v"(ta[0][3][0]), "+v"(ta[0][3][1]), "+v"(ta[1][0][0]),
                         "+v"(ta[1][0][1]), "+v"(ta[1][1][0]), "+v"(ta[1][1][1]), "+v"(ta[1][2][0]), "+v"(ta[1][2][1]),
                         "+v"(ta[1][3][0]), "+v"(ta[1][3][1]));
        if constexpr (!BKC)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(tb[0][0][0]), "+v"(tb[0][0][1]), "+v"(tb[0][1][0]), "+v"(tb[0][1][1]),
                         "+v"(tb[1][0][0]), "+v"(tb[1][0][1]), "+v"(tb[1][1][0]), "+v"(tb[1][1][1]));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    auto multiply = [&]() {
        __builtin_amdgcn_sched_barrier(0);  // (register-only MFMAs would otherwise be hoisted into the read phase)
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if constexpr (!AKC) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const s16x4 lo = ta[u][i][0], hi = ta[u][i][1];
                    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    af[u][i] = __builtin_bit_cast(bf16x8d, v);
                }
            }
            if constexpr (!BKC) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const s16x4 lo = tb[u][j][0], hi = tb[u][j][1];
                    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    bfr[u][j] = __builtin_bit_cast(bf16x8d, v);
                }
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[u][i], bfr[u][j], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };
#define PAIR_BARRIER() asm volatile("s_barrier" ::: "memory")
#define PAIR_LANDED() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

    if (nk > 0) request(0);
    if (nk > 1) request(1);
    if (nk > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // k-tile 0 (this wave's pieces); k-tile 1 may fly on
    else PAIR_LANDED();
    if (late) PAIR_BARRIER();  // the stagger
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        PAIR_BARRIER();                                   // X0
        if (!late && kt >= 1 && kt + 1 < nk) request(kt + 1);  // early group: the slot of k-tile kt - 1 is free now
        read_half(buf, 0);
        retire_reads();
        PAIR_BARRIER();                                   // Y0
        multiply();
        PAIR_BARRIER();                                   // X1
        read_half(buf, 1);
        if (late && kt + 1 < nk) PAIR_LANDED();           // late group: k-tile kt + 1 before the barrier that lets the early group read it
        retire_reads();
        PAIR_BARRIER();                                   // Y1
        if (late && kt + 2 < nk) request(kt + 2);         // late group: every read of this slot has returned
        multiply();
        if (!late && kt + 1 < nk) PAIR_LANDED();          // early group: k-tile kt + 1, before its own next X0
    }
    if (!late) PAIR_BARRIER();  // the stagger, paid back
    PAIR_BARRIER();             // everyone is done with the images: the epilogue reuses the LDS
#undef PAIR_BARRIER
#undef PAIR_LANDED

    if constexpr (!AKC) {
        if (do_colsum) {  // 16 threads hold partial sums of each column: combine through LDS in a fixed order
            float* red = reinterpret_cast<float*>(smem);  // [16][BM]
#pragma unroll
            for (int e = 0; e < 8; ++e) red[((tid & 255) >> 4) * BM + (tid >> 8) * 128 + (tid & 15) * 8 + e] = cs8[e];
            __syncthreads();
            if (tid < BM) {
#pragma unroll
                for (int u = 0; u < 16; ++u) csum += red[u * BM + tid];
            }
            __syncthreads();
        }
    }
    if constexpr (SPLITK) {
        if (gridDim.z > 1 && !splitk_combine<TM, TN, NT>(p, acc, csum, do_colsum, tile_m * p.tiles_n + tile_n, m0, BM, tid,
                                                          reinterpret_cast<int*>(smem)))
            return;
    }
    if (do_colsum && tid < BM && m0 + tid < p.M) p.colsum[m0 + tid] = csum;
    if (p.ce_part) ce_tile_partials<TM, TN, BM, BN, WN, NT>(p, acc, reinterpret_cast<float*>(smem), m0, n0, tile_n, tid);
    if (p.vec_epi) {  // one workgroup per CU: nothing hides the epilogue, so the vector form matters most here
        __syncthreads();
        epilogue_store_vec<TM, TN, 2>(p, acc, m0 + wrow, n0 + wcol, lane, reinterpret_cast<float*>(smem + wave * 16384));
        return;
    }
    epilogue_store<TM, TN>(p, acc, m0 + wrow, n0 + wcol, l31, half);
}

template <bool SPLITK>
static void launch_pair(const GemmParams& p, bool AKC, bool BKC, dim3 grid, hipStream_t st) {
    const dim3 block(512);
    if (AKC && BKC) hipLaunchKernelGGL((gemm_bf16s_pair_kernel<true, true, SPLITK>), grid, block, 0, st, p);
    else if (AKC) hipLaunchKernelGGL((gemm_bf16s_pair_kernel<true, false, SPLITK>), grid, block, 0, st, p);
    else if (BKC) hipLaunchKernelGGL((gemm_bf16s_pair_kernel<false, true, SPLITK>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((gemm_bf16s_pair_kernel<false, false, SPLITK>), grid, block, 0, st, p);
}

// The DMA kernel needs: zero padding behind the reduction extent of every k-contiguous operand up to the next
// multiple of 64 (its pitch says whether the padding exists; capdec_cast_bf16 writes the zeros -- a caller whose wide
// pitch means something else says so with capdec_epilogue.bf16_unpadded), and a reduction tail that at least one
// operand zeroes (two k-major operands have no padding: K must then be a multiple of 64).
bool gemm_bf16s_dma_ok(const GemmParams& p, bool AKC, bool BKC) {
    const int k64 = cdiv(p.K, DBK) * DBK;
    if (AKC && p.lda < k64) return false;
    if (BKC && p.ldb < k64) return false;
    if (!AKC && !BKC && p.K % DBK) return false;
    if (p.K % DBK && ((AKC && (p.unpadded & 1)) || (BKC && (p.unpadded & 2)))) return false;
    // 32-bit element offsets inside each operand
    if ((uint64_t)(AKC ? p.M : p.K) * (uint64_t)p.lda >= (1ull << 31) || (uint64_t)(BKC ? p.N : p.K) * (uint64_t)p.ldb >= (1ull << 31))
        return false;
    return p.lda >= 8 && p.ldb >= 8;
}

template <int NBUF, bool SPLITK>
static void launch_nbuf(const GemmParams& p, bool AKC, bool BKC, dim3 grid, hipStream_t st) {
    const dim3 block(256);
    if (AKC && BKC) hipLaunchKernelGGL((gemm_bf16s_dma_kernel<true, true, NBUF, SPLITK>), grid, block, 0, st, p);
    else if (AKC) hipLaunchKernelGGL((gemm_bf16s_dma_kernel<true, false, NBUF, SPLITK>), grid, block, 0, st, p);
    else if (BKC) hipLaunchKernelGGL((gemm_bf16s_dma_kernel<false, true, NBUF, SPLITK>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((gemm_bf16s_dma_kernel<false, false, NBUF, SPLITK>), grid, block, 0, st, p);
}

bool gemm_bf16s_p8_ok(const GemmParams& p, bool AKC, bool BKC);              // gemm_bf16s_p8.hip
void launch_gemm_bf16s_p8(const GemmParams& p, bool AKC, bool BKC, dim3 grid, hipStream_t st);

// bm = 128: nbuf image pairs (1 or 2); bm = 256: the 256 x 256 pair kernel, or (nbuf = 2, i.e. k-tile field 32, where
// the operands allow it) the eight-phase kernel of gemm_bf16s_p8.hip
void launch_gemm_bf16s_dma(const GemmParams& p, bool AKC, bool BKC, int bm, int nbuf, dim3 grid, hipStream_t st) {
    if (bm == 256) {
        if (nbuf == 2 && gemm_bf16s_p8_ok(p, AKC, BKC)) launch_gemm_bf16s_p8(p, AKC, BKC, grid, st);
        else if (grid.z > 1) launch_pair<true>(p, AKC, BKC, grid, st);
        else launch_pair<false>(p, AKC, BKC, grid, st);
    } else if (nbuf == 2) launch_nbuf<2, true>(p, AKC, BKC, grid, st);
    else if (grid.z > 1) launch_nbuf<1, true>(p, AKC, BKC, grid, st);
    else launch_nbuf<1, false>(p, AKC, BKC, grid, st);
}

}  // namespace capdec
