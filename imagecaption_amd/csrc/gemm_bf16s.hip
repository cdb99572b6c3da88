// bf16-STORAGE GEMM (dtype = CAPDEC_BF16S): both operands live in memory as bf16 -- half the operand bytes of the
// fp32-operand kernels (gemm_bf16.hip rounds fp32 operands on their way into LDS and is bound by reading them
// through L2: 0.12-0.15 of the bf16 matrix peak at the C3 shapes) and no conversion in the loaders.
// Either operand may be stored reduction-index-contiguous ([M][ld], 16-byte pieces go straight to LDS) or
// reduction-index-major ([K][ld], the layout the weight-gradient and input-gradient products see): there a thread
// loads an 8 (k) x 8 (m) block as eight 16-byte row pieces, transposes it in registers (32 byte-permutes) and
// writes eight 16-byte k-runs -- so NO transposed copy of any tensor is ever made in memory; one bf16 copy per
// tensor serves the forward, input-gradient and weight-gradient products alike.
// v_mfma_f32_32x32x16_bf16, fp32 accumulate, the same fused fp32 epilogue, in-launch split-K, bias-gradient sums and
// cross-entropy partials as the other GEMM kernels; fp32 results.
//
// Block tile BM x BN x 64, 4 waves (2 x 2); LDS rows padded to 144 B (conflict-free ds_read_b128 fragments and
// 16-byte writes); register prefetch one k-tile ahead.
//   64 x 64, 128 x 128: double-buffered LDS, one barrier per k-tile.  A wave's 64 x 64 part of the 128-tile needs
//       one ds_read_b128 per MFMA: 4 SIMDs x 8 LDS cycles per 32 MFMA cycles -- the LDS pipe is exactly saturated, and
//       the measured rate stops at ~0.23 of the matrix peak.
//   256 x 128: a wave owns 128 x 64 (6 fragment reads per 8 MFMAs: LDS pipe 75 % busy at full matrix rate); ONE LDS
//       image (55 KB, two workgroups per CU), the register prefetch covers the store-side barrier pair.
#include "common.h"
#include "gemm_common.h"

namespace capdec {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int SBK = 64;       // k-tile
constexpr int SLD = SBK + 8;  // LDS row stride in bf16 elements (144 B)

// One operand tile (R rows of the output dimension x 64 k) on its way global -> registers -> LDS.
//   KC = true : stored [row][ld] (k contiguous): R * 8 pieces of 16 B, R * 8 / 256 per thread.
//   KC = false: stored [k][ld] (row contiguous): R blocks of 8 k x 8 rows, one per thread of `R` threads starting at
//               thread FIRST (A takes the low threads, B the high ones: disjoint when both are k-major).
template <int R, bool KC, int FIRST>
struct TileMover {
    static constexpr int NREG = KC ? R * 8 / 256 : 8;
    u32x4 r[NREG];

    __device__ __forceinline__ void load(const __bf16* __restrict__ g, int ld, int row0, int nrows, int k0, int kend, int tid) {
        if constexpr (KC) {
#pragma unroll
            for (int i = 0; i < NREG; ++i) {
                const int f = tid + i * 256;
                const int row = row0 + (f >> 3), gk = k0 + (f & 7) * 8;
                u32x4 v = {0u, 0u, 0u, 0u};
                if (row < nrows && gk < kend) {
                    v = *reinterpret_cast<const u32x4*>(g + (size_t)row * ld + gk);
                    const int nv = kend - gk;  // elements of this piece inside K (the row pitch covers all 8)
                    if (nv < 8) {
                        unsigned int w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int c = 0; c < 4; ++c) w[c] = 2 * c >= nv ? 0u : (2 * c + 1 >= nv ? (w[c] & 0xffffu) : w[c]);
                        v = u32x4{w[0], w[1], w[2], w[3]};
                    }
                }
                r[i] = v;
            }
        } else {
            const int u = tid - FIRST;
            if (u >= 0 && u < R) {
                const int kb = u & 7, row = row0 + (u >> 3) * 8;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int gk = k0 + kb * 8 + e;
                    r[e] = (row < nrows && gk < kend) ? *reinterpret_cast<const u32x4*>(g + (size_t)gk * ld + row) : u32x4{0u, 0u, 0u, 0u};
                }
            }
        }
    }

    __device__ __forceinline__ void store(__bf16* lds, int tid) const {
        if constexpr (KC) {
#pragma unroll
            for (int i = 0; i < NREG; ++i) {
                const int f = tid + i * 256;
                *reinterpret_cast<u32x4*>(&lds[(f >> 3) * SLD + (f & 7) * 8]) = r[i];
            }
        } else {
            const int u = tid - FIRST;
            if (u >= 0 && u < R) {
                const int kb = u & 7, rb = (u >> 3) * 8;
                // r[e] = 8 rows at k = e; wanted: per row j, the 8 k values: dword c = (r[2c].h[j], r[2c+1].h[j])
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    unsigned int w[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const unsigned int lo = r[2 * c][j >> 1], hi = r[2 * c + 1][j >> 1];
                        w[c] = (j & 1) ? __builtin_amdgcn_perm(hi, lo, 0x07060302u) : __builtin_amdgcn_perm(hi, lo, 0x05040100u);
                    }
                    *reinterpret_cast<u32x4*>(&lds[(rb + j) * SLD + kb * 8]) = u32x4{w[0], w[1], w[2], w[3]};
                }
            }
        }
    }
};

template <int BM, int BN, bool AKC, bool BKC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BM * BN >= 256 * 128 ? 2 : 1)))
void gemm_bf16s_kernel(GemmParams p) {
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int A_SZ = BM * SLD, B_SZ = BN * SLD;  // bf16 elements per buffer
    constexpr bool DB = BM * BN < 256 * 128;         // double-buffered LDS
    constexpr int NBUF = DB ? 2 : 1;
    __shared__ __attribute__((aligned(16))) __bf16 smem[NBUF * (A_SZ + B_SZ)];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int wrow = (wave >> 1) * (BM / 2), wcol = (wave & 1) * (BN / 2);
    const int nwg = gridDim.x;
    int wg = blockIdx.x;
    {   // XCD-aware tile order: a contiguous run of tiles per XCD (blocks b and b + 8 share one)
        const int q = nwg >> 3, r = nwg & 7, xcd = wg & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wg >> 3);
    }
    int tile_m, tile_n;
    {
        const int tiles_m = nwg / p.tiles_n;
        const int per_group = 8 * p.tiles_n;
        const int gid = wg / per_group, first_m = gid * 8;
        const int gsz = min(tiles_m - first_m, 8);
        const int in_g = wg - gid * per_group;
        tile_m = first_m + in_g % gsz;
        tile_n = in_g / gsz;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kbeg = blockIdx.z * p.ksplit;
    const int kend = min(p.K, kbeg + p.ksplit);
    const int nk = (kend - kbeg + SBK - 1) / SBK;
    const __bf16* A16 = reinterpret_cast<const __bf16*>(p.A);
    const __bf16* B16 = reinterpret_cast<const __bf16*>(p.B);

    TileMover<BM, AKC, 0> ma;
    TileMover<BN, BKC, 256 - BN> mb;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const bool do_colsum = p.colsum != nullptr && tile_n == 0;  // colsum[m] = sum_k A(m, k): the bias gradient
    float csum = 0.f;

    if (nk > 0) {
        ma.load(A16, p.lda, m0, p.M, kbeg, kend, tid);
        mb.load(B16, p.ldb, n0, p.N, kbeg, kend, tid);
        ma.store(smem, tid);
        mb.store(smem + A_SZ, tid);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = DB ? (kt & 1) : 0;
        if (kt + 1 < nk) {
            ma.load(A16, p.lda, m0, p.M, kbeg + (kt + 1) * SBK, kend, tid);
            mb.load(B16, p.ldb, n0, p.N, kbeg + (kt + 1) * SBK, kend, tid);
        }
        const __bf16* As = smem + buf * (A_SZ + B_SZ);
        const __bf16* Bs = As + A_SZ;
#pragma unroll
        for (int ks = 0; ks < SBK / 16; ++ks) {
            bf16x8 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[i] = *reinterpret_cast<const bf16x8*>(&As[(wrow + i * 32 + l31) * SLD + ks * 16 + half * 8]);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bf[j] = *reinterpret_cast<const bf16x8*>(&Bs[(wcol + j * 32 + l31) * SLD + ks * 16 + half * 8]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (do_colsum && tid < BM) {
#pragma unroll
            for (int k8 = 0; k8 < SBK / 8; ++k8) {
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(&As[tid * SLD + k8 * 8]);
#pragma unroll
                for (int e = 0; e < 8; ++e) csum += (float)v[e];
            }
        }
        if (kt + 1 < nk) {
            if constexpr (!DB) __syncthreads();  // every wave is done reading the one image
            __bf16* nb = smem + (DB ? (buf ^ 1) : 0) * (A_SZ + B_SZ);
            ma.store(nb, tid);
            mb.store(nb + A_SZ, tid);
        }
        __syncthreads();
    }

    // split-K: only the last-arriving slice of this tile goes on, with the slice-ordered sum in acc / csum
    if (gridDim.z > 1 && !splitk_combine<TM, TN>(p, acc, csum, do_colsum, tile_m * p.tiles_n + tile_n, m0, BM, tid,
                                                  reinterpret_cast<int*>(smem)))
        return;
    if (do_colsum && tid < BM && m0 + tid < p.M) p.colsum[m0 + tid] = csum;
    if (p.ce_part) ce_tile_partials<TM, TN, BM, BN>(p, acc, reinterpret_cast<float*>(smem), m0, n0, tile_n, tid);
    if (p.vec_epi) {  // the vector form of the epilogue (gemm_common.h): 32 rows of the wave block at a time through LDS
        __syncthreads();  // every wave is done with whatever used the LDS last
        epilogue_store_vec<TM, TN, 1>(p, acc, m0 + wrow, n0 + wcol, lane, reinterpret_cast<float*>(smem) + wave * (32 * 32 * TN));
        return;
    }
    epilogue_store<TM, TN>(p, acc, m0 + wrow, n0 + wcol, l31, half);
}

// Plan.  128 x 128 tiles (the LDS-DMA kernel where the operands allow it) unless a dimension is too short to fill
// them, then 64 x 64; K slices (combined in the launch) until the grid reaches ~512 workgroups while a slice keeps
// >= 512 of K (measured on the C3 shapes: profiles/r02_gemm_bf16s_sweep.txt).  The k-tile field of a plan selects the
// 128 x 128 kernel: default = LDS-DMA with one image pair, 32 = LDS-DMA with two image pairs, 16 = register-staged.
GemmPlan plan_gemm_bf16s(int M, int N, int K, const GemmForce& force) {
    if (force.bm) {
        const int split = force.split > 0 ? force.split : 1;
        const int ks = cdiv(cdiv(K, split), SBK) * SBK;
        if (force.bm == 256)  // k-tile field 32 with 256 x 256: the eight-phase kernel (gemm_bf16s_p8.hip)
            return GemmPlan{256, force.bn == 256 ? 256 : 128, force.bn == 256 && force.bk == 32 ? 32 : SBK, cdiv(K, ks), ks};
        const int bm = force.bm == 128 && force.bn == 128 ? 128 : 64;
        return GemmPlan{bm, bm, force.bk ? force.bk : SBK, cdiv(K, ks), ks};
    }
    // the eight-phase 256 x 256 kernel (one workgroup per CU) wherever its tiles fill the rounds they take: from 640 tiles up, and
    // from 160 up when the last round is at least 70 % full (measured with the store-streaming epilogue, nt products with K =
    // 1024: ahead of the 128 x 128 forms at 240, 480, 640, 800 and 1176 tiles, behind them at 80 and at 320 = 1.25 rounds;
    // operands it cannot take fall back in the launcher)
    {
        const long t = (long)cdiv(M, 256) * cdiv(N, 256), rounds = (t + 255) / 256;
        if (K >= 256 && (t >= 640 || (t >= 160 && 10 * t >= 7 * 256 * rounds))) return GemmPlan{256, 256, 32, 1, cdiv(K, SBK) * SBK};
    }
    const int bm = (M >= 96 && N >= 96) ? 128 : 64;
    const long tiles = (long)cdiv(M, bm) * cdiv(N, bm);
    int split = 1;
    while (tiles * split < 512 && K / (split * 2) >= (bm == 128 ? 512 : 256) && split < 32) split *= 2;
    const int ks = cdiv(cdiv(K, split), SBK) * SBK;
    return GemmPlan{bm, bm, SBK, cdiv(K, ks), ks};
}

template <int BM, int BN>
static void launch_layout(const GemmParams& p, bool AKC, bool BKC, dim3 grid, hipStream_t st) {
    if (AKC && BKC) hipLaunchKernelGGL((gemm_bf16s_kernel<BM, BN, true, true>), grid, dim3(256), 0, st, p);
    else if (AKC) hipLaunchKernelGGL((gemm_bf16s_kernel<BM, BN, true, false>), grid, dim3(256), 0, st, p);
    else if (BKC) hipLaunchKernelGGL((gemm_bf16s_kernel<BM, BN, false, true>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((gemm_bf16s_kernel<BM, BN, false, false>), grid, dim3(256), 0, st, p);
}

void launch_gemm_bf16s_dma(const GemmParams& p, bool AKC, bool BKC, int bm, int nbuf, dim3 grid, hipStream_t st);

void launch_gemm_bf16s(const GemmParams& p, const GemmPlan& pl, bool AKC, bool BKC, hipStream_t st) {
    dim3 grid(cdiv(p.M, pl.bm) * p.tiles_n, 1, pl.split);
    if (pl.bm == pl.bn && pl.bm >= 128 && pl.bk != 16 && gemm_bf16s_dma_ok(p, AKC, BKC))
        launch_gemm_bf16s_dma(p, AKC, BKC, pl.bm, pl.bk == 32 ? 2 : 1, grid, st);
    else if (pl.bm == 256) launch_layout<256, 128>(p, AKC, BKC, grid, st);
    else if (pl.bm == 128) launch_layout<128, 128>(p, AKC, BKC, grid, st);
    else launch_layout<64, 64>(p, AKC, BKC, grid, st);
}

}  // namespace capdec
