// bf16-compute GEMM: fp32 operands in HBM are rounded to bf16 (RNE) on their way into LDS,
// multiplied on v_mfma_f32_32x32x16_bf16 (16x the fp32 matrix rate) with fp32 accumulation, and the
// same fused fp32 epilogue as gemm_f32.hip writes fp32 results.  Selected by dtype = CAPDEC_BF16 in
// capdec_gemm ("autocast"-style mixed precision: master weights, activations, normalisation, softmax,
// loss and optimizer stay fp32).
//
// Block tile BM x BN x 64, 4 waves (2 x 2).  Both operands land in LDS as [row][k] bf16 with rows
// padded to 144 B, whatever their layout in memory (k-contiguous operands: float4 along k -> one
// ds_write_b64; row-contiguous operands: four float4 of consecutive k rows are transposed in
// registers -> four ds_write_b64).  A lane's MFMA fragment (8 consecutive k of one row) is then one
// ds_read_b128, conflict-free across each 16-lane group thanks to the 144-B stride.
#include "common.h"
#include "gemm_common.h"

namespace capdec {


typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int HBK = 64;       // k-tile
constexpr int HLD = HBK + 8;  // LDS row stride in bf16 elements (144 B)

__device__ __forceinline__ bf16x4 cvt4(float a, float b, float c, float d) {
    bf16x4 r;
    r[0] = (__bf16)a; r[1] = (__bf16)b; r[2] = (__bf16)c; r[3] = (__bf16)d;
    return r;
}

template <bool VEC>
__device__ __forceinline__ float4 hload4(const float* ptr, bool ok, int rem) {
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!ok || rem <= 0) return r;
    if (VEC && rem >= 4) return *reinterpret_cast<const float4*>(ptr);
    r.x = ptr[0];
    if (rem > 1) r.y = ptr[1];
    if (rem > 2) r.z = ptr[2];
    if (rem > 3) r.w = ptr[3];
    return r;
}

template <int BM, int BN, bool AK, bool BKM, bool VEC>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmParams p) {
    constexpr int NT = 256;
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int A_SZ = BM * HLD, B_SZ = BN * HLD;      // bf16 elements per buffer
    constexpr int AK_V4 = BM * HBK / 4 / NT;              // float4 per thread, k-contiguous loader
    constexpr int BK_V4 = BN * HBK / 4 / NT;
    constexpr int A_UNITS = (HBK / 4) * (BM / 4);         // 4x4 transpose units, row-contiguous loader
    constexpr int B_UNITS = (HBK / 4) * (BN / 4);
    constexpr int A_UPT = (A_UNITS + NT - 1) / NT, B_UPT = (B_UNITS + NT - 1) / NT;  // units per thread
    constexpr int A_REGS = AK ? AK_V4 : 4 * A_UPT, B_REGS = BKM ? BK_V4 : 4 * B_UPT;
    __shared__ __attribute__((aligned(16))) __bf16 smem[2 * (A_SZ + B_SZ)];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int wrow = (wave >> 1) * (BM / 2), wcol = (wave & 1) * (BN / 2);
    const int nwg = gridDim.x;
    int wg = blockIdx.x;
    {
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
    const int nk = (kend - kbeg + HBK - 1) / HBK;
    const bool interior = VEC && (m0 + BM <= p.M) && (n0 + BN <= p.N) && ((kend - kbeg) % HBK == 0);

    float4 ra[A_REGS], rb[B_REGS];

    auto gload = [&](int k0) {
        if (AK) {
#pragma unroll
            for (int i = 0; i < AK_V4; ++i) {
                const int f = tid + i * NT;
                const int row = f / (HBK / 4), kq = f % (HBK / 4);
                const int gm = m0 + row, gk = k0 + kq * 4;
                const float* ptr = p.A + (size_t)gm * p.lda + gk;
                ra[i] = interior ? *reinterpret_cast<const float4*>(ptr) : hload4<VEC>(ptr, gm < p.M, kend - gk);
            }
        } else {
#pragma unroll
            for (int u = 0; u < A_UPT; ++u) {
                const int unit = tid + u * NT;
                if (unit < A_UNITS) {
                    const int a_kq = unit / (BM / 4), a_rq = unit % (BM / 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int gk = k0 + a_kq * 4 + j, gm = m0 + a_rq * 4;
                        const float* ptr = p.A + (size_t)gk * p.lda + gm;
                        ra[u * 4 + j] = interior ? *reinterpret_cast<const float4*>(ptr)
                                                 : hload4<VEC>(ptr, gk < kend, p.M - gm);
                    }
                }
            }
        }
        if (BKM) {
#pragma unroll
            for (int i = 0; i < BK_V4; ++i) {
                const int f = tid + i * NT;
                const int row = f / (HBK / 4), kq = f % (HBK / 4);
                const int gn = n0 + row, gk = k0 + kq * 4;
                const float* ptr = p.B + (size_t)gn * p.ldb + gk;
                rb[i] = interior ? *reinterpret_cast<const float4*>(ptr) : hload4<VEC>(ptr, gn < p.N, kend - gk);
            }
        } else {
#pragma unroll
            for (int u = 0; u < B_UPT; ++u) {
                const int unit = tid + u * NT;
                if (unit < B_UNITS) {
                    const int b_kq = unit / (BN / 4), b_rq = unit % (BN / 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int gk = k0 + b_kq * 4 + j, gn = n0 + b_rq * 4;
                        const float* ptr = p.B + (size_t)gk * p.ldb + gn;
                        rb[u * 4 + j] = interior ? *reinterpret_cast<const float4*>(ptr)
                                                 : hload4<VEC>(ptr, gk < kend, p.N - gn);
                    }
                }
            }
        }
    };
    auto lstore = [&](int buf) {
        __bf16* As = smem + buf * (A_SZ + B_SZ);
        __bf16* Bs = As + A_SZ;
        if (AK) {
#pragma unroll
            for (int i = 0; i < AK_V4; ++i) {
                const int f = tid + i * NT;
                const int row = f / (HBK / 4), kq = f % (HBK / 4);
                *reinterpret_cast<bf16x4*>(&As[row * HLD + kq * 4]) = cvt4(ra[i].x, ra[i].y, ra[i].z, ra[i].w);
            }
        } else {
#pragma unroll
            for (int u = 0; u < A_UPT; ++u) {
                const int unit = tid + u * NT;
                if (unit < A_UNITS) {
                    const int a_kq = unit / (BM / 4), a_rq = unit % (BM / 4);
                    const float4* r = &ra[u * 4];
                    __bf16* dst = &As[(a_rq * 4) * HLD + a_kq * 4];
                    *reinterpret_cast<bf16x4*>(dst + 0 * HLD) = cvt4(r[0].x, r[1].x, r[2].x, r[3].x);
                    *reinterpret_cast<bf16x4*>(dst + 1 * HLD) = cvt4(r[0].y, r[1].y, r[2].y, r[3].y);
                    *reinterpret_cast<bf16x4*>(dst + 2 * HLD) = cvt4(r[0].z, r[1].z, r[2].z, r[3].z);
                    *reinterpret_cast<bf16x4*>(dst + 3 * HLD) = cvt4(r[0].w, r[1].w, r[2].w, r[3].w);
                }
            }
        }
        if (BKM) {
#pragma unroll
            for (int i = 0; i < BK_V4; ++i) {
                const int f = tid + i * NT;
                const int row = f / (HBK / 4), kq = f % (HBK / 4);
                *reinterpret_cast<bf16x4*>(&Bs[row * HLD + kq * 4]) = cvt4(rb[i].x, rb[i].y, rb[i].z, rb[i].w);
            }
        } else {
#pragma unroll
            for (int u = 0; u < B_UPT; ++u) {
                const int unit = tid + u * NT;
                if (unit < B_UNITS) {
                    const int b_kq = unit / (BN / 4), b_rq = unit % (BN / 4);
                    const float4* r = &rb[u * 4];
                    __bf16* dst = &Bs[(b_rq * 4) * HLD + b_kq * 4];
                    *reinterpret_cast<bf16x4*>(dst + 0 * HLD) = cvt4(r[0].x, r[1].x, r[2].x, r[3].x);
                    *reinterpret_cast<bf16x4*>(dst + 1 * HLD) = cvt4(r[0].y, r[1].y, r[2].y, r[3].y);
                    *reinterpret_cast<bf16x4*>(dst + 2 * HLD) = cvt4(r[0].z, r[1].z, r[2].z, r[3].z);
                    *reinterpret_cast<bf16x4*>(dst + 3 * HLD) = cvt4(r[0].w, r[1].w, r[2].w, r[3].w);
                }
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const bool do_colsum = (!AK) && p.colsum != nullptr && tile_n == 0;
    float csum = 0.f;

    if (nk > 0) {
        gload(kbeg);
        lstore(0);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload(kbeg + (kt + 1) * HBK);
        const __bf16* As = smem + buf * (A_SZ + B_SZ);
        const __bf16* Bs = As + A_SZ;
#pragma unroll
        for (int ks = 0; ks < HBK / 16; ++ks) {
            bf16x8 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[i] = *reinterpret_cast<const bf16x8*>(&As[(wrow + i * 32 + l31) * HLD + ks * 16 + half * 8]);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bf[j] = *reinterpret_cast<const bf16x8*>(&Bs[(wcol + j * 32 + l31) * HLD + ks * 16 + half * 8]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (do_colsum && tid < BM) {
#pragma unroll
            for (int k = 0; k < HBK; ++k) csum += (float)As[tid * HLD + k];
        }
        if (kt + 1 < nk) lstore(buf ^ 1);
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

template <int BM, int BN>
static void launch_bf16(const GemmParams& p, bool AK, bool BKM, bool vec, dim3 grid, hipStream_t st) {
#define CAPDEC_LAUNCH(ak, bk, v) hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, ak, bk, v>), grid, dim3(256), 0, st, p)
    if (vec) {
        if (AK && BKM) CAPDEC_LAUNCH(true, true, true);
        else if (AK && !BKM) CAPDEC_LAUNCH(true, false, true);
        else if (!AK && !BKM) CAPDEC_LAUNCH(false, false, true);
        else CAPDEC_LAUNCH(false, true, true);
    } else {
        if (AK && BKM) CAPDEC_LAUNCH(true, true, false);
        else if (AK && !BKM) CAPDEC_LAUNCH(true, false, false);
        else if (!AK && !BKM) CAPDEC_LAUNCH(false, false, false);
        else CAPDEC_LAUNCH(false, true, false);
    }
#undef CAPDEC_LAUNCH
}

// The bf16 product is operand-traffic bound (fp32 operands, 16x matrix rate): prefer the largest tile
// that still gives every CU >= 2 workgroups; weight-gradient shapes (small output, long K) split K.
GemmPlan plan_gemm_bf16(int M, int N, int K, const GemmForce& force) {
    if (force.bm) {
        const int split = force.split > 0 ? force.split : 1;
        const int ks = cdiv(cdiv(K, split), HBK) * HBK;
        const int bm = force.bm == 128 && force.bn == 128 ? 128 : 64;
        return GemmPlan{bm, bm, HBK, cdiv(K, ks), ks};
    }
    // calibrated on profiles/r01_gemm_sweep_bf16.txt: short-K products want many small workgroups
    // (latency hiding by occupancy); long-K products (weight gradients, the vocab input gradient)
    // want 128x128 tiles and enough K slices for >= 1024 workgroups.
    int bm = 64, split = 1;
    if (K >= 4096) {
        const long t128 = (long)cdiv(M, 128) * cdiv(N, 128);
        if (t128 >= 64) {
            bm = 128;
            while (t128 * split < 1024 && K / (split * 2) >= 512 && split < 32) split *= 2;
        } else {  // small outputs (512x512, 1536x512 weight gradients): 64x64 tiles, K cut down to 256-deep slices
            const long t64 = (long)cdiv(M, 64) * cdiv(N, 64);
            while (t64 * split < 768 && K / (split * 2) >= 256 && split < 32) split *= 2;
        }
    } else {
        const long tiles = (long)cdiv(M, 64) * cdiv(N, 64);
        while (tiles * split < 256 && K / (split * 2) >= 256 && split < 16) split *= 2;
    }
    const int ks = cdiv(cdiv(K, split), HBK) * HBK;
    return GemmPlan{bm, bm, HBK, cdiv(K, ks), ks};
}

void launch_gemm_bf16(const GemmParams& p, const GemmPlan& pl, bool AK, bool BKM, bool vec, hipStream_t st) {
    dim3 grid(cdiv(p.M, pl.bm) * p.tiles_n, 1, pl.split);
    if (pl.bm == 128) launch_bf16<128, 128>(p, AK, BKM, vec, grid, st);
    else launch_bf16<64, 64>(p, AK, BKM, vec, grid, st);
}

}  // namespace capdec
