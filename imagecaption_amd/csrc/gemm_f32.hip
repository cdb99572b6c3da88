// fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 FMA chains) with a fused
// bias / GELU / GELU' / dropout / residual epilogue, a fused bias-gradient column sum for
// weight-gradient products, and deterministic split-K.  See include/capdec.h (capdec_gemm).
//
// Block tile BM x BN x 16, 4 waves (2 x 2), each wave owns (BM/2) x (BN/2) as 32x32 accumulators.
// LDS images (double-buffered, register-prefetched one k-tile ahead):
//   k-contiguous operand (row-major, reduction index contiguous): [k/4][row][4]  -> one ds_read_b128
//       per lane yields the operands of four consecutive MFMAs (lane half h owns k = 8j+4h..8j+4h+3);
//       rows padded by 2 float4 so the 8-lane ds_write_b128 groups and 16-lane ds_read_b128 groups
//       hit distinct banks.
//   row-contiguous operand (reduction index is the storage row): [k][row] -> four ds_read_b32.
#include <mutex>
#include <string>
#include <unordered_map>

#include "common.h"
#include <cstdlib>
#include "gemm_common.h"

namespace capdec {

GemmForce g_gemm_force = {0, 0, 0, 0};

// Tuned plans (capdec_gemm_plan_set): what a caller measured to be the fastest (tile, split, k-tile) for one
// product (transposes, shape, compute type) on the device it runs on.  A hit replaces the cost model's choice;
// capdec_gemm_tune's process-wide force still comes first.
static std::mutex g_plan_mu;
static std::unordered_map<std::string, GemmForce> g_plans;
static std::string plan_key(int ta, int tb, int M, int N, int K, int dtype) {
    char buf[96];
    snprintf(buf, sizeof(buf), "%d:%d:%d:%d:%d:%d", ta ? 1 : 0, tb ? 1 : 0, M, N, K, dtype);
    return buf;
}
static GemmForce resolve_force(int ta, int tb, int M, int N, int K, int dtype) {
    if (g_gemm_force.bm) return g_gemm_force;
    std::lock_guard<std::mutex> lock(g_plan_mu);
    auto it = g_plans.find(plan_key(ta, tb, M, N, K, dtype));
    return it == g_plans.end() ? g_gemm_force : it->second;
}

bool gemm_plan_forced(int ta, int tb, int M, int N, int K, int dtype) { return resolve_force(ta, tb, M, N, K, dtype).bm != 0; }

template <bool VEC>
__device__ __forceinline__ float4 load4(const float* ptr, bool ok, int rem) {
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!ok || rem <= 0) return r;
    if (VEC && rem >= 4) return *reinterpret_cast<const float4*>(ptr);
    r.x = ptr[0];
    if (rem > 1) r.y = ptr[1];
    if (rem > 2) r.z = ptr[2];
    if (rem > 3) r.w = ptr[3];
    return r;
}

// AK / BKM: operand is k-contiguous in memory (A stored [M,K]; B stored [N,K]).
// Register budget = the main loop's: the occupancy each tile shape reaches there (LDS- or register-bound) is
// declared, so that the rarely executed split-K combine at the end is scheduled inside it instead of widening
// the allocation of every workgroup.
constexpr int gemm_f32_waves(int bm, int bn, int bk) {
    return bm * bn == 64 * 64 ? (bk == 16 ? 8 : 4) : bm * bn == 128 * 64 ? (bk == 16 ? 5 : 3) : 1;
}

// CE: the A loaders read LOGITS and turn them into cross-entropy gradients on their way into LDS (see
// gemm_common.h); a template flag so that the plain instantiations carry none of its registers.
// One BM x BN output tile (and K slice blockIdx.z) of one product: the body of gemm_f32_kernel, and of
// gemm_f32_grouped_kernel, whose workgroups each pick a product from a table first.  wg_raw / nwg: this workgroup's
// number among the product's nwg tiles (wg_raw & 7 = the XCD it runs on, for the tile order).
template <int BM, int BN, int GBK, bool AK, bool BKM, bool VEC, bool CE = false>
__device__ __forceinline__ void gemm_f32_tile(const GemmParams& p, const int wg_raw, const int nwg) {
    constexpr int NT = 256;
    constexpr int TM = BM / 64, TN = BN / 64;  // 32x32 accumulators per wave in each direction
    constexpr int A_LD = AK ? (BM + 2) * 4 : (BM + 4);  // floats per k-quad (AK) or per k (row-contig)
    constexpr int B_LD = BKM ? (BN + 2) * 4 : (BN + 4);
    constexpr int A_SZ = AK ? (GBK / 4) * A_LD : GBK * A_LD;
    constexpr int B_SZ = BKM ? (GBK / 4) * B_LD : GBK * B_LD;
    constexpr int A_V4 = BM * GBK / 4 / NT;
    constexpr int B_V4 = BN * GBK / 4 / NT;
    static_assert(A_V4 >= 1 && B_V4 >= 1, "tile too small for 256 threads");
    // the vector epilogue's wave-private regions (32 rows of a wave block each) + the cross-entropy combine area (BM x 2 float4)
    constexpr int EPI_SZ = 4 * 32 * 32 * TN + BM * 2 * 4 + BM;  // wave regions, CE partials, the tile's targets
    constexpr int SMEM_SZ = 2 * (A_SZ + B_SZ) > EPI_SZ ? 2 * (A_SZ + B_SZ) : EPI_SZ;
    __shared__ __attribute__((aligned(16))) float smem[SMEM_SZ];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int wrow = (wave >> 1) * (BM / 2), wcol = (wave & 1) * (BN / 2);

    // XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch), so give each XCD
    // a contiguous run of tiles (neighbouring tiles share an A row panel in that XCD's L2).
    int wg = wg_raw;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = wg & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wg >> 3);
    }
    // grouped order inside an XCD's run: 8 row panels x all column panels, row panel fastest, so
    // the ~160 workgroups resident on an XCD share few A and B panels in its 4 MiB L2
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
    const int nk = (kend - kbeg + GBK - 1) / GBK;

    float4 ra[A_V4], rb[B_V4];
    // Interior workgroups (whole tile in bounds, aligned, K range a multiple of the k-tile) stream
    // their operands through pre-computed per-thread pointers: one global_load_dwordx4 + one pointer
    // bump per float4, no bounds logic in the loop (the guarded loader's branches cost as many issue
    // slots as the MFMAs themselves at 4-5 waves per SIMD).
    const bool interior = VEC && (m0 + BM <= p.M) && (n0 + BN <= p.N) && ((kend - kbeg) % GBK == 0);
    const float* pa[A_V4];
    const float* pb[B_V4];
#pragma unroll
    for (int i = 0; i < A_V4; ++i) {
        const int f = tid + i * NT;
        if (AK) pa[i] = p.A + (size_t)(m0 + f / (GBK / 4)) * p.lda + kbeg + (f % (GBK / 4)) * 4;
        else pa[i] = p.A + (size_t)(kbeg + f / (BM / 4)) * p.lda + m0 + (f % (BM / 4)) * 4;
    }
#pragma unroll
    for (int i = 0; i < B_V4; ++i) {
        const int f = tid + i * NT;
        if (BKM) pb[i] = p.B + (size_t)(n0 + f / (GBK / 4)) * p.ldb + kbeg + (f % (GBK / 4)) * 4;
        else pb[i] = p.B + (size_t)(kbeg + f / (BN / 4)) * p.ldb + n0 + (f % (BN / 4)) * 4;
    }
    const size_t step_a = AK ? (size_t)GBK : (size_t)GBK * p.lda;
    const size_t step_b = BKM ? (size_t)GBK : (size_t)GBK * p.ldb;
    // fused cross-entropy backward (p.ce_lse != NULL): A holds logits.  Each float4 of A is four vocabulary columns
    // of one logits row; its row's lse and target travel with it and the transform runs on the way into LDS.
    constexpr bool ce = CE;
    float ce_l[CE ? A_V4 : 1];
    int ce_t[CE ? A_V4 : 1];
    auto ce_fetch = [&](int k0) {  // AK: the row is this thread's for the whole K loop; else it is the k index
#pragma unroll
        for (int i = 0; i < (CE ? A_V4 : 0); ++i) {
            const int f = tid + i * NT;
            const int row = AK ? m0 + f / (GBK / 4) : k0 + f / (BM / 4);
            const bool ok = AK ? row < p.M : row < kend;
            ce_l[i] = ok ? p.ce_lse[row] : 0.f;
            ce_t[i] = ok ? ce_target(p, row, AK ? p.K : p.M) : -1;
        }
    };
    auto ce_apply = [&](int k0) {
#pragma unroll
        for (int i = 0; i < (CE ? A_V4 : 0); ++i) {
            const int f = tid + i * NT;
            const int row = AK ? m0 + f / (GBK / 4) : k0 + f / (BM / 4);
            const int col0 = AK ? k0 + (f % (GBK / 4)) * 4 : m0 + (f % (BM / 4)) * 4;
            const bool row_ok = AK ? row < p.M : row < kend;
            ra[i] = ce_dlogits4(p, ra[i], ce_l[i], ce_t[i], col0, (AK ? kend : p.M) - col0, row_ok);
        }
    };
    auto gload_fast = [&]() {
#pragma unroll
        for (int i = 0; i < A_V4; ++i) {
            ra[i] = *reinterpret_cast<const float4*>(pa[i]);
            pa[i] += step_a;
        }
#pragma unroll
        for (int i = 0; i < B_V4; ++i) {
            rb[i] = *reinterpret_cast<const float4*>(pb[i]);
            pb[i] += step_b;
        }
    };
    auto gload_guarded = [&](int k0) {
#pragma unroll
        for (int i = 0; i < A_V4; ++i) {
            const int f = tid + i * NT;
            if (AK) {
                const int row = f / (GBK / 4), kq = f % (GBK / 4);
                const int gm = m0 + row, gk = k0 + kq * 4;
                ra[i] = load4<VEC>(p.A + (size_t)gm * p.lda + gk, gm < p.M, kend - gk);
            } else {
                const int k = f / (BM / 4), rq = f % (BM / 4);
                const int gk = k0 + k, gm = m0 + rq * 4;
                ra[i] = load4<VEC>(p.A + (size_t)gk * p.lda + gm, gk < kend, p.M - gm);
            }
        }
#pragma unroll
        for (int i = 0; i < B_V4; ++i) {
            const int f = tid + i * NT;
            if (BKM) {
                const int row = f / (GBK / 4), kq = f % (GBK / 4);
                const int gn = n0 + row, gk = k0 + kq * 4;
                rb[i] = load4<VEC>(p.B + (size_t)gn * p.ldb + gk, gn < p.N, kend - gk);
            } else {
                const int k = f / (BN / 4), rq = f % (BN / 4);
                const int gk = k0 + k, gn = n0 + rq * 4;
                rb[i] = load4<VEC>(p.B + (size_t)gk * p.ldb + gn, gk < kend, p.N - gn);
            }
        }
    };
    auto gload = [&](int k0) {
        if (interior) gload_fast();
        else gload_guarded(k0);
        if (!AK && ce) ce_fetch(k0);
    };
    auto lstore = [&](int buf, int k0) {
        float* As = smem + buf * (A_SZ + B_SZ);
        float* Bs = As + A_SZ;
        if (ce) ce_apply(k0);
#pragma unroll
        for (int i = 0; i < A_V4; ++i) {
            const int f = tid + i * NT;
            if (AK) {
                const int row = f / (GBK / 4), kq = f % (GBK / 4);
                *reinterpret_cast<float4*>(&As[kq * A_LD + row * 4]) = ra[i];
            } else {
                const int k = f / (BM / 4), rq = f % (BM / 4);
                *reinterpret_cast<float4*>(&As[k * A_LD + rq * 4]) = ra[i];
            }
        }
#pragma unroll
        for (int i = 0; i < B_V4; ++i) {
            const int f = tid + i * NT;
            if (BKM) {
                const int row = f / (GBK / 4), kq = f % (GBK / 4);
                *reinterpret_cast<float4*>(&Bs[kq * B_LD + row * 4]) = rb[i];
            } else {
                const int k = f / (BN / 4), rq = f % (BN / 4);
                *reinterpret_cast<float4*>(&Bs[k * B_LD + rq * 4]) = rb[i];
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

    if (AK && ce) ce_fetch(kbeg);
    if (nk > 0) {
        gload(kbeg);
        lstore(0, kbeg);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload(kbeg + (kt + 1) * GBK);
        const float* As = smem + buf * (A_SZ + B_SZ);
        const float* Bs = As + A_SZ;
#pragma unroll
        for (int kb = 0; kb < GBK / 8; ++kb) {
            float4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = wrow + i * 32 + l31;
                if (AK) {
                    af[i] = *reinterpret_cast<const float4*>(&As[(kb * 2 + half) * A_LD + row * 4]);
                } else {
                    const float* s = &As[(kb * 8 + half * 4) * A_LD + row];
                    af[i] = make_float4(s[0], s[A_LD], s[2 * A_LD], s[3 * A_LD]);
                }
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = wcol + j * 32 + l31;
                if (BKM) {
                    bf[j] = *reinterpret_cast<const float4*>(&Bs[(kb * 2 + half) * B_LD + col * 4]);
                } else {
                    const float* s = &Bs[(kb * 8 + half * 4) * B_LD + col];
                    bf[j] = make_float4(s[0], s[B_LD], s[2 * B_LD], s[3 * B_LD]);
                }
            }
            // k-step outermost: consecutive MFMAs hit different accumulators (no back-to-back dependency)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const float av = c == 0 ? af[i].x : c == 1 ? af[i].y : c == 2 ? af[i].z : af[i].w;
                        const float bv = c == 0 ? bf[j].x : c == 1 ? bf[j].y : c == 2 ? bf[j].z : bf[j].w;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                    }
        }
        if (do_colsum && tid < BM) {
#pragma unroll
            for (int k = 0; k < GBK; ++k) csum += As[k * A_LD + tid];
        }
        if (kt + 1 < nk) lstore(buf ^ 1, kbeg + (kt + 1) * GBK);
        __syncthreads();
    }

    // split-K: only the last-arriving slice of this tile goes on, with the slice-ordered sum in acc / csum
    if (gridDim.z > 1 && !splitk_combine<TM, TN>(p, acc, csum, do_colsum, tile_m * p.tiles_n + tile_n, m0, BM, tid,
                                                  reinterpret_cast<int*>(smem)))
        return;
    if (do_colsum && tid < BM && m0 + tid < p.M) p.colsum[m0 + tid] = csum;
    if (p.vec_epi) {  // the vector form of the epilogue (gemm_common.h): 32 rows of the wave block at a time through LDS
        __syncthreads();  // every wave is done with whatever used the LDS last
        if (p.ce_part)  // the logits product: the softmax partials come out of the same loop
            epilogue_store_vec_ce<TM, TN, BM>(p, acc, m0, n0, wrow, wcol, wave & 1, lane, smem + wave * (32 * 32 * TN),
                                              reinterpret_cast<float4*>(smem + 4 * 32 * 32 * TN), tile_n, tid);
        else
            epilogue_store_vec<TM, TN, 1>(p, acc, m0 + wrow, n0 + wcol, lane, smem + wave * (32 * 32 * TN));
        return;
    }
    if (p.ce_part) ce_tile_partials<TM, TN, BM, BN>(p, acc, smem, m0, n0, tile_n, tid);
    epilogue_store<TM, TN>(p, acc, m0 + wrow, n0 + wcol, l31, half);
}

template <int BM, int BN, int GBK, bool AK, bool BKM, bool VEC, bool CE = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(gemm_f32_waves(BM, BN, GBK))))
void gemm_f32_kernel(GemmParams p) {
    gemm_f32_tile<BM, BN, GBK, AK, BKM, VEC, CE>(p, blockIdx.x, gridDim.x);
}

// Several independent products of one layout in ONE launch (capdec_gemm_grouped): workgroup b belongs to the product
// whose block range [first[g], first[g + 1]) holds b (ranges start at multiples of 8, so a workgroup's number inside
// its product keeps the XCD parity the tile order relies on; the few padding workgroups of a range leave at once).
// Small products -- a decoder layer's weight gradients: 64 - 256 tiles each, which alone need split-K and its slab
// round trips to fill 256 CUs -- then run unsplit as one chip-filling grid.
template <int BM, int BN, int GBK, bool AK, bool BKM>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(gemm_f32_waves(BM, BN, GBK))))
void gemm_f32_grouped_kernel(GemmGroupParams g) {
    int i = 0;
#pragma unroll
    for (int k = 1; k < CAPDEC_GEMM_GROUP_MAX; ++k)
        if (k < g.n && (int)blockIdx.x >= g.first[k]) i = k;
    const int local = (int)blockIdx.x - g.first[i];
    const int tiles = g.tiles[i];
    if (local >= tiles) return;
    gemm_f32_tile<BM, BN, GBK, AK, BKM, true>(g.p[i], local, tiles);
}

// ---- host side: tile / split-K choice ---------------------------------------------------------
// MFMA-bound cost model, calibrated on tools/gemm_sweep.py (profiles/r01_gemm_sweep.txt).  A CU
// with r co-resident workgroups runs r waves per SIMD; one wave per SIMD cannot cover its own
// barrier / load stalls, so the matrix pipe's utilisation u(r) rises with r.  Cost = rounds of
// workgroups per CU x work per workgroup / u, plus the split-K partial round trip.
static GemmPlan plan_gemm(int M, int N, int K, const GemmForce& force) {
    static const int cand[4][2] = {{128, 128}, {128, 64}, {64, 128}, {64, 64}};
    static const double util[5] = {0.0, 0.55, 0.80, 0.97, 1.0};
    const int bk = force.bk ? force.bk : 16;
    if (force.bm) {
        int split = force.split > 0 ? force.split : 1;
        int ks = cdiv(cdiv(K, split), bk) * bk;
        return GemmPlan{force.bm, force.bn, bk, cdiv(K, ks), ks};
    }
    GemmPlan best{128, 128, bk, 1, K};
    double best_cost = 1e300;
    for (int c = 0; c < 4; ++c) {
        const int bm = cand[c][0], bn = cand[c][1];
        const int tiles = cdiv(M, bm) * cdiv(N, bn);
        const double units = (bm / 64) * (bn / 64);
        const int rmax = c == 0 ? 3 : 4;
        for (int split = 1; split <= 32; split *= 2) {
            int ks = cdiv(cdiv(K, split), bk) * bk;
            if (split > 1 && ks < 128) break;
            const int s_eff = cdiv(K, ks);
            const long blocks = (long)tiles * s_eff;
            const int per_cu = cdiv((int)(blocks > 2000000 ? 2000000 : blocks), 256);
            const int r = per_cu < rmax ? per_cu : rmax;
            double cost = per_cu * units * (cdiv(ks, 16) + 4.0) / util[r];
            if (s_eff > 1) cost += 1.3e-6 * (double)M * N * s_eff + 8.0;  // partial write + reduce pass
            cost *= 1.0 + 0.01 * c;  // ties -> larger tiles (less L2 traffic)
            if (cost < best_cost) {
                best_cost = cost;
                best = GemmPlan{bm, bn, bk, s_eff, ks};
            }
        }
    }
    return best;
}

template <int BM, int BN, int BK>
static void launch_cfg(const GemmParams& p, bool AK, bool BKM, bool vec, dim3 grid, hipStream_t st) {
    if (p.ce_lse) {  // the two backward products of the vocabulary projection: dlogits^T h and dlogits W (checked by the caller)
        if (AK) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, BK, true, false, true, true>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, BK, false, false, true, true>), grid, dim3(256), 0, st, p);
        return;
    }
#define CAPDEC_LAUNCH(ak, bk, v) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, BK, ak, bk, v>), grid, dim3(256), 0, st, p)
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

}  // namespace capdec

using namespace capdec;

extern "C" int capdec_gemm_tune(int bm, int bn, int split, int bk) {
    const bool ok = ((bm == 0) || (((bm == 64 || bm == 128) && (bn == 64 || bn == 128) || (bm == 256 && (bn == 128 || bn == 256))) &&
                                   split >= 1 && split <= 64)) &&
                    (bk == 0 || bk == 16 || bk == 32);
    CAPDEC_REQUIRE(ok, "capdec_gemm_tune: unsupported tile %dx%d split %d bk %d", bm, bn, split, bk);
    g_gemm_force = GemmForce{bm, bn, split, bk};
    return CAPDEC_OK;
}

// a plan whose tile count exceeds the ticket array runs unsplit (such shapes fill the chip without split-K anyway)
static GemmPlan clamp_plan(GemmPlan pl, int M, int N, int K) {
    if (pl.split > 1 && (long)cdiv(M, pl.bm) * cdiv(N, pl.bn) > GEMM_SPLITK_MAX_TILES) {
        pl.split = 1;
        pl.ksplit = K;
    }
    return pl;
}

static GemmPlan plan_for(int ta, int tb, int M, int N, int K, int dtype) {
    GemmForce f = resolve_force(ta, tb, M, N, K, dtype);
    if (f.bm == 256 && dtype != CAPDEC_BF16S) f.bm = f.bn = 128;  // the 256-row tiles exist for bf16 storage only
    return dtype == CAPDEC_BF16 ? plan_gemm_bf16(M, N, K, f) : dtype == CAPDEC_F32X3 ? plan_gemm_f32x3(M, N, K, f)
           : dtype == CAPDEC_BF16S ? plan_gemm_bf16s(M, N, K, f) : plan_gemm(M, N, K, f);
}

extern "C" int capdec_gemm_ce_tiles(int M, int N, int K, int dtype) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    return cdiv(N, plan_for(0, 1, M, N, K, dtype).bn);  // the logits product: A [M,K], B stored [N,K]
}

extern "C" int capdec_gemm_plan_set(int transA, int transB, int M, int N, int K, int dtype, int bm, int bn, int split,
                                    int bk) {
    const bool ok = bm == 0 || (((bm == 64 || bm == 128) && (bn == 64 || bn == 128) || (bm == 256 && (bn == 128 || bn == 256) && dtype == CAPDEC_BF16S)) &&
                                split >= 1 && split <= 64 && (bk == 0 || bk == 16 || bk == 32));
    CAPDEC_REQUIRE(ok && M > 0 && N > 0 && K > 0, "capdec_gemm_plan_set: unsupported plan %dx%d split %d bk %d", bm, bn, split, bk);
    std::lock_guard<std::mutex> lock(g_plan_mu);
    if (bm == 0) g_plans.erase(plan_key(transA, transB, M, N, K, dtype));
    else g_plans[plan_key(transA, transB, M, N, K, dtype)] = GemmForce{bm, bn, split, bk};
    return CAPDEC_OK;
}

extern "C" void capdec_gemm_plan_clear(void) {
    std::lock_guard<std::mutex> lock(g_plan_mu);
    g_plans.clear();
}

extern "C" size_t capdec_gemm_workspace_bytes(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    size_t need = 0;  // the largest need over both transposes of either operand and the compute types
    if (M <= 64 && K % 128 == 0 && K >= 128 && (long)cdiv(N, 32) * cdiv(M, 32) <= GEMM_SPLITK_MAX_TILES)
        need = gemm_skinny_workspace_bytes(M, N, K);  // few-row products (gemm_skinny.hip)
    for (int ta = 0; ta < 2; ++ta)
        for (int tb = 0; tb < 2; ++tb)
            for (int dtype : {CAPDEC_F32, CAPDEC_BF16, CAPDEC_F32X3, CAPDEC_BF16S}) {
                const size_t b = splitk_workspace_bytes(clamp_plan(plan_for(ta, tb, M, N, K, dtype), M, N, K), M, N);
                if (b > need) need = b;
            }
    return need;
}

// fill GemmParams from the public arguments (shared by capdec_gemm's checks below and capdec_gemm_grouped)
static void fill_params(GemmParams& p, int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                        const capdec_epilogue* ep) {
    memset(&p, 0, sizeof(p));
    p.A = (const float*)A;
    p.B = (const float*)B;
    p.C = (float*)C;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.drop = make_dropout(0.f, 0);
    if (ep) {
        p.bias = (const float*)ep->bias;
        p.addend = (const float*)ep->addend;
        p.pre = (float*)ep->pre;
        p.aux = (const float*)ep->aux;
        p.colsum = (float*)ep->colsum;
        p.ld_addend = ep->ld_addend;
        p.act = ep->act;
        p.drop = make_dropout(ep->drop_p, ep->drop_seed);
    }
}

template <int BM, int BN, int BK>
static void launch_grouped(const GemmGroupParams& g, bool AK, bool BKM, int blocks, hipStream_t st) {
    const dim3 grid(blocks), block(256);
    if (AK && BKM) hipLaunchKernelGGL((gemm_f32_grouped_kernel<BM, BN, BK, true, true>), grid, block, 0, st, g);
    else if (AK && !BKM) hipLaunchKernelGGL((gemm_f32_grouped_kernel<BM, BN, BK, true, false>), grid, block, 0, st, g);
    else if (!AK && !BKM) hipLaunchKernelGGL((gemm_f32_grouped_kernel<BM, BN, BK, false, false>), grid, block, 0, st, g);
    else hipLaunchKernelGGL((gemm_f32_grouped_kernel<BM, BN, BK, false, true>), grid, block, 0, st, g);
}

extern "C" int capdec_gemm_grouped(const capdec_gemm_desc* descs, int n, int bm, int bn, capdec_stream_t stream) {
    CAPDEC_REQUIRE(descs && n > 0 && n <= CAPDEC_GEMM_GROUP_MAX, "capdec_gemm_grouped: 1 .. %d products per launch, got %d",
                   CAPDEC_GEMM_GROUP_MAX, n);
    CAPDEC_REQUIRE((bm == 0 && bn == 0) || ((bm == 64 || bm == 128) && (bn == 64 || bn == 128)),
                   "capdec_gemm_grouped: tile %dx%d not in {64,128}^2 (0, 0 = default)", bm, bn);
    if (bm == 0) bm = bn = 64;
    GemmGroupParams g;
    memset(&g, 0, sizeof(g));
    g.n = n;
    int first = 0;
    const int ta = descs[0].transA != 0, tb = descs[0].transB != 0;
    for (int i = 0; i < n; ++i) {
        const capdec_gemm_desc& d = descs[i];
        CAPDEC_REQUIRE((d.transA != 0) == ta && (d.transB != 0) == tb, "capdec_gemm_grouped: product %d has another layout than product 0", i);
        CAPDEC_REQUIRE(d.M > 0 && d.N > 0 && d.K > 0 && d.A && d.B && d.C, "capdec_gemm_grouped: product %d: bad shape / null operand", i);
        CAPDEC_REQUIRE(d.lda >= (ta ? d.M : d.K) && d.ldb >= (tb ? d.K : d.N) && d.ldc >= d.N,
                       "capdec_gemm_grouped: product %d: leading dimension too small", i);
        CAPDEC_REQUIRE(aligned16(d.A) && aligned16(d.B) && d.lda % 4 == 0 && d.ldb % 4 == 0,
                       "capdec_gemm_grouped: product %d: operands must be 16-byte aligned with leading dimensions that are multiples of 4", i);
        const capdec_epilogue* ep = d.ep;
        CAPDEC_REQUIRE(!ep || (!ep->ce_partials && !ep->ce_lse && !ep->c16), "capdec_gemm_grouped: product %d: no cross-entropy fusion / bf16 result here", i);
        CAPDEC_REQUIRE(!ep || ep->act != CAPDEC_ACT_GELU_BWD || ep->aux, "capdec_gemm_grouped: product %d: ACT_GELU_BWD needs aux", i);
        CAPDEC_REQUIRE(!ep || !ep->colsum || ta, "capdec_gemm_grouped: product %d: colsum needs transA=1", i);
        CAPDEC_REQUIRE(!ep || !ep->addend || ep->ld_addend >= d.N, "capdec_gemm_grouped: product %d: ld_addend too small", i);
        CAPDEC_REQUIRE(!ep || (ep->drop_p >= 0.f && ep->drop_p < 1.f), "capdec_gemm_grouped: product %d: drop_p out of range", i);
        CAPDEC_REQUIRE(!(ep && ep->drop_p > 0.f) || (uint64_t)d.M * (uint64_t)d.N < (1ull << 32),
                       "capdec_gemm_grouped: product %d: M*N exceeds 2^32 with dropout in the epilogue", i);
        fill_params(g.p[i], d.M, d.N, d.K, d.A, d.lda, d.B, d.ldb, d.C, d.ldc, ep);
        g.p[i].ksplit = d.K;  // unsplit
        g.p[i].tiles_n = cdiv(d.N, bn);
        const long tiles = (long)cdiv(d.M, bm) * g.p[i].tiles_n;
        CAPDEC_REQUIRE(tiles + first < (1l << 30), "capdec_gemm_grouped: too many tiles");
        g.first[i] = first;
        g.tiles[i] = (int)tiles;
        first += ((int)tiles + 7) / 8 * 8;
    }
    hipStream_t st = (hipStream_t)stream;
    const bool AK = !ta, BKM = tb;
    if (bm == 128 && bn == 128) launch_grouped<128, 128, 16>(g, AK, BKM, first, st);
    else if (bm == 128 && bn == 64) launch_grouped<128, 64, 16>(g, AK, BKM, first, st);
    else if (bm == 64 && bn == 128) launch_grouped<64, 128, 16>(g, AK, BKM, first, st);
    else launch_grouped<64, 64, 16>(g, AK, BKM, first, st);
    return check_launch("capdec_gemm_grouped");
}

extern "C" int capdec_gemm(int transA, int transB, int M, int N, int K, const void* A, int lda, const void* B,
                           int ldb, void* C, int ldc, const capdec_epilogue* ep, int dtype, void* workspace,
                           size_t workspace_bytes, capdec_stream_t stream) {
    CAPDEC_REQUIRE(dtype == CAPDEC_F32 || dtype == CAPDEC_BF16 || dtype == CAPDEC_F32X3 || dtype == CAPDEC_BF16S,
                   "capdec_gemm: dtype %d not supported", dtype);
    CAPDEC_REQUIRE(M > 0 && N > 0 && K > 0, "capdec_gemm: bad shape M=%d N=%d K=%d", M, N, K);
    CAPDEC_REQUIRE(A && B && (C || (ep && ep->c16)), "capdec_gemm: null operand");
    CAPDEC_REQUIRE(lda >= (transA ? M : K) && ldb >= (transB ? K : N) && ldc >= N,
                   "capdec_gemm: leading dimension too small (lda=%d ldb=%d ldc=%d)", lda, ldb, ldc);
    // the epilogue's dropout mask is indexed by the 32-bit element number m*N + n: only products WITH dropout are bound
    // by it (the vocabulary projection has none and is the largest M*N of the path)
    CAPDEC_REQUIRE(!(ep && ep->drop_p > 0.f) || (uint64_t)M * (uint64_t)N < (1ull << 32),
                   "capdec_gemm: M*N exceeds 2^32 with dropout in the epilogue (32-bit mask index)");
    CAPDEC_REQUIRE((uint64_t)cdiv(M, 64) * (uint64_t)cdiv(N, 64) < (1ull << 31), "capdec_gemm: too many tiles");
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.A = (const float*)A;
    p.B = (const float*)B;
    p.C = (float*)C;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.drop = make_dropout(0.f, 0);
    if (ep) {
        p.bias = (const float*)ep->bias;
        p.addend = (const float*)ep->addend;
        p.pre = (float*)ep->pre;
        p.aux = (const float*)ep->aux;
        p.colsum = (float*)ep->colsum;
        p.ld_addend = ep->ld_addend;
        p.act = ep->act;
        p.drop = make_dropout(ep->drop_p, ep->drop_seed);
        p.c16 = (__bf16*)ep->c16;
        p.ldc16 = ep->ldc16;
        p.unpadded = ep->bf16_unpadded;
        CAPDEC_REQUIRE(!p.c16 || ep->ldc16 >= N, "capdec_gemm: c16 needs ldc16 >= N");
        // bf16 logits: the loss's target logit must then come from the fp32 accumulators (the partials' 4th component)
        CAPDEC_REQUIRE(!p.c16 || !ep->ce_partials || ep->ce_targets,
                       "capdec_gemm: c16 with ce_partials needs ce_targets (the target logits leave through the partials)");
        p.ce_part = (float4*)ep->ce_partials;
        p.ce_lse = (const float*)ep->ce_lse;
        p.ce_y = ep->ce_targets;
        if (p.ce_lse) {
            const int V = transA ? M : K;  // A is the [rows, V] logits block either way
            CAPDEC_REQUIRE(ep->ce_targets && dtype == CAPDEC_F32 && ep->ce_eps >= 0.f && ep->ce_eps < 1.f,
                           "capdec_gemm: ce_lse needs ce_targets, 0 <= ce_eps < 1 and dtype CAPDEC_F32 "
                           "(other compute types: capdec_ce_dlogits, then a plain product)");
            p.ce_gs = ep->ce_grad_scale;
            p.ce_u = ep->ce_eps / (float)V;
            p.ce_w = 1.f - ep->ce_eps;
        }
        CAPDEC_REQUIRE(!p.ce_part || (!transA && aligned16(ep->ce_partials) && ep->act == CAPDEC_ACT_NONE &&
                                      !ep->addend && ep->drop_p == 0.f),
                       "capdec_gemm: ce_partials is for the plain logits product (A not transposed, bias only)");
        CAPDEC_REQUIRE(ep->act != CAPDEC_ACT_GELU_BWD || ep->aux, "capdec_gemm: ACT_GELU_BWD needs aux");
        CAPDEC_REQUIRE(!ep->colsum || transA, "capdec_gemm: colsum needs transA=1");
        CAPDEC_REQUIRE(!ep->addend || ep->ld_addend >= N, "capdec_gemm: ld_addend too small");
        CAPDEC_REQUIRE(ep->drop_p >= 0.f && ep->drop_p < 1.f, "capdec_gemm: drop_p out of range");
    }
    // a few rows against a weight matrix (decode steps): the K-split-inside-the-workgroup kernel, unless a plan was
    // forced (tuning hook / measured plan) for this product
    if (gemm_skinny_ok(p, !transA, transB != 0, dtype) && !resolve_force(transA, transB, M, N, K, dtype).bm) {
        if (const int rc = launch_gemm_skinny(p, workspace, workspace_bytes, (hipStream_t)stream)) return rc;
        return check_launch("capdec_gemm");
    }
    GemmPlan pl = clamp_plan(plan_for(transA, transB, M, N, K, dtype), M, N, K);
    if (dtype == CAPDEC_BF16S && pl.bn == 256 && !gemm_bf16s_dma_ok(p, !transA, transB != 0)) {
        // the 256 x 256 block exists only in the LDS-DMA form; these operands need the register-staged 256 x 128 one
        CAPDEC_REQUIRE(!p.ce_part, "capdec_gemm: ce_partials was sized by capdec_gemm_ce_tiles for a 256-column plan, which "
                                   "needs bf16 operands whose pitch covers K rounded up to 64 (zero padded)");
        pl.bn = 128;
        pl = clamp_plan(pl, M, N, K);
    }
    if (p.ce_part) {  // a row's partials come from whole column tiles: no K split (the vocabulary product has none)
        pl.split = 1;
        pl.ksplit = K;
    }
    if (pl.split > 1) {
        const size_t need = splitk_workspace_bytes(pl, M, N);
        if (!workspace || workspace_bytes < need || !aligned16(workspace)) {
            set_error("capdec_gemm: split-K needs %zu workspace bytes (16-byte aligned), got %zu", need, workspace_bytes);
            return CAPDEC_ERR_WORKSPACE;
        }
        const size_t tiles_m = cdiv(M, pl.bm), tiles = tiles_m * cdiv(N, pl.bn);
        p.counters = (int*)workspace;
        p.ws = (float*)((char*)workspace + GEMM_COUNTER_BYTES);
        p.ws_colsum = p.ws + tiles * pl.split * pl.bm * pl.bn;
        p.m_pad = (int)tiles_m * pl.bm;
    }
    p.ksplit = pl.ksplit;
    p.tiles_n = cdiv(N, pl.bn);
    const bool no_vec_epi = getenv("CAPDEC_NO_VEC_EPI") != nullptr;  // A/B hook (read per call): the element form everywhere
    p.vec_epi = !no_vec_epi && epilogue_vec_ok(p);
    const bool vec = aligned16(A) && aligned16(B) && (lda % 4 == 0) && (ldb % 4 == 0);
    const bool AK = !transA, BKM = transB != 0;
    if (dtype == CAPDEC_BF16S) {
        // bf16 operands move in 16-byte pieces of 8 elements: pitches are multiples of 8 that cover the contiguous
        // extent ROUNDED UP to 8 (a piece that straddles the edge is read whole and masked in registers)
        const int ea = (transA ? M : K), eb = (transB ? K : N);
        CAPDEC_REQUIRE(aligned16(A) && aligned16(B) && lda % 8 == 0 && ldb % 8 == 0 && lda >= (ea + 7) / 8 * 8 && ldb >= (eb + 7) / 8 * 8,
                       "capdec_gemm: CAPDEC_BF16S operands are bf16 matrices, 16-byte aligned, with leading dimensions "
                       "that are multiples of 8 and cover the row rounded up to 8 (lda=%d ldb=%d)", lda, ldb);
        CAPDEC_REQUIRE(!p.ce_lse, "capdec_gemm: CAPDEC_BF16S takes dlogits from capdec_ce_dlogits_bf16, not ce_lse");
    }
    CAPDEC_REQUIRE(!p.ce_lse || (vec && !BKM), "capdec_gemm: ce_lse needs transB = 0 and 16-byte aligned operands with "
                                               "leading dimensions that are multiples of 4");
    dim3 grid(cdiv(M, pl.bm) * p.tiles_n, 1, pl.split);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == CAPDEC_BF16S) {
        launch_gemm_bf16s(p, pl, AK, BKM, st);
    } else if (dtype == CAPDEC_BF16) {
        launch_gemm_bf16(p, pl, AK, BKM, vec, st);
    } else if (dtype == CAPDEC_F32X3) {
        const int lrc = launch_gemm_f32x3(p, pl, AK, BKM, vec, st);
        if (lrc) return lrc;
    } else if (pl.bk == 32) {
        if (pl.bm == 128 && pl.bn == 128) launch_cfg<128, 128, 32>(p, AK, BKM, vec, grid, st);
        else if (pl.bm == 128 && pl.bn == 64) launch_cfg<128, 64, 32>(p, AK, BKM, vec, grid, st);
        else if (pl.bm == 64 && pl.bn == 128) launch_cfg<64, 128, 32>(p, AK, BKM, vec, grid, st);
        else launch_cfg<64, 64, 32>(p, AK, BKM, vec, grid, st);
    } else {
        if (pl.bm == 128 && pl.bn == 128) launch_cfg<128, 128, 16>(p, AK, BKM, vec, grid, st);
        else if (pl.bm == 128 && pl.bn == 64) launch_cfg<128, 64, 16>(p, AK, BKM, vec, grid, st);
        else if (pl.bm == 64 && pl.bn == 128) launch_cfg<64, 128, 16>(p, AK, BKM, vec, grid, st);
        else launch_cfg<64, 64, 16>(p, AK, BKM, vec, grid, st);
    }
    return check_launch("capdec_gemm");
}
