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
#include "common.h"
#include "gemm_common.h"

namespace capdec {

GemmForce g_gemm_force = {0, 0, 0, 0};

typedef float f32x16 __attribute__((ext_vector_type(16)));

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
template <int BM, int BN, int GBK, bool AK, bool BKM, bool VEC>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmParams p) {
    constexpr int NT = 256;
    constexpr int TM = BM / 64, TN = BN / 64;  // 32x32 accumulators per wave in each direction
    constexpr int A_LD = AK ? (BM + 2) * 4 : (BM + 4);  // floats per k-quad (AK) or per k (row-contig)
    constexpr int B_LD = BKM ? (BN + 2) * 4 : (BN + 4);
    constexpr int A_SZ = AK ? (GBK / 4) * A_LD : GBK * A_LD;
    constexpr int B_SZ = BKM ? (GBK / 4) * B_LD : GBK * B_LD;
    constexpr int A_V4 = BM * GBK / 4 / NT;
    constexpr int B_V4 = BN * GBK / 4 / NT;
    static_assert(A_V4 >= 1 && B_V4 >= 1, "tile too small for 256 threads");
    __shared__ __attribute__((aligned(16))) float smem[2 * (A_SZ + B_SZ)];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int wrow = (wave >> 1) * (BM / 2), wcol = (wave & 1) * (BN / 2);

    // XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch), so give each XCD
    // a contiguous run of tiles (neighbouring tiles share an A row panel in that XCD's L2).
    const int nwg = gridDim.x;
    int wg = blockIdx.x;
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
    };
    auto lstore = [&](int buf) {
        float* As = smem + buf * (A_SZ + B_SZ);
        float* Bs = As + A_SZ;
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

    if (nk > 0) {
        gload(kbeg);
        lstore(0);
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
        if (kt + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }

    const bool split = gridDim.z > 1;
    if (do_colsum && tid < BM && m0 + tid < p.M) {
        if (split) p.ws_colsum[(size_t)blockIdx.z * p.M + m0 + tid] = csum;
        else p.colsum[m0 + tid] = csum;
    }
    float* wsz = split ? p.ws + (size_t)blockIdx.z * p.M * p.N : nullptr;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wcol + j * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wrow + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (m < p.M && n < p.N) {
                    if (split) wsz[(size_t)m * p.N + n] = acc[i][j][r];
                    else p.C[(size_t)m * p.ldc + n] = epilogue_apply(p, acc[i][j][r], m, n);
                }
            }
        }
}

__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(GemmParams p, int S, int vec) {
    const size_t total = (size_t)p.M * p.N;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    if (vec) {  // N % 4 == 0: one float4 of every slab per thread, slabs summed in slab order
        const size_t total4 = total >> 2;
        for (size_t e4 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e4 < total4; e4 += stride) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
            for (int z = 0; z < S; ++z) {
                const float4 t = reinterpret_cast<const float4*>(p.ws + (size_t)z * total)[e4];
                v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
            }
            const size_t e = e4 << 2;
            const int m = (int)(e / p.N), n = (int)(e % p.N);
            float* c = p.C + (size_t)m * p.ldc + n;
            c[0] = epilogue_apply(p, v.x, m, n);
            c[1] = epilogue_apply(p, v.y, m, n + 1);
            c[2] = epilogue_apply(p, v.z, m, n + 2);
            c[3] = epilogue_apply(p, v.w, m, n + 3);
        }
    } else {
        for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
            float v = 0.f;
            for (int z = 0; z < S; ++z) v += p.ws[(size_t)z * total + e];
            const int m = (int)(e / p.N), n = (int)(e % p.N);
            p.C[(size_t)m * p.ldc + n] = epilogue_apply(p, v, m, n);
        }
    }
    if (p.colsum && blockIdx.x == 0) {
        for (int m = threadIdx.x; m < p.M; m += blockDim.x) {
            float v = 0.f;
            for (int z = 0; z < S; ++z) v += p.ws_colsum[(size_t)z * p.M + m];
            p.colsum[m] = v;
        }
    }
}

// ---- host side: tile / split-K choice ---------------------------------------------------------
// MFMA-bound cost model, calibrated on tools/gemm_sweep.py (profiles/r01_gemm_sweep.txt).  A CU
// with r co-resident workgroups runs r waves per SIMD; one wave per SIMD cannot cover its own
// barrier / load stalls, so the matrix pipe's utilisation u(r) rises with r.  Cost = rounds of
// workgroups per CU x work per workgroup / u, plus the split-K partial round trip.
static GemmPlan plan_gemm(int M, int N, int K) {
    static const int cand[4][2] = {{128, 128}, {128, 64}, {64, 128}, {64, 64}};
    static const double util[5] = {0.0, 0.55, 0.80, 0.97, 1.0};
    const int bk = g_gemm_force.bk ? g_gemm_force.bk : 16;
    if (g_gemm_force.bm) {
        int split = g_gemm_force.split > 0 ? g_gemm_force.split : 1;
        int ks = cdiv(cdiv(K, split), bk) * bk;
        return GemmPlan{g_gemm_force.bm, g_gemm_force.bn, bk, cdiv(K, ks), ks};
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
    const bool ok = ((bm == 0) || ((bm == 64 || bm == 128) && (bn == 64 || bn == 128) && split >= 1 && split <= 64)) &&
                    (bk == 0 || bk == 16 || bk == 32);
    CAPDEC_REQUIRE(ok, "capdec_gemm_tune: unsupported tile %dx%d split %d bk %d", bm, bn, split, bk);
    g_gemm_force = GemmForce{bm, bn, split, bk};
    return CAPDEC_OK;
}

extern "C" size_t capdec_gemm_workspace_bytes(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const GemmPlan a = plan_gemm(M, N, K), b = plan_gemm_bf16(M, N, K), c = plan_gemm_f32x3(M, N, K);
    int split = a.split > b.split ? a.split : b.split;
    if (c.split > split) split = c.split;
    if (split <= 1) return 0;
    return ((size_t)split * M * N + (size_t)split * M) * sizeof(float);
}

extern "C" int capdec_gemm(int transA, int transB, int M, int N, int K, const void* A, int lda, const void* B,
                           int ldb, void* C, int ldc, const capdec_epilogue* ep, int dtype, void* workspace,
                           size_t workspace_bytes, capdec_stream_t stream) {
    CAPDEC_REQUIRE(dtype == CAPDEC_F32 || dtype == CAPDEC_BF16 || dtype == CAPDEC_F32X3,
                   "capdec_gemm: dtype %d not supported", dtype);
    CAPDEC_REQUIRE(M > 0 && N > 0 && K > 0, "capdec_gemm: bad shape M=%d N=%d K=%d", M, N, K);
    CAPDEC_REQUIRE(A && B && C, "capdec_gemm: null operand");
    CAPDEC_REQUIRE(lda >= (transA ? M : K) && ldb >= (transB ? K : N) && ldc >= N,
                   "capdec_gemm: leading dimension too small (lda=%d ldb=%d ldc=%d)", lda, ldb, ldc);
    CAPDEC_REQUIRE((uint64_t)M * (uint64_t)N < (1ull << 32), "capdec_gemm: M*N exceeds 2^32 (dropout index)");
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
        CAPDEC_REQUIRE(ep->act != CAPDEC_ACT_GELU_BWD || ep->aux, "capdec_gemm: ACT_GELU_BWD needs aux");
        CAPDEC_REQUIRE(!ep->colsum || transA, "capdec_gemm: colsum needs transA=1");
        CAPDEC_REQUIRE(!ep->addend || ep->ld_addend >= N, "capdec_gemm: ld_addend too small");
        CAPDEC_REQUIRE(ep->drop_p >= 0.f && ep->drop_p < 1.f, "capdec_gemm: drop_p out of range");
    }
    GemmPlan pl = dtype == CAPDEC_BF16 ? plan_gemm_bf16(M, N, K)
                  : dtype == CAPDEC_F32X3 ? plan_gemm_f32x3(M, N, K) : plan_gemm(M, N, K);
    if (pl.split > 1) {
        const size_t need = ((size_t)pl.split * M * N + (size_t)pl.split * M) * sizeof(float);
        if (!workspace || workspace_bytes < need) {
            set_error("capdec_gemm: split-K needs %zu workspace bytes, got %zu", need, workspace_bytes);
            return CAPDEC_ERR_WORKSPACE;
        }
        p.ws = (float*)workspace;
        p.ws_colsum = p.ws + (size_t)pl.split * M * N;
    }
    p.ksplit = pl.ksplit;
    p.tiles_n = cdiv(N, pl.bn);
    const bool vec = aligned16(A) && aligned16(B) && (lda % 4 == 0) && (ldb % 4 == 0);
    const bool AK = !transA, BKM = transB != 0;
    dim3 grid(cdiv(M, pl.bm) * p.tiles_n, 1, pl.split);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == CAPDEC_BF16) {
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
    int rc = check_launch("capdec_gemm");
    if (rc) return rc;
    if (pl.split > 1) {
        const size_t total = (size_t)M * N;
        const int vec4 = (N % 4 == 0) && aligned16(workspace) ? 1 : 0;
        int blocks = (int)(((vec4 ? total / 4 : total) + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, p, pl.split, vec4);
        rc = check_launch("capdec_gemm(split-K reduce)");
    }
    return rc;
}
