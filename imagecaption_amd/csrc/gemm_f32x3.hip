// Split-fp32 GEMM (dtype = CAPDEC_F32X3): fp32 accuracy on the bf16 matrix cores.
//
// Every fp32 operand element a is split exactly into three bf16 terms, a = a0 + a1 + a2
// (a0 = rne_bf16(a), a1 = rne_bf16(a - a0), a2 = a - a0 - a1: 3 x 8 significand bits + signs cover
// fp32's 24 bits, so the split loses nothing), and the product is assembled from the six partial
// products of weight >= 2^-16,
//     a*b ~= a0 b0 + (a0 b1 + a1 b0) + (a0 b2 + a1 b1 + a2 b0),
// each a v_mfma_f32_32x32x16_bf16 with fp32 accumulation (bf16 x bf16 is exact in fp32).  The dropped
// terms a1 b2 + a2 b1 + a2 b2 are <= 2^-23 |ab| worst case, ~2^-26 typical: below the rounding a fp32
// FMA chain commits per step, so the result is as accurate as the exact-fp32 MFMA path of
// gemm_f32.hip (tests/test_ops_gpu.py compares both against float64) at 16/6 = 2.7x its matrix rate.
// Non-finite operands come out as NaN (inf - inf in the residual), finite data is unaffected.
//
// Block tile BM x BN x 32, 256 threads (2 x 2 waves over the tile): every wave loads, splits, stores
// AND multiplies -- the structure of gemm_bf16.hip with three planes and six products.  One LDS buffer
// (dynamic, 30-60 KiB) plus a register stage; the split / store phase of one workgroup is covered by
// the MFMAs of the other workgroups resident on the CU (2 at 128x128, 3 at 128x64, 4 at 64x64).
// (A first version with four dedicated producer waves feeding four consumer waves was 5-25 % slower
// at every shape: one MFMA-issuing wave per SIMD cannot hide its own LDS read latency.)
// Where it stands (PMC, vocab projection, 128x128): matrix pipe busy 46 %, LDS 34 %, 5.7 vector
// instructions per MFMA (~50 % of vector issue): nothing saturated, the two waves per SIMD expose their
// LDS-read waits and store phases.  150-170 TFLOP/s fp32-equivalent on large products.  A variant that
// split each operand once into bf16 planes in HBM (no split instructions in the GEMM) was NOT faster.
// LDS images, three bf16 planes per operand:
//   k-contiguous operand  : [row][40]   (80-B rows)  -> fragment = one ds_read_b128, conflict-free
//   row-contiguous operand: [k][rows+32] (rows*2+64 B per k) -> written as loaded (no register
//                           transpose), fragment = two ds_read_b64_tr_b16 (hardware transpose),
//                           4 k-rows land on 4 distinct 64-B bank quarters
#include "common.h"
#include "gemm_common.h"

namespace capdec {


typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int XBK = 32;   // k-tile
constexpr int XLDK = 40;  // row stride of the [row][k] image, bf16 elements
constexpr int XPAD = 32;  // row padding of the [k][row] image, bf16 elements

struct Split4 {
    bf16x4 t0, t1, t2;
};

__device__ __forceinline__ Split4 split4(float4 v) {
    const f32x4 x = {v.x, v.y, v.z, v.w};
    Split4 s;
    s.t0 = __builtin_convertvector(x, bf16x4);
    const f32x4 r1 = x - __builtin_convertvector(s.t0, f32x4);
    s.t1 = __builtin_convertvector(r1, bf16x4);
    const f32x4 r2 = r1 - __builtin_convertvector(s.t1, f32x4);
    s.t2 = __builtin_convertvector(r2, bf16x4);
    return s;
}

template <bool VEC>
__device__ __forceinline__ float4 xload4(const float* ptr, bool ok, int rem) {
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!ok || rem <= 0) return r;
    if (VEC && rem >= 4) return *reinterpret_cast<const float4*>(ptr);
    r.x = ptr[0];
    if (rem > 1) r.y = ptr[1];
    if (rem > 2) r.z = ptr[2];
    if (rem > 3) r.w = ptr[3];
    return r;
}

__device__ __forceinline__ bf16x4 lds_read_tr(const __bf16* ptr) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)ptr);
    return __builtin_bit_cast(bf16x4, v);
}

// one operand: global -> registers -> three-plane LDS image; fragment reads
template <int ROWS, bool KC, bool VEC>
struct Operand {
    static constexpr int NT = 256;                                    // threads per workgroup
    static constexpr int REGS = ROWS * XBK / 4 / NT;                  // float4 per thread per k-tile
    static constexpr int LD = KC ? XLDK : ROWS + XPAD;                // image row stride (bf16 elements)
    static constexpr int PLANE = KC ? ROWS * XLDK : XBK * (ROWS + XPAD);
    static constexpr int RQ = ROWS / 4;

    // src(row, k) = base[row * ld + k] if KC else base[k * ld + row]
    template <bool INTERIOR>
    __device__ static __forceinline__ void gload(float4* r, const float* base, int ld, int row0, int nrows, int k0, int kend,
                                                 int tid) {
#pragma unroll
        for (int i = 0; i < REGS; ++i) {
            const int f = tid + i * NT;
            if (KC) {
                const int row = f >> 3, kq = f & 7;
                const int gr = row0 + row, gk = k0 + kq * 4;
                const float* ptr = base + (size_t)gr * ld + gk;
                r[i] = INTERIOR ? *reinterpret_cast<const float4*>(ptr) : xload4<VEC>(ptr, gr < nrows, kend - gk);
            } else {
                const int kr = f / RQ, rq = f % RQ;
                const int gk = k0 + kr, gr = row0 + rq * 4;
                const float* ptr = base + (size_t)gk * ld + gr;
                r[i] = INTERIOR ? *reinterpret_cast<const float4*>(ptr) : xload4<VEC>(ptr, gk < kend, nrows - gr);
            }
        }
    }

    __device__ static __forceinline__ void lstore(const float4* r, __bf16* img, int tid) {
#pragma unroll
        for (int i = 0; i < REGS; ++i) {
            const int f = tid + i * NT;
            const int off = KC ? (f >> 3) * XLDK + (f & 7) * 4 : (f / RQ) * LD + (f % RQ) * 4;
            const Split4 s = split4(r[i]);
            *reinterpret_cast<bf16x4*>(img + off) = s.t0;
            *reinterpret_cast<bf16x4*>(img + PLANE + off) = s.t1;
            *reinterpret_cast<bf16x4*>(img + 2 * PLANE + off) = s.t2;
        }
    }

    // fragment of 32 rows starting at `row` for MFMA k-step ks (16 k), plane `pl`
    __device__ static __forceinline__ bf16x8 frag(const __bf16* img, int pl, int row, int ks, int lane) {
        const __bf16* base = img + pl * PLANE;
        if (KC) {
            return *reinterpret_cast<const bf16x8*>(base + (row + (lane & 31)) * XLDK + ks * 16 + (lane >> 5) * 8);
        } else {
            const int q = (lane & 15) >> 2, pp = lane & 3;
            const __bf16* ptr = base + (ks * 16 + (lane >> 5) * 8 + q) * LD + row + ((lane >> 4) & 1) * 16 + pp * 4;
            const bf16x4 lo = lds_read_tr(ptr);
            const bf16x4 hi = lds_read_tr(ptr + 4 * LD);
            return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
    }
};

template <int BM, int BN, bool AK, bool BKM, bool VEC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BM * BN >= 128 * 128 ? 2 : (BM * BN >= 128 * 64 ? 3 : 4))))
void gemm_f32x3_kernel(GemmParams p) {
    typedef Operand<BM, AK, VEC> OA;
    typedef Operand<BN, BKM, VEC> OB;
    constexpr int TM = BM / 64, TN = BN / 64;
    extern __shared__ __attribute__((aligned(16))) __bf16 x3_smem[];
    __bf16* As = x3_smem;
    __bf16* Bs = As + 3 * OA::PLANE;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
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
    const int nk = (kend - kbeg + XBK - 1) / XBK;
    const bool interior = VEC && (m0 + BM <= p.M) && (n0 + BN <= p.N) && ((kend - kbeg) % XBK == 0);
    const bool do_colsum = (!AK) && p.colsum != nullptr && tile_n == 0;

    const int l31 = lane & 31, half = lane >> 5;
    const int wrow = (wave >> 1) * (BM / 2), wcol = (wave & 1) * (BN / 2);
    f32x16 acc[TM][TN], cor[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f, cor[i][j][r] = 0.f;

    float4 ra[OA::REGS], rb[OB::REGS];
    float csum = 0.f;
    auto fetch = [&](int kt) {
        const int k0 = kbeg + kt * XBK;
        if (interior) {
            OA::template gload<true>(ra, p.A, p.lda, m0, p.M, k0, kend, tid);
            OB::template gload<true>(rb, p.B, p.ldb, n0, p.N, k0, kend, tid);
        } else {
            OA::template gload<false>(ra, p.A, p.lda, m0, p.M, k0, kend, tid);
            OB::template gload<false>(rb, p.B, p.ldb, n0, p.N, k0, kend, tid);
        }
    };
    if (nk > 0) {
        fetch(0);
        OA::lstore(ra, As, tid);
        OB::lstore(rb, Bs, tid);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) fetch(kt + 1);
#pragma unroll
        for (int ks = 0; ks < XBK / 16; ++ks) {
            bf16x8 af[3][TM], bf[3][TN];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[pl][i] = OA::frag(As, pl, wrow + i * 32, ks, lane);
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[pl][j] = OB::frag(Bs, pl, wcol + j * 32, ks, lane);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    f32x16 c = cor[i][j];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2][i], bf[0][j], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[2][j], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bf[1][j], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bf[0][j], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[1][j], c, 0, 0, 0);
                    cor[i][j] = c;
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[0][j], acc[i][j], 0, 0, 0);
                }
        }
        if (do_colsum && tid < BM) {  // A image is [k][m]; the three planes sum to the fp32 value exactly
#pragma unroll 8
            for (int k = 0; k < XBK; ++k)
                csum += ((float)As[k * OA::LD + tid] + (float)As[OA::PLANE + k * OA::LD + tid]) +
                        (float)As[2 * OA::PLANE + k * OA::LD + tid];
        }
        __syncthreads();
        if (kt + 1 < nk) {
            OA::lstore(ra, As, tid);
            OB::lstore(rb, Bs, tid);
        }
        __syncthreads();
    }

#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] += cor[i][j];  // main term + the five corrections
    // split-K: only the last-arriving slice of this tile goes on, with the slice-ordered sum in acc / csum
    if (gridDim.z > 1 && !splitk_combine<TM, TN>(p, acc, csum, do_colsum, tile_m * p.tiles_n + tile_n, m0, BM, tid,
                                                  reinterpret_cast<int*>(x3_smem)))
        return;
    if (do_colsum && tid < BM && m0 + tid < p.M) p.colsum[m0 + tid] = csum;
    if (p.ce_part) ce_tile_partials<TM, TN, BM, BN>(p, acc, reinterpret_cast<float*>(x3_smem), m0, n0, tile_n, tid);
    if (p.vec_epi) {  // the vector form of the epilogue (gemm_common.h): 32 rows of the wave block at a time through LDS
        __syncthreads();  // every wave is done with whatever used the LDS last
        epilogue_store_vec<TM, TN, 1>(p, acc, m0 + wrow, n0 + wcol, lane, reinterpret_cast<float*>(x3_smem) + wave * (32 * 32 * TN));
        return;
    }
    epilogue_store<TM, TN>(p, acc, m0 + wrow, n0 + wcol, l31, half);
}

template <int BM, int BN, bool AK, bool BKM, bool VEC>
static int launch_one(const GemmParams& p, dim3 grid, hipStream_t st) {
    constexpr size_t lds = 3 * (size_t)(Operand<BM, AK, VEC>::PLANE + Operand<BN, BKM, VEC>::PLANE) * sizeof(__bf16);
    static bool configured[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return CAPDEC_ERR_LAUNCH;
    if (!configured[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f32x3_kernel<BM, BN, AK, BKM, VEC>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            set_error("capdec_gemm(f32x3): cannot reserve %zu bytes of LDS", lds);
            return CAPDEC_ERR_LAUNCH;
        }
        configured[dev] = true;
    }
    hipLaunchKernelGGL((gemm_f32x3_kernel<BM, BN, AK, BKM, VEC>), grid, dim3(256), lds, st, p);
    return CAPDEC_OK;
}

template <int BM, int BN>
static int launch_x3(const GemmParams& p, bool AK, bool BKM, bool vec, dim3 grid, hipStream_t st) {
#define CAPDEC_LAUNCH(ak, bk, v) return launch_one<BM, BN, ak, bk, v>(p, grid, st)
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

GemmPlan plan_gemm_f32x3(int M, int N, int K, const GemmForce& force) {
    if (force.bm) {
        const int split = force.split > 0 ? force.split : 1;
        const int ks = cdiv(cdiv(K, split), XBK) * XBK;
        int bm = force.bm, bn = force.bn;
        if (bm == 64 && bn == 128) bm = 128, bn = 64;  // not instantiated; nearest
        return GemmPlan{bm, bn, XBK, cdiv(K, ks), ks};
    }
    // Cost model calibrated on tools/gemm_sweep.py (SWEEP_DTYPE=f32x3, profiles/r01_gemm_sweep_f32x3.txt):
    // a CU holds `res` workgroups of a tile shape; a round of them costs res x (MFMA time of one
    // workgroup) / eff, where eff (matrix-pipe utilisation) grows with the tile because the split work
    // per MFMA shrinks; split-K adds its partial round trip.
    static const struct { int bm, bn, res; double eff; } cand[3] = {{128, 128, 2, 0.47}, {128, 64, 3, 0.43}, {64, 64, 4, 0.37}};
    GemmPlan best{64, 64, XBK, 1, cdiv(K, XBK) * XBK};
    double best_cost = 1e300;
    for (const auto& c : cand) {
        const long tiles = (long)cdiv(M, c.bm) * cdiv(N, c.bn);
        const double units = (c.bm / 64) * (c.bn / 64);
        for (int split = 1; split <= 32; split *= 2) {
            const int ks = cdiv(cdiv(K, split), XBK) * XBK;
            if (split > 1 && ks < 256) break;
            const int s_eff = cdiv(K, ks);
            double rounds = (double)(tiles * s_eff) / (256.0 * c.res);
            if (rounds < 4.0) {  // few rounds: a started round costs at least half a round
                const double whole = (double)(long)rounds, frac = rounds - whole;
                rounds = whole + (frac > 0.0 ? (frac > 0.5 ? frac : 0.5) : 0.0);
                if (rounds < 1.0) rounds = 1.0;
            }
            double cost = rounds * c.res * units * 0.0928 * (ks / 16.0 + 4.0) / c.eff;  // microseconds
            if (s_eff > 1) cost += (double)M * N * s_eff * 8.0 / 3.0e6 + 6.0;
            if (cost < best_cost * 0.999) {
                best_cost = cost;
                best = GemmPlan{c.bm, c.bn, XBK, s_eff, ks};
            }
        }
    }
    return best;
}

int launch_gemm_f32x3(const GemmParams& p, const GemmPlan& pl, bool AK, bool BKM, bool vec, hipStream_t st) {
    dim3 grid(cdiv(p.M, pl.bm) * p.tiles_n, 1, pl.split);
    if (pl.bm == 128 && pl.bn == 128) return launch_x3<128, 128>(p, AK, BKM, vec, grid, st);
    if (pl.bm == 128 && pl.bn == 64) return launch_x3<128, 64>(p, AK, BKM, vec, grid, st);
    return launch_x3<64, 64>(p, AK, BKM, vec, grid, st);
}

}  // namespace capdec
