// LayerNorm forward / backward (optionally fused with erf-GELU and dropout), one wave per row.
// HBM-bound: every row is read once into registers (float4 per lane), reduced with wave
// shuffles, and written once.  Backward accumulates dgamma/dbeta partials in registers across a
// grid-stride loop over rows and reduces them deterministically (fixed block count, fixed order).
// See include/capdec.h (capdec_layernorm_fwd / _bwd).
#include <mutex>

#include "common.h"

namespace capdec {

constexpr int LN_MAX_BLOCKS = 512;

// The forward LayerNorm arithmetic, spelled with explicit rounding intrinsics: ln_fwd_kernel and ln2_fwd_kernel (two
// norms back to back in one launch) must give the same bits, and left to itself the compiler contracts the same
// source expression differently in different surroundings (measured: 1-ulp row differences between the two kernels).
__device__ __forceinline__ float ln_sum4(const float4& v) { return __fadd_rn(__fadd_rn(v.x, v.y), __fadd_rn(v.z, v.w)); }
__device__ __forceinline__ float ln_sq4(const float4& v, float mean) {
    const float a = __fsub_rn(v.x, mean), b = __fsub_rn(v.y, mean), c = __fsub_rn(v.z, mean), e = __fsub_rn(v.w, mean);
    return __fadd_rn(__fmaf_rn(a, a, __fmul_rn(b, b)), __fmaf_rn(c, c, __fmul_rn(e, e)));
}
__device__ __forceinline__ float ln_out(float v, float mean, float rstd, float g, float b) {
    return __fmaf_rn(__fmul_rn(__fsub_rn(v, mean), rstd), g, b);
}
__device__ __forceinline__ float ln_rstd(float sumsq, int d, float eps) { return rsqrtf(__fadd_rn(__fdiv_rn(sumsq, (float)d), eps)); }

// the bf16 copies some callers want next to (or instead of) an fp32 result: four values -> 8 bytes, and the zero padding
// columns d .. ld16 - 1 of a row (capdec_cast_bf16's layout; d % 4 == 0 and ld16 % 8 == 0, so whole 8-byte groups)
typedef __bf16 ln_bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_bf16x4(__bf16* dst, const float* o) {
    ln_bf16x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (__bf16)o[j];
    *reinterpret_cast<ln_bf16x4*>(dst) = v;
}
__device__ __forceinline__ void zero_pad_bf16(__bf16* row, int d, int ld16, int lane) {
    for (int c = d + lane * 4; c < ld16; c += 256) *reinterpret_cast<uint2*>(row + c) = make_uint2(0u, 0u);
}

template <int NV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float* __restrict__ y,
                                                      float* __restrict__ mean_out, float* __restrict__ rstd_out, int M,
                                                      int d, float eps, int act, Dropout drop, __bf16* __restrict__ y16,
                                                      int ld16) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nq = d >> 2;
    const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * d);
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int q = lane + 64 * i;
        v[i] = q < nq ? xr[q] : make_float4(0.f, 0.f, 0.f, 0.f);
        s = __fadd_rn(s, ln_sum4(v[i]));
    }
    const float mean = __fdiv_rn(wave_sum(s), (float)d);
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int q = lane + 64 * i;
        if (q < nq) ss = __fadd_rn(ss, ln_sq4(v[i], mean));
    }
    const float rstd = ln_rstd(wave_sum(ss), d, eps);
    if (lane == 0) {
        if (mean_out) mean_out[row] = mean;
        if (rstd_out) rstd_out[row] = rstd;
    }
    float4* yr = reinterpret_cast<float4*>(y + (size_t)row * d);
    const float4* g4 = reinterpret_cast<const float4*>(gamma);
    const float4* b4 = reinterpret_cast<const float4*>(beta);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int q = lane + 64 * i;
        if (q < nq) {
            const float4 g = g4[q], b = b4[q];
            float o[4] = {ln_out(v[i].x, mean, rstd, g.x, b.x), ln_out(v[i].y, mean, rstd, g.y, b.y),
                          ln_out(v[i].z, mean, rstd, g.z, b.z), ln_out(v[i].w, mean, rstd, g.w, b.w)};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (act == CAPDEC_ACT_GELU) o[j] = gelu_erf(o[j]);
                if (drop.on()) o[j] = drop.apply(o[j], (uint32_t)row * (uint32_t)d + (uint32_t)(q * 4 + j));
            }
            yr[q] = make_float4(o[0], o[1], o[2], o[3]);
            if (y16) store_bf16x4(y16 + (size_t)row * ld16 + q * 4, o);
        }
    }
    if (y16) zero_pad_bf16(y16 + (size_t)row * ld16, d, ld16, lane);
}

// y = LN2(add[row / W] + LN1(x)): the two LayerNorms around the cross-attention branch of a decode step when the
// memory is one position (the branch is then a per-image constant, decode.py).  One wave per row, the row stays in
// registers between the two norms; each norm is ln_fwd_kernel's arithmetic, expression for expression, so the
// result equals LN1 -> add -> LN2 issued as three launches bit for bit (tests/test_decode_gpu.py).
template <int NV>
__global__ __launch_bounds__(256) void ln2_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma1,
                                                       const float* __restrict__ beta1, const float* __restrict__ add, int W,
                                                       const float* __restrict__ gamma2, const float* __restrict__ beta2,
                                                       float* __restrict__ y, int M, int d, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nq = d >> 2;
    const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * d);
    const float4* ar = reinterpret_cast<const float4*>(add + (size_t)(row / W) * d);
    float4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int q = lane + 64 * i;
        v[i] = q < nq ? xr[q] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) s = __fadd_rn(s, ln_sum4(v[i]));
        const float mean = __fdiv_rn(wave_sum(s), (float)d);
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int q = lane + 64 * i;
            if (q < nq) ss = __fadd_rn(ss, ln_sq4(v[i], mean));
        }
        const float rstd = ln_rstd(wave_sum(ss), d, eps);
        const float4* g4 = reinterpret_cast<const float4*>(pass ? gamma2 : gamma1);
        const float4* b4 = reinterpret_cast<const float4*>(pass ? beta2 : beta1);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int q = lane + 64 * i;
            if (q < nq) {
                const float4 g = g4[q], b = b4[q];
                float4 o = make_float4(ln_out(v[i].x, mean, rstd, g.x, b.x), ln_out(v[i].y, mean, rstd, g.y, b.y),
                                       ln_out(v[i].z, mean, rstd, g.z, b.z), ln_out(v[i].w, mean, rstd, g.w, b.w));
                if (pass == 0) {
                    const float4 u = ar[q];
                    o = make_float4(__fadd_rn(u.x, o.x), __fadd_rn(u.y, o.y), __fadd_rn(u.z, o.z), __fadd_rn(u.w, o.w));
                    v[i] = o;
                } else {
                    reinterpret_cast<float4*>(y + (size_t)row * d)[q] = o;
                }
            }
        }
    }
}

template <int NV>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                      float* __restrict__ dx, float* __restrict__ dx_drop,
                                                      float* __restrict__ part, int M, int d, int act, Dropout drop,
                                                      Dropout drop2, __bf16* __restrict__ dxd16, int ld16) {
    __shared__ float red[2][NV * 256];  // the four waves add their column partials into it one after the other
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nq = d >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(gamma);
    const float4* b4 = reinterpret_cast<const float4*>(beta);
    float4 gm[NV], bt[NV];
    float dg[NV][4], db[NV][4];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int q = lane + 64 * i;
        gm[i] = q < nq ? g4[q] : make_float4(0.f, 0.f, 0.f, 0.f);
        bt[i] = (q < nq && act == CAPDEC_ACT_GELU) ? b4[q] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < 4; ++j) dg[i][j] = db[i][j] = 0.f;
    }
    const float inv_d = 1.0f / (float)d;
    // one row ahead: a wave walks 2 - 3 rows here (grid-stride) and each row is two dependent round trips to memory
    // (loads, then the stores behind two wave reductions); with the NEXT row's loads issued before this row's
    // arithmetic the walk costs one exposed latency, not one per row
    float4 nx[NV], nd[NV];
    float nmean = 0.f, nrstd = 0.f;
    auto fetch = [&](int row) {
        if (row < M) {
            nmean = mean_in[row];
            nrstd = rstd_in[row];
            const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * d);
            const float4* dyr = reinterpret_cast<const float4*>(dy + (size_t)row * d);
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int q = lane + 64 * i;
                if (q < nq) {
                    nx[i] = xr[q];
                    nd[i] = dyr[q];
                }
            }
        }
    };
    fetch(blockIdx.x * 4 + wave);
    for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
        const float mean = nmean, rstd = nrstd;
        float4 cx[NV], cd[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            cx[i] = nx[i];
            cd[i] = nd[i];
        }
        fetch(row + gridDim.x * 4);
        float xh[NV][4], g[NV][4];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int q = lane + 64 * i;
            if (q < nq) {
                const float4 xv = cx[i], dv = cd[i];
                const float xa[4] = {xv.x, xv.y, xv.z, xv.w};
                float da[4] = {dv.x, dv.y, dv.z, dv.w};
                const float ga[4] = {gm[i].x, gm[i].y, gm[i].z, gm[i].w};
                const float ba[4] = {bt[i].x, bt[i].y, bt[i].z, bt[i].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    xh[i][j] = (xa[j] - mean) * rstd;
                    if (drop.on()) da[j] = drop.apply(da[j], (uint32_t)row * (uint32_t)d + (uint32_t)(q * 4 + j));
                    if (act == CAPDEC_ACT_GELU) da[j] *= gelu_erf_grad(xh[i][j] * ga[j] + ba[j]);
                    dg[i][j] += da[j] * xh[i][j];
                    db[i][j] += da[j];
                    g[i][j] = da[j] * ga[j];
                    s1 += g[i][j];
                    s2 += g[i][j] * xh[i][j];
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) xh[i][j] = g[i][j] = 0.f;
            }
        }
        const float c1 = wave_sum(s1) * inv_d, c2 = wave_sum(s2) * inv_d;
        float4* dxr = reinterpret_cast<float4*>(dx + (size_t)row * d);
        float4* dxd = dx_drop ? reinterpret_cast<float4*>(dx_drop + (size_t)row * d) : nullptr;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int q = lane + 64 * i;
            if (q < nq) {
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = rstd * (g[i][j] - c1 - xh[i][j] * c2);
                dxr[q] = make_float4(o[0], o[1], o[2], o[3]);
                if (dxd || dxd16) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        o[j] = drop2.on() ? drop2.apply(o[j], (uint32_t)row * (uint32_t)d + (uint32_t)(q * 4 + j)) : o[j];
                    if (dxd) dxd[q] = make_float4(o[0], o[1], o[2], o[3]);
                    if (dxd16) store_bf16x4(dxd16 + (size_t)row * ld16 + q * 4, o);
                }
            }
        }
        if (dxd16) zero_pad_bf16(dxd16 + (size_t)row * ld16, d, ld16, lane);
    }
    // block reduction of the per-wave partial column sums in wave order (fixed: reproducible), in ONE [2][d] LDS
    // image (4 KiB at d = 512; a per-wave image each cost 16 KiB and with it this kernel's place on CUs it shares
    // with the weight-gradient products of the other lane), then one partial row per block
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int i = 0; i < NV; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = (lane + 64 * i) * 4 + j;
                    red[0][c] = w == 0 ? dg[i][j] : red[0][c] + dg[i][j];
                    red[1][c] = w == 0 ? db[i][j] : red[1][c] + db[i][j];
                }
        }
        __syncthreads();
    }
    for (int c = threadIdx.x; c < d; c += 256) {
        part[((size_t)blockIdx.x * 2 + 0) * d + c] = red[0][c];
        part[((size_t)blockIdx.x * 2 + 1) * d + c] = red[1][c];
    }
}

// 16 columns x 16 row groups per block: each thread sums a strided 1/16 of the partial rows
// (independent loads), then the 16 are combined in a fixed order (deterministic).
__global__ __launch_bounds__(256) void ln_bwd_finalize_kernel(const float* __restrict__ part, float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta, int nblk, int d) {
    __shared__ float sh[2][16][17];
    const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    float a = 0.f, b = 0.f;
    if (c < d) {
#pragma unroll 4
        for (int k = rg; k < nblk; k += 16) {
            a += part[((size_t)k * 2 + 0) * d + c];
            b += part[((size_t)k * 2 + 1) * d + c];
        }
    }
    sh[0][rg][cl] = a;
    sh[1][rg][cl] = b;
    __syncthreads();
    if (rg == 0 && c < d) {
        float x = 0.f, y = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            x += sh[0][r][cl];
            y += sh[1][r][cl];
        }
        dgamma[c] = x;
        dbeta[c] = y;
    }
}

static int ln_blocks(int M) {
    int b = (M + 3) / 4;
    return b > LN_MAX_BLOCKS ? LN_MAX_BLOCKS : b;
}

}  // namespace capdec

using namespace capdec;

extern "C" size_t capdec_layernorm_bwd_workspace_bytes(int M, int d) {
    if (M <= 0 || d <= 0) return 0;
    return (size_t)ln_blocks(M) * 2 * d * sizeof(float);
}

#define LN_DISPATCH(NVREQ, CALL)           \
    if ((NVREQ) <= 1) { CALL(1); }         \
    else if ((NVREQ) <= 2) { CALL(2); }    \
    else if ((NVREQ) <= 4) { CALL(4); }    \
    else if ((NVREQ) <= 8) { CALL(8); }    \
    else { CALL(16); }

extern "C" int capdec_layernorm2_fwd(const void* x, const void* gamma1, const void* beta1, const void* add, int W,
                                     const void* gamma2, const void* beta2, void* y, int M, int d, float eps,
                                     capdec_stream_t stream) {
    CAPDEC_REQUIRE(x && gamma1 && beta1 && add && gamma2 && beta2 && y, "capdec_layernorm2_fwd: null pointer");
    CAPDEC_REQUIRE(M > 0 && W > 0 && d > 0 && d % 4 == 0 && d <= 4096, "capdec_layernorm2_fwd: bad shape M=%d W=%d d=%d", M, W, d);
    CAPDEC_REQUIRE(aligned16(x) && aligned16(y) && aligned16(add) && aligned16(gamma1) && aligned16(beta1) && aligned16(gamma2) &&
                       aligned16(beta2), "capdec_layernorm2_fwd: unaligned");
    const int nv = (d + 255) / 256;
    hipStream_t st = (hipStream_t)stream;
#define CALL(NV)                                                                                                        \
    hipLaunchKernelGGL((ln2_fwd_kernel<NV>), dim3((M + 3) / 4), dim3(256), 0, st, (const float*)x, (const float*)gamma1, \
                       (const float*)beta1, (const float*)add, W, (const float*)gamma2, (const float*)beta2, (float*)y, M, \
                       d, eps)
    LN_DISPATCH(nv, CALL)
#undef CALL
    return check_launch("capdec_layernorm2_fwd");
}

static int ln_fwd_launch(const char* what, const void* x, const void* gamma, const void* beta, void* y, void* y16, int ldy16,
                         void* mean, void* rstd, int M, int d, float eps, int act, float drop_p, uint64_t drop_seed,
                         capdec_stream_t stream) {
    CAPDEC_REQUIRE(x && gamma && beta && y, "%s: null pointer", what);
    CAPDEC_REQUIRE(M > 0 && d > 0 && d % 4 == 0 && d <= 4096, "%s: bad shape M=%d d=%d", what, M, d);
    CAPDEC_REQUIRE(aligned16(x) && aligned16(y) && aligned16(gamma) && aligned16(beta), "%s: unaligned", what);
    CAPDEC_REQUIRE((uint64_t)M * (uint64_t)d < (1ull << 32), "%s: M*d exceeds 2^32", what);
    const Dropout dr = make_dropout(drop_p, drop_seed);
    const int nv = (d + 255) / 256;
    hipStream_t st = (hipStream_t)stream;
#define CALL(NV)                                                                                                    \
    hipLaunchKernelGGL((ln_fwd_kernel<NV>), dim3((M + 3) / 4), dim3(256), 0, st, (const float*)x, (const float*)gamma, \
                       (const float*)beta, (float*)y, (float*)mean, (float*)rstd, M, d, eps, act, dr, (__bf16*)y16, ldy16)
    LN_DISPATCH(nv, CALL)
#undef CALL
    return check_launch(what);
}

extern "C" int capdec_layernorm_fwd(const void* x, const void* gamma, const void* beta, void* y, void* mean, void* rstd,
                                    int M, int d, float eps, int act, float drop_p, uint64_t drop_seed,
                                    capdec_stream_t stream) {
    return ln_fwd_launch("capdec_layernorm_fwd", x, gamma, beta, y, nullptr, 0, mean, rstd, M, d, eps, act, drop_p, drop_seed,
                         stream);
}

extern "C" int capdec_layernorm_fwd_y16(const void* x, const void* gamma, const void* beta, void* y, void* y16, int ldy16,
                                        void* mean, void* rstd, int M, int d, float eps, int act, float drop_p,
                                        uint64_t drop_seed, capdec_stream_t stream) {
    CAPDEC_REQUIRE(y16 && aligned16(y16) && ldy16 % 8 == 0 && ldy16 >= d,
                   "capdec_layernorm_fwd_y16: y16 must be a 16-byte aligned bf16 matrix with ldy16 %% 8 == 0, ldy16 >= d");
    return ln_fwd_launch("capdec_layernorm_fwd_y16", x, gamma, beta, y, y16, ldy16, mean, rstd, M, d, eps, act, drop_p,
                         drop_seed, stream);
}

// Shared launch plan of capdec_layernorm_bwd and capdec_layernorm_bwd_noalloc: `part` is the
// [ln_blocks(M)][2][d] block of per-workgroup column partials (dgamma | dbeta), fully overwritten
// by ln_bwd_kernel before ln_bwd_finalize_kernel reads it (nothing in it has to start at zero).
static int ln_bwd_launch(const char* what, const void* dy, const void* x, const void* gamma, const void* beta,
                         const void* mean, const void* rstd, void* dx, void* dx_drop, void* dgamma, void* dbeta, int M,
                         int d, int act, float drop_p, uint64_t drop_seed, float drop2_p, uint64_t drop2_seed,
                         float* part, hipStream_t st) {
    const int nblk = ln_blocks(M);
    const Dropout dr = make_dropout(drop_p, drop_seed), dr2 = make_dropout(drop2_p, drop2_seed);
    const int nv = (d + 255) / 256;
#define CALL(NV)                                                                                                     \
    hipLaunchKernelGGL((ln_bwd_kernel<NV>), dim3(nblk), dim3(256), 0, st, (const float*)dy, (const float*)x,          \
                       (const float*)gamma, (const float*)beta, (const float*)mean, (const float*)rstd, (float*)dx,   \
                       (float*)dx_drop, part, M, d, act, dr, dr2, (__bf16*)nullptr, 0)
    if (nv <= 1) { CALL(1); } else if (nv <= 2) { CALL(2); } else if (nv <= 4) { CALL(4); } else { CALL(8); }
#undef CALL
    int rc = check_launch(what);
    if (rc) return rc;
    hipLaunchKernelGGL(ln_bwd_finalize_kernel, dim3((d + 15) / 16), dim3(256), 0, st, (const float*)part,
                       (float*)dgamma, (float*)dbeta, nblk, d);
    return check_launch(what);
}

static int ln_bwd_check(const char* what, const void* dy, const void* x, const void* gamma, const void* beta,
                        const void* mean, const void* rstd, void* dx, void* dx_drop, void* dgamma, void* dbeta, int M,
                        int d) {
    CAPDEC_REQUIRE(dy && x && gamma && beta && mean && rstd && dx && dgamma && dbeta, "%s: null pointer", what);
    CAPDEC_REQUIRE(M > 0 && d > 0 && d % 4 == 0 && d <= 2048, "%s: bad shape M=%d d=%d", what, M, d);
    CAPDEC_REQUIRE(aligned16(x) && aligned16(dy) && aligned16(dx) && aligned16(gamma) && aligned16(beta) &&
                       (!dx_drop || aligned16(dx_drop)), "%s: unaligned", what);
    CAPDEC_REQUIRE((uint64_t)M * (uint64_t)d < (1ull << 32), "%s: M*d exceeds 2^32", what);
    return CAPDEC_OK;
}

extern "C" int capdec_layernorm_bwd(const void* dy, const void* x, const void* gamma, const void* beta, const void* mean,
                                    const void* rstd, void* dx, void* dx_drop, void* dgamma, void* dbeta, int M, int d,
                                    int act, float drop_p, uint64_t drop_seed, float drop2_p, uint64_t drop2_seed,
                                    void* workspace, size_t workspace_bytes, capdec_stream_t stream) {
    if (int rc = ln_bwd_check("capdec_layernorm_bwd", dy, x, gamma, beta, mean, rstd, dx, dx_drop, dgamma, dbeta, M, d))
        return rc;
    const size_t need = (size_t)ln_blocks(M) * 2 * d * sizeof(float);
    if (!workspace || workspace_bytes < need) {
        set_error("capdec_layernorm_bwd: needs %zu workspace bytes, got %zu", need, workspace_bytes);
        return CAPDEC_ERR_WORKSPACE;
    }
    return ln_bwd_launch("capdec_layernorm_bwd", dy, x, gamma, beta, mean, rstd, dx, dx_drop, dgamma, dbeta, M, d, act,
                         drop_p, drop_seed, drop2_p, drop2_seed, (float*)workspace, (hipStream_t)stream);
}

// The two halves of capdec_layernorm_bwd as separate calls: dx (+ dx_drop) and the per-workgroup column partials
// first, the parameter gradients from those partials later -- e.g. on another stream: dgamma / dbeta are weight
// gradients, nothing on the backward critical path waits for them.
static int ln_bwd_dx_launch(const char* what, const void* dy, const void* x, const void* gamma, const void* beta,
                            const void* mean, const void* rstd, void* dx, void* dx_drop, void* dx_drop16, int lddd,
                            void* partials, size_t partials_bytes, int M, int d, int act, float drop_p, uint64_t drop_seed,
                            float drop2_p, uint64_t drop2_seed, capdec_stream_t stream) {
    float dummy;  // ln_bwd_check wants non-null parameter-gradient pointers; this call writes none
    if (int rc = ln_bwd_check(what, dy, x, gamma, beta, mean, rstd, dx, dx_drop, &dummy, &dummy, M, d)) return rc;
    const size_t need = (size_t)ln_blocks(M) * 2 * d * sizeof(float);
    if (!partials || partials_bytes < need) {
        set_error("%s: needs %zu bytes of partials, got %zu", what, need, partials_bytes);
        return CAPDEC_ERR_WORKSPACE;
    }
    const Dropout dr = make_dropout(drop_p, drop_seed), dr2 = make_dropout(drop2_p, drop2_seed);
    const int nv = (d + 255) / 256, nblk = ln_blocks(M);
    hipStream_t st = (hipStream_t)stream;
#define CALL(NV)                                                                                                     \
    hipLaunchKernelGGL((ln_bwd_kernel<NV>), dim3(nblk), dim3(256), 0, st, (const float*)dy, (const float*)x,          \
                       (const float*)gamma, (const float*)beta, (const float*)mean, (const float*)rstd, (float*)dx,   \
                       (float*)dx_drop, (float*)partials, M, d, act, dr, dr2, (__bf16*)dx_drop16, lddd)
    if (nv <= 1) { CALL(1); } else if (nv <= 2) { CALL(2); } else if (nv <= 4) { CALL(4); } else { CALL(8); }
#undef CALL
    return check_launch(what);
}

extern "C" int capdec_layernorm_bwd_dx(const void* dy, const void* x, const void* gamma, const void* beta, const void* mean,
                                       const void* rstd, void* dx, void* dx_drop, void* partials, size_t partials_bytes,
                                       int M, int d, int act, float drop_p, uint64_t drop_seed, float drop2_p,
                                       uint64_t drop2_seed, capdec_stream_t stream) {
    return ln_bwd_dx_launch("capdec_layernorm_bwd_dx", dy, x, gamma, beta, mean, rstd, dx, dx_drop, nullptr, 0, partials,
                            partials_bytes, M, d, act, drop_p, drop_seed, drop2_p, drop2_seed, stream);
}

extern "C" int capdec_layernorm_bwd_dx16(const void* dy, const void* x, const void* gamma, const void* beta,
                                         const void* mean, const void* rstd, void* dx, void* dx_drop16, int lddd,
                                         void* partials, size_t partials_bytes, int M, int d, int act, float drop_p,
                                         uint64_t drop_seed, float drop2_p, uint64_t drop2_seed, capdec_stream_t stream) {
    CAPDEC_REQUIRE(dx_drop16 && aligned16(dx_drop16) && lddd % 8 == 0 && lddd >= d,
                   "capdec_layernorm_bwd_dx16: dx_drop16 must be a 16-byte aligned bf16 matrix with lddd %% 8 == 0, lddd >= d");
    return ln_bwd_dx_launch("capdec_layernorm_bwd_dx16", dy, x, gamma, beta, mean, rstd, dx, nullptr, dx_drop16, lddd,
                            partials, partials_bytes, M, d, act, drop_p, drop_seed, drop2_p, drop2_seed, stream);
}

extern "C" int capdec_layernorm_bwd_dparams(const void* partials, void* dgamma, void* dbeta, int M, int d,
                                            capdec_stream_t stream) {
    CAPDEC_REQUIRE(partials && dgamma && dbeta && M > 0 && d > 0 && d % 4 == 0 && d <= 2048,
                   "capdec_layernorm_bwd_dparams: bad arguments");
    hipLaunchKernelGGL(ln_bwd_finalize_kernel, dim3((d + 15) / 16), dim3(256), 0, (hipStream_t)stream, (const float*)partials,
                       (float*)dgamma, (float*)dbeta, ln_blocks(M), d);
    return check_launch("capdec_layernorm_bwd_dparams");
}

// ---- the one entry point that allocates nothing (include/capdec.h: capdec_layernorm_bwd_noalloc) -------------
// The column-partial block lives in a file-scope device array instead of a caller workspace.  `scratch` below is
// only a host-side stand-in for that array: each GPU holds its own copy at its own address, looked up once per
// device with hipGetSymbolAddress (first call on that device, with the device current) and kept in g_scratch[];
// host code passes THAT pointer to the kernels, never the symbol.
namespace capdec {
constexpr size_t LN_NOALLOC_BYTES = CAPDEC_LN_NOALLOC_SCRATCH_BYTES;  // 4 MiB per GPU, held until the process ends
static __device__ __align__(256) float scratch[LN_NOALLOC_BYTES / sizeof(float)];
constexpr int LN_MAX_DEVICES = 64;
static float* g_scratch[LN_MAX_DEVICES];  // per device; written once under g_scratch_mu
static std::mutex g_scratch_mu;

static int scratch_of_current_device(float** out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess || dev < 0 || dev >= LN_MAX_DEVICES) {
        set_error("capdec_layernorm_bwd_noalloc: hipGetDevice failed or device index %d >= %d (%s)", dev, LN_MAX_DEVICES,
                  hipGetErrorString(e));
        return CAPDEC_ERR_LAUNCH;
    }
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch[dev]) {
        float* ptr = nullptr;
        e = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !ptr) {
            set_error("capdec_layernorm_bwd_noalloc: hipGetSymbolAddress(scratch) failed on device %d: %s", dev,
                      hipGetErrorString(e));
            return CAPDEC_ERR_LAUNCH;
        }
        g_scratch[dev] = ptr;
    }
    *out = g_scratch[dev];
    return CAPDEC_OK;
}
}  // namespace capdec

extern "C" size_t capdec_layernorm_bwd_noalloc_max_bytes(void) { return LN_NOALLOC_BYTES; }

extern "C" int capdec_layernorm_bwd_noalloc(const void* dy, const void* x, const void* gamma, const void* beta,
                                            const void* mean, const void* rstd, void* dx, void* dx_drop, void* dgamma,
                                            void* dbeta, int M, int d, int act, float drop_p, uint64_t drop_seed,
                                            float drop2_p, uint64_t drop2_seed, capdec_stream_t stream) {
    const char* what = "capdec_layernorm_bwd_noalloc";
    if (int rc = ln_bwd_check(what, dy, x, gamma, beta, mean, rstd, dx, dx_drop, dgamma, dbeta, M, d)) return rc;
    const size_t need = (size_t)ln_blocks(M) * 2 * d * sizeof(float);
    if (need > LN_NOALLOC_BYTES) {
        set_error("%s: M=%d d=%d needs %zu bytes of column partials, the built-in scratch holds %zu "
                  "(use capdec_layernorm_bwd with a caller workspace)", what, M, d, need, (size_t)LN_NOALLOC_BYTES);
        return CAPDEC_ERR_WORKSPACE;
    }
    float* part = nullptr;
    if (int rc = scratch_of_current_device(&part)) return rc;
    return ln_bwd_launch(what, dy, x, gamma, beta, mean, rstd, dx, dx_drop, dgamma, dbeta, M, d, act, drop_p, drop_seed,
                         drop2_p, drop2_seed, part, (hipStream_t)stream);
}
