// Small HBM-bound elementwise kernels used by the sub-module-at-a-time call path
// (the reference's decode loop calls .embedding / .pos_encoder / ... separately, train.py:317-332).
#include "common.h"

namespace capdec {

// out[r,:] = dropout(x[r,:] + pe[t0 + (r / div) % mod, :])
__global__ __launch_bounds__(256) void pe_add_kernel(const float* __restrict__ x, const float* __restrict__ pe,
                                                      float* __restrict__ out, int R, int d, int div, int mod, int t0,
                                                      Dropout drop) {
    const int nq = d >> 2;
    const size_t total = (size_t)R * nq, stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        const int r = (int)(i / nq), q = (int)(i % nq);
        const int t = t0 + (r / div) % mod;
        const float4 a = reinterpret_cast<const float4*>(x)[i];
        const float4 b = reinterpret_cast<const float4*>(pe + (size_t)t * d)[q];
        float o[4] = {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w};
        if (drop.on()) {
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = drop.apply(o[j], (uint32_t)(i * 4 + j));
        }
        reinterpret_cast<float4*>(out)[i] = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// out[i] = dy[i] * keep(i)/(1-p) * (pre ? gelu'(pre[i]) : 1)      (n % 4 == 0)
__global__ __launch_bounds__(256) void act_dropout_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ pre,
                                                               float* __restrict__ out, size_t n4, Dropout drop) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const float4 a = reinterpret_cast<const float4*>(dy)[i];
        float o[4] = {a.x, a.y, a.z, a.w};
        if (drop.on()) {
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = drop.apply(o[j], (uint32_t)(i * 4 + j));
        }
        if (pre) {
            const float4 u = reinterpret_cast<const float4*>(pre)[i];
            o[0] *= gelu_erf_grad(u.x); o[1] *= gelu_erf_grad(u.y); o[2] *= gelu_erf_grad(u.z); o[3] *= gelu_erf_grad(u.w);
        }
        reinterpret_cast<float4*>(out)[i] = make_float4(o[0], o[1], o[2], o[3]);
    }
}

}  // namespace capdec

using namespace capdec;

static int blocks_for(size_t n) {
    size_t b = (n + 255) / 256;
    return (int)(b > 2048 ? 2048 : (b ? b : 1));
}

extern "C" int capdec_pe_add(const void* x, const void* pe, void* out, int R, int d, int div, int mod, int t0,
                             float drop_p, uint64_t drop_seed, capdec_stream_t stream) {
    CAPDEC_REQUIRE(x && pe && out && R > 0 && d > 0 && d % 4 == 0 && div > 0 && mod > 0 && t0 >= 0,
                   "capdec_pe_add: bad arguments");
    CAPDEC_REQUIRE(aligned16(x) && aligned16(pe) && aligned16(out), "capdec_pe_add: unaligned");
    CAPDEC_REQUIRE((uint64_t)R * d < (1ull << 32), "capdec_pe_add: R*d exceeds 2^32");
    hipLaunchKernelGGL(pe_add_kernel, dim3(blocks_for((size_t)R * d / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)x, (const float*)pe, (float*)out, R, d, div, mod, t0, make_dropout(drop_p, drop_seed));
    return check_launch("capdec_pe_add");
}

extern "C" int capdec_act_dropout_bwd(const void* dy, const void* pre, void* out, size_t n, float drop_p,
                                      uint64_t drop_seed, capdec_stream_t stream) {
    CAPDEC_REQUIRE(dy && out && n > 0 && n % 4 == 0 && n < (1ull << 32), "capdec_act_dropout_bwd: bad arguments");
    CAPDEC_REQUIRE(aligned16(dy) && aligned16(out) && (!pre || aligned16(pre)), "capdec_act_dropout_bwd: unaligned");
    hipLaunchKernelGGL(act_dropout_bwd_kernel, dim3(blocks_for(n / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)dy, (const float*)pre, (float*)out, n / 4, make_dropout(drop_p, drop_seed));
    return check_launch("capdec_act_dropout_bwd");
}
