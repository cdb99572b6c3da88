// Peak probes (measurement infrastructure, SURVEY.md section 8d: "the build must measure its own peaks
// on the box"): a register-resident MFMA loop per wave, and a streaming copy.  bench.py / tools use them
// to state the SUSTAINED matrix rate and HBM bandwidth next to the nominal figures of the roofline.
#include "common.h"

namespace capdec {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// 4 independent accumulators per wave, 8 waves per CU-sized block: the matrix pipe never waits
__global__ __launch_bounds__(256) void probe_mfma_f32_kernel(float* out, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const float a = 1.0f + (threadIdx.x & 7) * 0.125f, b = 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) out[0] = s;  // keep the loop alive
}

__global__ __launch_bounds__(256) void probe_mfma_bf16_kernel(float* out, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) a[j] = (__bf16)(1.0f + j * 0.125f), b[j] = (__bf16)0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) out[0] = s;
}

__global__ __launch_bounds__(256) void probe_copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t n4) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) dst[i] = src[i];
}

}  // namespace capdec

using namespace capdec;

/* Launches `blocks` workgroups of 4 waves, each wave issuing 4*iters MFMAs: returns the FLOPs issued via *flops. */
extern "C" int capdec_probe_mfma(int dtype, int blocks, int iters, void* out, double* flops, capdec_stream_t stream) {
    CAPDEC_REQUIRE(out && blocks > 0 && iters > 0 && (dtype == CAPDEC_F32 || dtype == CAPDEC_BF16), "capdec_probe_mfma: bad arguments");
    const double per_mfma = dtype == CAPDEC_F32 ? 2.0 * 32 * 32 * 2 : 2.0 * 32 * 32 * 16;
    if (flops) *flops = per_mfma * 4.0 * iters * 4.0 * blocks;
    if (dtype == CAPDEC_F32) hipLaunchKernelGGL(probe_mfma_f32_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (float*)out, iters);
    else hipLaunchKernelGGL(probe_mfma_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (float*)out, iters);
    return check_launch("capdec_probe_mfma");
}

extern "C" int capdec_probe_copy(const void* src, void* dst, size_t bytes, capdec_stream_t stream) {
    CAPDEC_REQUIRE(src && dst && bytes >= 16 && bytes % 16 == 0 && aligned16(src) && aligned16(dst), "capdec_probe_copy: bad arguments");
    hipLaunchKernelGGL(probe_copy_kernel, dim3(256 * 16), dim3(256), 0, (hipStream_t)stream, (const float4*)src, (float4*)dst, bytes / 16);
    return check_launch("capdec_probe_copy");
}
