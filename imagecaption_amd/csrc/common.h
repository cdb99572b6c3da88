// Shared device/host helpers for the capdec gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/capdec.h"

namespace capdec {

// ---- error plumbing -------------------------------------------------------------------------
void set_error(const char* fmt, ...);
int check_launch(const char* what);  // hipGetLastError -> 0 or CAPDEC_ERR_LAUNCH

#define CAPDEC_REQUIRE(cond, ...)                \
    do {                                         \
        if (!(cond)) {                           \
            capdec::set_error(__VA_ARGS__);      \
            return CAPDEC_ERR_ARG;               \
        }                                        \
    } while (0)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- counter-based dropout RNG ---------------------------------------------------------------
// keep(i) = mix(mix(i + s1) ^ s2) >= thresh, (s1, s2) = the two halves of a 64-bit per-site seed.
// mix = "lowbias32" integer finaliser (bijective, full avalanche).  Forward and backward kernels
// regenerate the same mask from (seed, element index); no mask tensor is ever stored.
__host__ __device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}
struct Dropout {
    uint32_t s1, s2, thresh;  // thresh = round(p * 2^32); 0 => disabled
    float scale;              // 1 / (1 - p)
    __host__ __device__ __forceinline__ bool on() const { return thresh != 0; }
    __host__ __device__ __forceinline__ bool keep(uint32_t idx) const {
        return mix32(mix32(idx + s1) ^ s2) >= thresh;
    }
    __host__ __device__ __forceinline__ float apply(float v, uint32_t idx) const {
        return keep(idx) ? v * scale : 0.0f;
    }
};
static inline Dropout make_dropout(float p, uint64_t seed) {
    Dropout d;
    d.s1 = (uint32_t)seed;
    d.s2 = (uint32_t)(seed >> 32);
    if (p <= 0.0f) {
        d.thresh = 0;
        d.scale = 1.0f;
    } else {
        double t = (double)p * 4294967296.0;
        d.thresh = t >= 4294967295.0 ? 4294967295u : (uint32_t)(t + 0.5);
        if (d.thresh == 0) d.thresh = 1;
        d.scale = 1.0f / (1.0f - p);
    }
    return d;
}

// ---- math -----------------------------------------------------------------------------------
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

}  // namespace capdec
