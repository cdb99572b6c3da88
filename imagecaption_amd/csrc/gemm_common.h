// Shared between gemm_f32.hip and gemm_bf16.hip: kernel parameters, the fused epilogue, plans.
#pragma once
#include "common.h"

namespace capdec {

struct GemmParams {
    const float* A;
    const float* B;
    float* C;
    int M, N, K, lda, ldb, ldc;
    const float* bias;
    const float* addend;
    float* pre;
    const float* aux;
    float* colsum;
    int ld_addend, act;
    Dropout drop;
    int ksplit;        // K elements per z-slice (multiple of the k-tile)
    float* ws;         // [S][M][N] partial products when gridDim.z > 1
    float* ws_colsum;  // [S][M]
    int tiles_n;
};

__device__ __forceinline__ float epilogue_apply(const GemmParams& p, float v, int m, int n) {
    if (p.bias) v += p.bias[n];
    const size_t o = (size_t)m * p.ldc + n;
    if (p.pre) p.pre[o] = v;
    if (p.act == CAPDEC_ACT_GELU) v = gelu_erf(v);
    else if (p.act == CAPDEC_ACT_GELU_BWD) v *= gelu_erf_grad(p.aux[o]);
    if (p.drop.on()) v = p.drop.apply(v, (uint32_t)m * (uint32_t)p.N + (uint32_t)n);
    if (p.addend) v += p.addend[(size_t)m * p.ld_addend + n];
    return v;
}

struct GemmPlan {
    int bm, bn, bk, split, ksplit;
};
struct GemmForce {
    int bm, bn, split, bk;
};
extern GemmForce g_gemm_force;  // tuning hook (capdec_gemm_tune)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

GemmPlan plan_gemm_bf16(int M, int N, int K);
void launch_gemm_bf16(const GemmParams& p, const GemmPlan& pl, bool AK, bool BKM, bool vec, hipStream_t st);
GemmPlan plan_gemm_f32x3(int M, int N, int K);
int launch_gemm_f32x3(const GemmParams& p, const GemmPlan& pl, bool AK, bool BKM, bool vec, hipStream_t st);

}  // namespace capdec
