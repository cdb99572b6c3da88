// fp32 -> bf16 copies for the bf16-storage products (gemm_bf16s.hip).  ONE bf16 copy per tensor, in the tensor's own
// orientation -- the GEMM reads either orientation, so nothing is transposed here: a matrix [R, C] (row stride lds
// floats) becomes dst [R][ldd] bf16 with ldd a multiple of 8 >= C; the padding columns C .. ldd-1 are written as
// zeros.  HBM-bound streaming kernels: 32 B in, 16 B out per thread step; round-to-nearest-even (the hardware's
// v_cvt_pk_bf16_f32).
//   capdec_cast_bf16        plain copy
//   capdec_cast_bf16_multi  the same copy of MANY matrices in one launch (every weight operand of a training step:
//                           40 launches of a few microseconds each become one), over a table planned once
//   capdec_ce_dlogits_bf16  the label-smoothed cross-entropy gradient of a logits block, written straight to its
//                           bf16 copy (ref train.py:41-50 backward): the fp32 dlogits matrix is never materialised
#include "common.h"

namespace capdec {

typedef __bf16 bf16x8c __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void load8(const float* s, int c0, int C, bool vec, float (&v)[8]) {
    if (vec && c0 + 8 <= C) {
        const float4 a = *reinterpret_cast<const float4*>(s), b = *reinterpret_cast<const float4*>(s + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = c0 + e < C ? s[e] : 0.f;
    }
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ src, int lds, __bf16* __restrict__ dst, int ldd,
                                                         int R, int C, int vec) {
    const int c8 = ldd / 8;  // 8-element groups per destination row
    const size_t total = (size_t)R * c8;
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (size_t)gridDim.x * 256) {
        const int r = (int)(g / c8), c0 = (int)(g % c8) * 8;
        float v[8];
        load8(src + (size_t)r * lds + c0, c0, C, vec != 0, v);
        bf16x8c o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[e];
        *reinterpret_cast<bf16x8c*>(dst + (size_t)r * ldd + c0) = o;
    }
}

// many matrices, one launch: workgroup b belongs to the table entry whose [first_block, first_block + blocks) holds b
// (a uniform scan over at most CAPDEC_CAST_MULTI_MAX entries) and walks that matrix with its entry's workgroups
__global__ __launch_bounds__(256) void cast_bf16_multi_kernel(const capdec_cast_desc* __restrict__ table, int n) {
    int t = 0;
    while (t + 1 < n && table[t + 1].first_block <= (int)blockIdx.x) ++t;
    const capdec_cast_desc d = table[t];
    const float* src = (const float*)d.src;
    __bf16* dst = (__bf16*)d.dst;
    const int c8 = d.ldd / 8;
    const size_t total = (size_t)d.R * c8;
    for (size_t g = (size_t)((int)blockIdx.x - d.first_block) * 256 + threadIdx.x; g < total; g += (size_t)d.blocks * 256) {
        const int r = (int)(g / c8), c0 = (int)(g % c8) * 8;
        float v[8];
        load8(src + (size_t)r * d.lds + c0, c0, d.C, d.vec != 0, v);
        bf16x8c o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[e];
        *reinterpret_cast<bf16x8c*>(dst + (size_t)r * d.ldd + c0) = o;
    }
}

// the same table walk for plain fp32 copies (capdec_copy_f32_multi): groups of 4 floats, no padding written
__global__ __launch_bounds__(256) void copy_f32_multi_kernel(const capdec_cast_desc* __restrict__ table, int n) {
    int t = 0;
    while (t + 1 < n && table[t + 1].first_block <= (int)blockIdx.x) ++t;
    const capdec_cast_desc d = table[t];
    const float* src = (const float*)d.src;
    float* dst = (float*)d.dst;
    const int c4 = (d.C + 3) / 4;
    const size_t total = (size_t)d.R * c4;
    for (size_t g = (size_t)((int)blockIdx.x - d.first_block) * 256 + threadIdx.x; g < total; g += (size_t)d.blocks * 256) {
        const int r = (int)(g / c4), c0 = (int)(g % c4) * 4;
        const float* s = src + (size_t)r * d.lds + c0;
        float* o = dst + (size_t)r * d.ldd + c0;
        if (d.vec && c0 + 4 <= d.C) {
            *reinterpret_cast<float4*>(o) = *reinterpret_cast<const float4*>(s);
        } else {
            for (int e = 0; e < 4 && c0 + e < d.C; ++e) o[e] = s[e];
        }
    }
}

// captions [B, T1] (row stride ld) -> tgt_in = captions[:, :-1], tgt_out = captions[:, 1:], both dense [B, T1 - 1]
__global__ __launch_bounds__(256) void shift_captions_kernel(const int64_t* __restrict__ cap, int ld, int B, int T1,
                                                              int64_t* __restrict__ tin, int64_t* __restrict__ tout) {
    const int T = T1 - 1;
    const size_t total = (size_t)B * T;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int b = (int)(i / T), t = (int)(i % T);
        const int64_t a = cap[(size_t)b * ld + t], c = cap[(size_t)b * ld + t + 1];
        tin[i] = a;
        tout[i] = c;
    }
}

// one workgroup per logits row: dst[m][v] = gs * (exp(z - lse[m]) - eps/V - (1 - eps) * [v == y[m]])
__global__ __launch_bounds__(256) void ce_dlogits_bf16_kernel(const float* __restrict__ z, int ldl, const float* __restrict__ lse_in,
                                                               const int64_t* __restrict__ targets, int V, float eps, float gs,
                                                               __bf16* __restrict__ dst, int ldd, int vec) {
    const int m = blockIdx.x;
    const float* zr = z + (size_t)m * ldl;
    const float lse = lse_in[m], u = eps / (float)V, w = 1.f - eps;
    int64_t y64 = targets[m];
    const int y = (int)(y64 < 0 ? 0 : (y64 >= V ? V - 1 : y64));
    for (int c0 = threadIdx.x * 8; c0 < ldd; c0 += 256 * 8) {
        float v[8];
        load8(zr + c0, c0, V, vec != 0, v);
        bf16x8c o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float g = __expf(v[e] - lse) - u;
            if (c0 + e == y) g -= w;
            o[e] = (__bf16)(c0 + e < V ? g * gs : 0.f);
        }
        *reinterpret_cast<bf16x8c*>(dst + (size_t)m * ldd + c0) = o;
    }
}

// the same for logits that are already bf16, in place (a thread owns the 8 elements it reads and rewrites)
__global__ __launch_bounds__(256) void ce_dlogits_bf16_inplace_kernel(__bf16* z, int ld, const float* __restrict__ lse_in,
                                                                       const int64_t* __restrict__ targets, int V, float eps, float gs) {
    const int m = blockIdx.x;
    __bf16* zr = z + (size_t)m * ld;
    const float lse = lse_in[m], u = eps / (float)V, w = 1.f - eps;
    int64_t y64 = targets[m];
    const int y = (int)(y64 < 0 ? 0 : (y64 >= V ? V - 1 : y64));
    for (int c0 = threadIdx.x * 8; c0 < ld; c0 += 256 * 8) {
        const bf16x8c v = *reinterpret_cast<const bf16x8c*>(zr + c0);
        bf16x8c o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float g = __expf((float)v[e] - lse) - u;
            if (c0 + e == y) g -= w;
            o[e] = (__bf16)(c0 + e < V ? g * gs : 0.f);
        }
        *reinterpret_cast<bf16x8c*>(zr + c0) = o;
    }
}

}  // namespace capdec

using namespace capdec;

extern "C" int capdec_cast_bf16(const void* src, int lds, void* dst, int ldd, int R, int C, capdec_stream_t stream) {
    CAPDEC_REQUIRE(src && dst && R > 0 && C > 0 && lds >= C, "capdec_cast_bf16: bad arguments R=%d C=%d lds=%d", R, C, lds);
    CAPDEC_REQUIRE(ldd % 8 == 0 && ldd >= C && aligned16(dst),
                   "capdec_cast_bf16: ldd=%d must be a multiple of 8 that covers the row, dst 16-byte aligned", ldd);
    const size_t groups = (size_t)R * (ldd / 8);
    size_t blocks = (groups + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    const int vec = aligned16(src) && lds % 4 == 0;
    hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float*)src, lds,
                       (__bf16*)dst, ldd, R, C, vec);
    return check_launch("capdec_cast_bf16");
}

extern "C" int capdec_cast_bf16_multi_plan(capdec_cast_desc* descs, int n, void* table_dev, int* total_blocks) {
    CAPDEC_REQUIRE(descs && table_dev && total_blocks && n > 0 && n <= CAPDEC_CAST_MULTI_MAX,
                   "capdec_cast_bf16_multi_plan: bad arguments (n=%d, at most %d matrices per table)", n, CAPDEC_CAST_MULTI_MAX);
    int first = 0;
    for (int i = 0; i < n; ++i) {
        capdec_cast_desc& d = descs[i];
        CAPDEC_REQUIRE(d.src && d.dst && d.R > 0 && d.C > 0 && d.lds >= d.C, "capdec_cast_bf16_multi_plan: matrix %d: bad R=%d C=%d lds=%d",
                       i, d.R, d.C, d.lds);
        CAPDEC_REQUIRE(d.ldd % 8 == 0 && d.ldd >= d.C && aligned16(d.dst),
                       "capdec_cast_bf16_multi_plan: matrix %d: ldd=%d must be a multiple of 8 that covers the row, dst 16-byte aligned",
                       i, d.ldd);
        const size_t groups = (size_t)d.R * (d.ldd / 8);
        size_t blocks = (groups + 1023) / 1024;  // four 8-element groups per thread
        if (blocks > 4096) blocks = 4096;
        d.vec = aligned16(d.src) && d.lds % 4 == 0;
        d.first_block = first;
        d.blocks = (int)blocks;
        first += (int)blocks;
    }
    *total_blocks = first;
    if (hipMemcpy(table_dev, descs, (size_t)n * sizeof(capdec_cast_desc), hipMemcpyHostToDevice) != hipSuccess) {
        set_error("capdec_cast_bf16_multi_plan: copying the table to the device failed");
        return CAPDEC_ERR_LAUNCH;
    }
    return 0;
}

extern "C" int capdec_cast_bf16_multi(const void* table_dev, int n, int total_blocks, capdec_stream_t stream) {
    CAPDEC_REQUIRE(table_dev && n > 0 && n <= CAPDEC_CAST_MULTI_MAX && total_blocks > 0,
                   "capdec_cast_bf16_multi: bad arguments (n=%d, total_blocks=%d)", n, total_blocks);
    hipLaunchKernelGGL(cast_bf16_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                       (const capdec_cast_desc*)table_dev, n);
    return check_launch("capdec_cast_bf16_multi");
}

extern "C" int capdec_copy_f32_multi_plan(capdec_cast_desc* descs, int n, void* table_dev, int* total_blocks) {
    CAPDEC_REQUIRE(descs && table_dev && total_blocks && n > 0 && n <= CAPDEC_CAST_MULTI_MAX,
                   "capdec_copy_f32_multi_plan: bad arguments (n=%d, at most %d matrices per table)", n, CAPDEC_CAST_MULTI_MAX);
    int first = 0;
    for (int i = 0; i < n; ++i) {
        capdec_cast_desc& d = descs[i];
        CAPDEC_REQUIRE(d.src && d.dst && d.R > 0 && d.C > 0 && d.lds >= d.C && d.ldd >= d.C,
                       "capdec_copy_f32_multi_plan: matrix %d: bad R=%d C=%d lds=%d ldd=%d", i, d.R, d.C, d.lds, d.ldd);
        const size_t groups = (size_t)d.R * ((d.C + 3) / 4);
        size_t blocks = (groups + 1023) / 1024;  // four groups per thread
        if (blocks > 4096) blocks = 4096;
        d.vec = aligned16(d.src) && aligned16(d.dst) && d.lds % 4 == 0 && d.ldd % 4 == 0;
        d.first_block = first;
        d.blocks = (int)blocks;
        d.reserved = 1;  // fp32 destination
        first += (int)blocks;
    }
    *total_blocks = first;
    if (hipMemcpy(table_dev, descs, (size_t)n * sizeof(capdec_cast_desc), hipMemcpyHostToDevice) != hipSuccess) {
        set_error("capdec_copy_f32_multi_plan: copying the table to the device failed");
        return CAPDEC_ERR_LAUNCH;
    }
    return 0;
}

extern "C" int capdec_copy_f32_multi(const void* table_dev, int n, int total_blocks, capdec_stream_t stream) {
    CAPDEC_REQUIRE(table_dev && n > 0 && n <= CAPDEC_CAST_MULTI_MAX && total_blocks > 0,
                   "capdec_copy_f32_multi: bad arguments (n=%d, total_blocks=%d)", n, total_blocks);
    hipLaunchKernelGGL(copy_f32_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                       (const capdec_cast_desc*)table_dev, n);
    return check_launch("capdec_copy_f32_multi");
}

extern "C" int capdec_shift_captions(const int64_t* captions, int ld, int B, int T1, int64_t* tgt_in, int64_t* tgt_out,
                                     capdec_stream_t stream) {
    CAPDEC_REQUIRE(captions && tgt_in && tgt_out && B > 0 && T1 > 1 && ld >= T1, "capdec_shift_captions: bad arguments B=%d T1=%d ld=%d",
                   B, T1, ld);
    const size_t total = (size_t)B * (T1 - 1);
    size_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(shift_captions_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, captions, ld, B, T1, tgt_in,
                       tgt_out);
    return check_launch("capdec_shift_captions");
}

extern "C" int capdec_ce_dlogits_bf16(const void* logits, int ldl, const void* lse, const int64_t* targets, int M, int V,
                                      float eps, float grad_scale, void* dst, int ldd, capdec_stream_t stream) {
    CAPDEC_REQUIRE(logits && lse && targets && dst && M > 0 && V > 0 && ldl >= V, "capdec_ce_dlogits_bf16: bad arguments");
    CAPDEC_REQUIRE(ldd % 8 == 0 && ldd >= V && aligned16(dst) && eps >= 0.f && eps < 1.f,
                   "capdec_ce_dlogits_bf16: ldd=%d must be a multiple of 8 that covers V, dst 16-byte aligned, 0 <= eps < 1", ldd);
    const int vec = aligned16(logits) && ldl % 4 == 0;
    hipLaunchKernelGGL(ce_dlogits_bf16_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, (const float*)logits, ldl,
                       (const float*)lse, targets, V, eps, grad_scale, (__bf16*)dst, ldd, vec);
    return check_launch("capdec_ce_dlogits_bf16");
}

extern "C" int capdec_ce_dlogits_bf16_inplace(void* z16, int ld, const void* lse, const int64_t* targets, int M, int V, float eps,
                                              float grad_scale, capdec_stream_t stream) {
    CAPDEC_REQUIRE(z16 && lse && targets && M > 0 && V > 0, "capdec_ce_dlogits_bf16_inplace: bad arguments");
    CAPDEC_REQUIRE(ld % 8 == 0 && ld >= V && aligned16(z16) && eps >= 0.f && eps < 1.f,
                   "capdec_ce_dlogits_bf16_inplace: ld=%d must be a multiple of 8 that covers V, z16 16-byte aligned, 0 <= eps < 1", ld);
    hipLaunchKernelGGL(ce_dlogits_bf16_inplace_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, (__bf16*)z16, ld,
                       (const float*)lse, targets, V, eps, grad_scale);
    return check_launch("capdec_ce_dlogits_bf16_inplace");
}
