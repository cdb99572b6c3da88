// Cross-attention branch for a memory of ONE position (S = 1: the reference's own configuration, the
// projected ViT CLS vector, ref models.py:104).  Softmax over one key is 1, so head h's context is that
// key's value v[b, h-slice], kept or dropped as a whole by the attention-probability dropout
// (torch/nn/functional.py:6613-6631 applies dropout to the [T, 1] probabilities), and the out-projection
// of the context is a sum of per-head, per-IMAGE constants:
//     out_proj(c)[b,t,:] = sum_h keep_h(b,t)/(1-p) * U[b,h,:],      U[b,h,:] = Wo[:, h-slice] v[b, h-slice].
// U is a tiny GEMM (B*H rows); these kernels do the per-token part, replacing a [B*T, d] x [d, d] product
// forward and two backward, memory-bound instead (see include/capdec.h).  Same dropout element indices as
// the attention kernels (probabilities) and the GEMM epilogue (output dropout).
#include "common.h"

namespace capdec {

struct Cross1Params {
    const float* U;       // [B*H, d]
    const float* bias;    // [d]
    const float* addend;  // [B*T, d]
    float* out;           // [B*T, d]
    const float* g;       // [B*T, d]  gradient w.r.t. the branch output (output dropout already applied)
    float* dU;            // [B*H, d]
    float* dbias_part;    // [B, d]    per-image column sums of g
    int B, T, H, d;
    Dropout attn, outd;
};

__global__ __launch_bounds__(256) void cross1_fwd_kernel(Cross1Params p) {
    const int c4 = p.d / 4;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)p.B * p.T * c4) return;
    const int m = (int)(idx / c4), c = (int)(idx % c4);
    const int b = m / p.T, t = m % p.T;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int h = 0; h < p.H; ++h) {  // fixed order: bitwise reproducible
        const int bh = b * p.H + h;
        const float w = p.attn.on() ? (p.attn.keep((uint32_t)bh * (uint32_t)p.T + (uint32_t)t) ? p.attn.scale : 0.f) : 1.f;
        const float4 u = *reinterpret_cast<const float4*>(p.U + (size_t)bh * p.d + c * 4);
        acc.x += w * u.x; acc.y += w * u.y; acc.z += w * u.z; acc.w += w * u.w;
    }
    const float4 bi = *reinterpret_cast<const float4*>(p.bias + c * 4);
    float v[4] = {acc.x + bi.x, acc.y + bi.y, acc.z + bi.z, acc.w + bi.w};
    if (p.outd.on()) {
        const uint32_t e = (uint32_t)m * (uint32_t)p.d + (uint32_t)c * 4u;  // the GEMM epilogue's index m*N + n
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = p.outd.apply(v[j], e + j);
    }
    const float4 a = *reinterpret_cast<const float4*>(p.addend + (size_t)m * p.d + c * 4);
    *reinterpret_cast<float4*>(p.out + (size_t)m * p.d + c * 4) = make_float4(v[0] + a.x, v[1] + a.y, v[2] + a.z, v[3] + a.w);
}

// block = (image b, 16 float4 columns) x 16 token lanes; every thread walks t = lane, lane+16, ... and keeps one
// accumulator per head (+ the unmasked sum for the bias gradient); the 16 lanes are folded in fixed order.
template <int H>
__global__ __launch_bounds__(256) void cross1_bwd_kernel(Cross1Params p) {
    __shared__ float4 red[16][17];
    const int c4 = p.d / 4, groups = c4 / 16;
    const int b = blockIdx.x / groups, c = (blockIdx.x % groups) * 16 + (threadIdx.x & 15);
    const int tl = threadIdx.x >> 4, cl = threadIdx.x & 15;
    float4 acc[H + 1];
#pragma unroll
    for (int h = 0; h <= H; ++h) acc[h] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = tl; t < p.T; t += 16) {
        const float4 g = *reinterpret_cast<const float4*>(p.g + ((size_t)b * p.T + t) * p.d + c * 4);
#pragma unroll
        for (int h = 0; h < H; ++h) {
            const int bh = b * H + h;
            const float w = p.attn.on() ? (p.attn.keep((uint32_t)bh * (uint32_t)p.T + (uint32_t)t) ? p.attn.scale : 0.f) : 1.f;
            acc[h].x += w * g.x; acc[h].y += w * g.y; acc[h].z += w * g.z; acc[h].w += w * g.w;
        }
        acc[H].x += g.x; acc[H].y += g.y; acc[H].z += g.z; acc[H].w += g.w;
    }
#pragma unroll
    for (int h = 0; h <= H; ++h) {
        __syncthreads();
        red[tl][cl] = acc[h];
        __syncthreads();
        if (tl == 0) {
            float4 s = red[0][cl];
            for (int j = 1; j < 16; ++j) {
                const float4 a = red[j][cl];
                s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
            }
            float* dst = h < H ? p.dU + (size_t)(b * H + h) * p.d : p.dbias_part + (size_t)b * p.d;
            *reinterpret_cast<float4*>(dst + c * 4) = s;
        }
    }
}

// The two reshapes around the B*H-row products of this branch (they used to be torch elementwise / diagonal / sum
// calls on the training path):
//   spread: vblk[(b, h), j] = v[b, j] inside head h's slice, 0 elsewhere (the out-projection's operand);
//   gather: dk[b, :] = 0 (exactly, as torch computes it), dv[b, j] = dvblk[(b, head of j), j], and
//           dbias[j] = sum_b dbias_part[b, j] in image order (fixed: bitwise reproducible).
__global__ __launch_bounds__(256) void cross1_spread_kernel(const float* __restrict__ v, int ldv, float* __restrict__ vblk,
                                                            int B, int H, int d) {
    const int c4 = d / 4, dh = d / H;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)B * H * c4) return;
    const int row = (int)(idx / c4), c = (int)(idx % c4);
    const int b = row / H, h = row % H;
    float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((c * 4) / dh == h) x = *reinterpret_cast<const float4*>(v + (size_t)b * ldv + c * 4);
    *reinterpret_cast<float4*>(vblk + (size_t)row * d + c * 4) = x;
}

__global__ __launch_bounds__(256) void cross1_gather_kernel(const float* __restrict__ dvblk, const float* __restrict__ dbias_part,
                                                            float* __restrict__ dk, int lddk, float* __restrict__ dv, int lddv,
                                                            float* __restrict__ dbias, int B, int H, int d) {
    const int c4 = d / 4, dh = d / H;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx < (size_t)B * c4) {
        const int b = (int)(idx / c4), c = (int)(idx % c4);
        const int h = (c * 4) / dh;
        *reinterpret_cast<float4*>(dk + (size_t)b * lddk + c * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(dv + (size_t)b * lddv + c * 4) =
            *reinterpret_cast<const float4*>(dvblk + (size_t)(b * H + h) * d + c * 4);
    }
    if (idx < (size_t)c4) {  // the first c4 threads also fold the per-image bias partials
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int b = 0; b < B; ++b) {
            const float4 a = *reinterpret_cast<const float4*>(dbias_part + (size_t)b * d + idx * 4);
            s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
        }
        *reinterpret_cast<float4*>(dbias + idx * 4) = s;
    }
}

}  // namespace capdec

using namespace capdec;

static int cross1_check(const char* what, int B, int T, int H, int d) {
    CAPDEC_REQUIRE(B > 0 && T > 0 && H > 0 && d > 0 && d % 64 == 0 && d % H == 0, "%s: bad shape B=%d T=%d H=%d d=%d (d %% 64 == 0)",
                   what, B, T, H, d);
    CAPDEC_REQUIRE((uint64_t)B * T * d < (1ull << 32) && (uint64_t)B * H * T < (1ull << 32), "%s: B*T*d exceeds 2^32", what);
    return 0;
}

extern "C" int capdec_cross1_fwd(const void* U, const void* bias, const void* addend, void* out, int B, int T, int H, int d,
                                 float attn_p, uint64_t attn_seed, float out_p, uint64_t out_seed, capdec_stream_t stream) {
    if (int rc = cross1_check("capdec_cross1_fwd", B, T, H, d)) return rc;
    CAPDEC_REQUIRE(U && bias && addend && out && aligned16(U) && aligned16(bias) && aligned16(addend) && aligned16(out),
                   "capdec_cross1_fwd: operands must be non-null and 16-byte aligned");
    Cross1Params p;
    memset(&p, 0, sizeof(p));
    p.U = (const float*)U; p.bias = (const float*)bias; p.addend = (const float*)addend; p.out = (float*)out;
    p.B = B; p.T = T; p.H = H; p.d = d;
    p.attn = make_dropout(attn_p, attn_seed);
    p.outd = make_dropout(out_p, out_seed);
    const size_t total = (size_t)B * T * (d / 4);
    hipLaunchKernelGGL(cross1_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("capdec_cross1_fwd");
}

extern "C" int capdec_cross1_bwd(const void* g, void* dU, void* dbias_part, int B, int T, int H, int d, float attn_p,
                                 uint64_t attn_seed, capdec_stream_t stream) {
    if (int rc = cross1_check("capdec_cross1_bwd", B, T, H, d)) return rc;
    CAPDEC_REQUIRE(g && dU && dbias_part && aligned16(g) && aligned16(dU) && aligned16(dbias_part),
                   "capdec_cross1_bwd: operands must be non-null and 16-byte aligned");
    CAPDEC_REQUIRE(H == 1 || H == 2 || H == 4 || H == 8 || H == 16, "capdec_cross1_bwd: %d heads not in {1,2,4,8,16}", H);
    Cross1Params p;
    memset(&p, 0, sizeof(p));
    p.g = (const float*)g; p.dU = (float*)dU; p.dbias_part = (float*)dbias_part;
    p.B = B; p.T = T; p.H = H; p.d = d;
    p.attn = make_dropout(attn_p, attn_seed);
    p.outd = make_dropout(0.f, 0);
    const dim3 grid(B * (d / 64));
    hipStream_t st = (hipStream_t)stream;
    switch (H) {
        case 1: hipLaunchKernelGGL(cross1_bwd_kernel<1>, grid, dim3(256), 0, st, p); break;
        case 2: hipLaunchKernelGGL(cross1_bwd_kernel<2>, grid, dim3(256), 0, st, p); break;
        case 4: hipLaunchKernelGGL(cross1_bwd_kernel<4>, grid, dim3(256), 0, st, p); break;
        case 8: hipLaunchKernelGGL(cross1_bwd_kernel<8>, grid, dim3(256), 0, st, p); break;
        default: hipLaunchKernelGGL(cross1_bwd_kernel<16>, grid, dim3(256), 0, st, p); break;
    }
    return check_launch("capdec_cross1_bwd");
}

extern "C" int capdec_cross1_spread(const void* v, int ldv, void* vblk, int B, int H, int d, capdec_stream_t stream) {
    if (int rc = cross1_check("capdec_cross1_spread", B, 1, H, d)) return rc;
    CAPDEC_REQUIRE(v && vblk && aligned16(v) && aligned16(vblk) && ldv % 4 == 0 && ldv >= d && (d / H) % 4 == 0,
                   "capdec_cross1_spread: operands must be non-null, 16-byte aligned, ldv %% 4 == 0, ldv >= d, head dim %% 4 == 0");
    const size_t total = (size_t)B * H * (d / 4);
    hipLaunchKernelGGL(cross1_spread_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)v, ldv, (float*)vblk, B, H, d);
    return check_launch("capdec_cross1_spread");
}

extern "C" int capdec_cross1_gather(const void* dvblk, const void* dbias_part, void* dk, int lddk, void* dv, int lddv,
                                    void* dbias, int B, int H, int d, capdec_stream_t stream) {
    if (int rc = cross1_check("capdec_cross1_gather", B, 1, H, d)) return rc;
    CAPDEC_REQUIRE(dvblk && dbias_part && dk && dv && dbias && aligned16(dvblk) && aligned16(dbias_part) && aligned16(dk) &&
                       aligned16(dv) && aligned16(dbias) && lddk % 4 == 0 && lddv % 4 == 0 && lddk >= d && lddv >= d &&
                       (d / H) % 4 == 0,
                   "capdec_cross1_gather: operands must be non-null, 16-byte aligned, ld %% 4 == 0, ld >= d, head dim %% 4 == 0");
    const size_t total = (size_t)B * (d / 4);
    hipLaunchKernelGGL(cross1_gather_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)dvblk, (const float*)dbias_part, (float*)dk, lddk, (float*)dv, lddv, (float*)dbias, B, H, d);
    return check_launch("capdec_cross1_gather");
}
