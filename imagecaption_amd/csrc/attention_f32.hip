// Fused attention (scores + softmax + context) for the caption decoder, fp32 on the gfx950 matrix
// cores (v_mfma_f32_32x32x2_f32), forward and flash-style backward.  See include/capdec.h.
//
// One wave (64-thread workgroup) owns one (batch, head, 32-row tile); nothing crosses workgroups,
// so there are no atomics and results are bitwise reproducible.
//   forward / dQ : the wave owns 32 queries and walks 32-key tiles.  It computes S^T = K Q^T, so a
//                  lane holds 16 scores of ONE query (its partner lane^32 the other 16): the softmax
//                  row reduction is 15 fmax + one cross-half shuffle.  The S^T accumulator registers
//                  are then fed straight back as the A operand of P V (or dS K): MFMA sums over the
//                  accumulator's row index (the key), so no lane movement and no LDS round trip.
//   dK / dV      : the wave owns 32 keys and walks 32-query tiles with S = Q K^T (query on the
//                  accumulator rows), whose registers are the A operand of P^T dO and dS^T Q.
// K/V (or Q/dO) tiles are staged through LDS as [row][dh+4]: the +4 padding makes the 16-lane
// ds_read_b128 groups of the "row on lane" operand reads bank-conflict free, and the same image
// serves the "row from register index" reads (32 consecutive dwords per half-wave).
// Per-query scalars (softmax rescale, 1/l, lse, delta) reach the register-row layout through a
// 32-float LDS line read back as four float4.
#include "common.h"
#include "attention_common.h"

namespace capdec {

// A 32 x DH tile moves global -> registers (coalesced: 16 consecutive lanes read one 256-B row
// piece) -> LDS, in two separate steps so the global loads of the NEXT tile can be in flight
// while the current one is being multiplied (register prefetch, no extra LDS).
// b16 (wave-uniform): the matrix is bf16 (g points at its first element, ldg in bf16 elements); the four elements
// (8 B) of a lane stay PACKED in .x / .y of its register slot -- nothing here depends on the loaded value, so the
// load stays in flight under the MFMAs as the fp32 ones do -- and tile_lstore widens them on the way into LDS.
template <int DH>
__device__ __forceinline__ void tile_gload(float4* regs, const float* g, int ldg, int row0, int T, int lane, bool b16 = false) {
#pragma unroll
    for (int i = 0; i < DH / 8; ++i) {
        const int f = lane + 64 * i;
        const int row = f / (DH / 4), c4 = f % (DH / 4);
        const int t = row0 + row;
        regs[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t < T) {
            if (b16) {
                const float2 u = *reinterpret_cast<const float2*>(reinterpret_cast<const __bf16*>(g) + (size_t)t * ldg + c4 * 4);
                regs[i].x = u.x;
                regs[i].y = u.y;
            } else {
                regs[i] = *reinterpret_cast<const float4*>(g + (size_t)t * ldg + c4 * 4);
            }
        }
    }
}
// first element of head h of image b in a K / V matrix that is fp32 or (b16) bf16
__device__ __forceinline__ const float* kv_base(const float* m, size_t elem_off, bool b16) {
    return b16 ? reinterpret_cast<const float*>(reinterpret_cast<const __bf16*>(m) + elem_off) : m + elem_off;
}
template <int DH>
__device__ __forceinline__ void tile_lstore(float* s, const float4* regs, float mul, int lane, bool b16 = false) {
    constexpr int LD = DH + 4;
#pragma unroll
    for (int i = 0; i < DH / 8; ++i) {
        const int f = lane + 64 * i;
        const int row = f / (DH / 4), c4 = f % (DH / 4);
        float4 v = regs[i];
        if (b16) {  // .x / .y hold four packed bf16 values (tile_gload): widen exactly
            const unsigned int ux = __float_as_uint(v.x), uy = __float_as_uint(v.y);
            v = make_float4(__uint_as_float(ux << 16), __uint_as_float(ux & 0xffff0000u), __uint_as_float(uy << 16),
                            __uint_as_float(uy & 0xffff0000u));
        }
        v.x *= mul; v.y *= mul; v.z *= mul; v.w *= mul;
        *reinterpret_cast<float4*>(&s[row * LD + c4 * 4]) = v;
    }
}
// fragment of the "row on lane" MFMA operand from an LDS image: row = lane&31, k = 8*jj + 4*half + c.
// (Fragment-shaped loads straight from global touch 32 lines per instruction; staging the tile
// through LDS in full rows first is the cheaper path on this chip.)
template <int DH>
__device__ __forceinline__ void lds_frag(float4* f, const float* s, int l31, int half) {
    constexpr int LD = DH + 4;
#pragma unroll
    for (int jj = 0; jj < DH / 8; ++jj) f[jj] = *reinterpret_cast<const float4*>(&s[l31 * LD + 8 * jj + 4 * half]);
}

// acc += A(lds rows on lanes) * B(register fragment)^T  over the head dimension
template <int DH>
__device__ __forceinline__ f32x16 mma_lds_reg(const float* s, const float4* bf, int l31, int half) {
    constexpr int LD = DH + 4;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int jj = 0; jj < DH / 8; ++jj) {
        const float4 a = *reinterpret_cast<const float4*>(&s[l31 * LD + 8 * jj + 4 * half]);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bf[jj].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bf[jj].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bf[jj].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bf[jj].w, acc, 0, 0, 0);
    }
    return acc;
}
// acc = A(register fragment) * B(lds rows on lanes)^T
template <int DH>
__device__ __forceinline__ f32x16 mma_reg_lds(const float4* af, const float* s, int l31, int half) {
    constexpr int LD = DH + 4;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int jj = 0; jj < DH / 8; ++jj) {
        const float4 b = *reinterpret_cast<const float4*>(&s[l31 * LD + 8 * jj + 4 * half]);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[jj].x, b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[jj].y, b.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[jj].z, b.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[jj].w, b.w, acc, 0, 0, 0);
    }
    return acc;
}
// out[dt] += X^T * S  where X = accumulator-layout tile x[16] (rows = its register index) and S
// is an LDS image whose row index matches X's row index.
template <int DH>
__device__ __forceinline__ void mma_acc_lds(f32x16* out, const float* x, const float* s, int l31, int half) {
    constexpr int LD = DH + 4;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float* srow = s + row_of(r, half) * LD + l31;
#pragma unroll
        for (int dt = 0; dt < DH / 32; ++dt)
            out[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[r], srow[dt * 32], out[dt], 0, 0, 0);
    }
}

// spread a per-lane-row scalar (valid in lanes 0..31, row = lane) to the register-row layout
__device__ __forceinline__ void spread_rows(float* line, float val, float* outv, int lane, int half) {
    __syncthreads();
    if (lane < 32) line[lane] = val;
    __syncthreads();
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 t = *reinterpret_cast<const float4*>(&line[8 * g + 4 * half]);
        outv[4 * g + 0] = t.x; outv[4 * g + 1] = t.y; outv[4 * g + 2] = t.z; outv[4 * g + 3] = t.w;
    }
}

template <int DH>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(DH <= 64 ? 3 : 1))) void attn_fwd_kernel(AttnParams p) {
    constexpr int LD = DH + 4;
    // ONE tile image, holding K while the scores are computed and V while they are applied: half the
    // LDS and (with a single staging register set that always carries the next tile needed) ~100
    // fewer registers than separate K and V images, which is what lets a third wave live on a SIMD.
    __shared__ __attribute__((aligned(16))) float Ts[32 * LD];
    __shared__ __attribute__((aligned(16))) float line[32];
    const int lane = threadIdx.x, l31 = lane & 31, half = lane >> 5;
    const int nqt = (p.Tq + 31) / 32;
    // causal: the tiles of one (b, h) differ in length (query tile i walks i + 1 key tiles), so the
    // longest ones are dispatched first and the short ones fill the tail of the launch
    int bh, qt;
    attn_item(p.B * p.H, nqt, p.causal != 0, bh, qt);
    if (p.causal) qt = nqt - 1 - qt;
    const int b = bh / p.H, h = bh % p.H;
    const int q0 = qt * 32, off = p.Tk - p.Tq;
    const float* qg = p.q + (size_t)b * p.Tq * p.ldq + h * DH;
    const bool b16 = p.kv_bf16 != 0;
    const float* kg = kv_base(p.k, (size_t)b * p.Tk * p.ldk + h * DH, b16);
    const float* vg = kv_base(p.v, (size_t)b * p.Tk * p.ldv + h * DH, b16);

    // Diagnostic builds only (tools/attn_stamps.py; never in the shipped library): -DCAPDEC_ATTN_STAMPS records
    // s_memtime at the phase boundaries of every wave, -DCAPDEC_ATTN_NOMMA replaces the matrix products by a few
    // vector operations (what the kernel costs with its arithmetic taken out).
#ifdef CAPDEC_ATTN_STAMPS
    unsigned long long st_[8];
    int sti_ = 0;
#define STAMP() do { if (sti_ < 8) st_[sti_++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP() do { } while (0)
#endif
#ifdef CAPDEC_ATTN_TWICE
    for (int rep_ = 0; rep_ < CAPDEC_ATTN_TWICE; ++rep_) {
    asm volatile("" ::: "memory");
#else
    {
#endif
    STAMP();
    float4 qf[DH / 8], tr[DH / 8];
    const int query = q0 + l31;
    int kmax = p.Tk;
    if (p.causal) kmax = min(p.Tk, q0 + 31 + off + 1);
    const int nkt = kmax > 0 ? (kmax + 31) / 32 : 0;
    tile_gload<DH>(qf, qg, p.ldq, q0, p.Tq, lane);
    if (nkt > 0) tile_gload<DH>(tr, kg, p.ldk, 0, p.Tk, lane, b16);
    tile_lstore<DH>(Ts, qf, p.scale, lane);
    STAMP();  // 1: Q arrived
    __syncthreads();
    lds_frag<DH>(qf, Ts, l31, half);
    f32x16 oacc[DH / 32];
#pragma unroll
    for (int dt = 0; dt < DH / 32; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    for (int kt = 0; kt < nkt; ++kt) {
        const int k0 = kt * 32;
        __syncthreads();
        tile_lstore<DH>(Ts, tr, 1.f, lane, b16);  // K tile
        if (kt == 0) STAMP();  // 2: K0 arrived and stored
        __syncthreads();
        tile_gload<DH>(tr, vg, p.ldv, k0, p.Tk, lane, b16);  // V tile flies under the score MFMAs
#ifdef CAPDEC_ATTN_NOMMA
        f32x16 s;
        for (int r = 0; r < 16; ++r) s[r] = Ts[l31 * LD + r] * qf[r % (DH / 8)].x;
#else
        f32x16 s = mma_lds_reg<DH>(Ts, qf, l31, half);
#endif
#ifdef CAPDEC_ATTN_STAMPS
        asm volatile("s_nop 0" : "+v"(s));
        if (kt == 0) STAMP();  // 3: S done
#endif
        float mloc = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + row_of(r, half);
            const bool vis = key < p.Tk && (!p.causal || key <= query + off);
            s[r] = vis ? s[r] : -INFINITY;
            mloc = fmaxf(mloc, s[r]);
        }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const float m_new = fmaxf(m_run, mloc);
        const float m_use = m_new == -INFINITY ? 0.f : m_new;
        const float alpha = __expf(m_run - m_use);  // m_run = -inf -> 0
        float pr[16];
        float lsum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            pr[r] = __expf(s[r] - m_use);
            lsum += pr[r];
        }
        lsum += __shfl_xor(lsum, 32, 64);
        l_run = l_run * alpha + lsum;
        m_run = m_new;
        if (p.drop.on()) {
            const uint32_t base = ((uint32_t)bh * (uint32_t)p.Tq + (uint32_t)query) * (uint32_t)p.Tk + (uint32_t)k0;
#pragma unroll
            for (int r = 0; r < 16; ++r) pr[r] = p.drop.apply(pr[r], base + (uint32_t)row_of(r, half));
        }
        if (kt == 0) STAMP();  // 4: softmax + dropout done
        __syncthreads();  // every lane is done reading K
        if (lane < 32) line[lane] = alpha;
        tile_lstore<DH>(Ts, tr, 1.f, lane, b16);  // V tile
        if (kt == 0) STAMP();  // 5: V0 arrived and stored
        __syncthreads();
        if (kt + 1 < nkt) tile_gload<DH>(tr, kg, p.ldk, k0 + 32, p.Tk, lane, b16);  // next K under the P V MFMAs
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 t = *reinterpret_cast<const float4*>(&line[8 * g + 4 * half]);
#pragma unroll
            for (int dt = 0; dt < DH / 32; ++dt) {
                oacc[dt][4 * g + 0] *= t.x; oacc[dt][4 * g + 1] *= t.y;
                oacc[dt][4 * g + 2] *= t.z; oacc[dt][4 * g + 3] *= t.w;
            }
        }
#ifdef CAPDEC_ATTN_NOMMA
        for (int dt = 0; dt < DH / 32; ++dt) for (int r = 0; r < 16; ++r) oacc[dt][r] += pr[r] * Ts[r * LD + l31 + dt * 32];
#else
        mma_acc_lds<DH>(oacc, pr, Ts, l31, half);
#endif
#ifdef CAPDEC_ATTN_STAMPS
        asm volatile("s_nop 0" : "+v"(oacc[0]));
        if (kt == 0) STAMP();  // 6: PV0 done
#endif
    }
    STAMP();  // loop done
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
    float irow[16];
    spread_rows(line, inv, irow, lane, half);
    float* og = p.out + (size_t)b * p.Tq * p.ldo + h * DH;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int qq = q0 + row_of(r, half);
        if (qq < p.Tq) {
#pragma unroll
            for (int dt = 0; dt < DH / 32; ++dt) og[(size_t)qq * p.ldo + dt * 32 + l31] = oacc[dt][r] * irow[r];
        }
    }
    if (p.lse && lane < 32 && query < p.Tq)
        p.lse[(size_t)bh * p.Tq + query] = l_run > 0.f ? m_run + __logf(l_run) : INFINITY;
    }
#ifdef CAPDEC_ATTN_STAMPS
    {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long end = __builtin_amdgcn_s_memtime();
        // stamp area: after the [B*H*Tq] lse block, 10 x u64 per workgroup (8 stamps, end, HW_ID | XCC_ID << 32)
        unsigned long long* dbg = reinterpret_cast<unsigned long long*>(p.lse + (((size_t)p.B * p.H * p.Tq + 1) & ~(size_t)1));
        if (lane == 0) {
            for (int i = 0; i < 8; ++i) dbg[(size_t)blockIdx.x * 10 + i] = i < sti_ ? st_[i] : 0ull;
            dbg[(size_t)blockIdx.x * 10 + 8] = end;
            dbg[(size_t)blockIdx.x * 10 + 9] = (unsigned long long)__builtin_amdgcn_s_getreg(0xf804 /* HW_REG_HW_ID */) |
                                               ((unsigned long long)__builtin_amdgcn_s_getreg(0xf814 /* HW_REG_XCC_ID */) << 32);
        }
    }
#endif
}

// dQ (and delta = rowsum(dO * O)); wave owns 32 queries.
template <int DH>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) void attn_bwd_dq_kernel(AttnParams p) {
    constexpr int LD = DH + 4;
    __shared__ __attribute__((aligned(16))) float Ks[32 * LD];
    __shared__ __attribute__((aligned(16))) float Vs[32 * LD];
    const int lane = threadIdx.x, l31 = lane & 31, half = lane >> 5;
    const int nqt = (p.Tq + 31) / 32;
    // causal: the tiles of one (b, h) differ in length (query tile i walks i + 1 key tiles), so the
    // longest ones are dispatched first and the short ones fill the tail of the launch
    int bh, qt;
    attn_item(p.B * p.H, nqt, p.causal != 0, bh, qt);
    if (p.causal) qt = nqt - 1 - qt;
    const int b = bh / p.H, h = bh % p.H;
    const int q0 = qt * 32, off = p.Tk - p.Tq;
    const int query = q0 + l31;
    const float* qg = p.q + (size_t)b * p.Tq * p.ldq + h * DH;
    const bool b16 = p.kv_bf16 != 0;
    const float* kg = kv_base(p.k, (size_t)b * p.Tk * p.ldk + h * DH, b16);
    const float* vg = kv_base(p.v, (size_t)b * p.Tk * p.ldv + h * DH, b16);
    const float* og = p.o + (size_t)b * p.Tq * p.ldo + h * DH;
    const float* dog = p.dout + (size_t)b * p.Tq * p.lddo + h * DH;

    float4 qf[DH / 8], dof[DH / 8], kr[DH / 8], vr[DH / 8];
    int kmax = p.Tk;
    if (p.causal) kmax = min(p.Tk, q0 + 31 + off + 1);
    const int nkt = kmax > 0 ? (kmax + 31) / 32 : 0;
    // Q, dO and O tiles arrive in full rows and are re-read as MFMA fragments from LDS
    tile_gload<DH>(qf, qg, p.ldq, q0, p.Tq, lane);
    tile_gload<DH>(dof, dog, p.lddo, q0, p.Tq, lane);
    tile_gload<DH>(kr, og, p.ldo, q0, p.Tq, lane);  // O (kr is free until the key loop)
    tile_lstore<DH>(Ks, qf, p.scale, lane);
    tile_lstore<DH>(Vs, dof, 1.f, lane);
    __syncthreads();
    lds_frag<DH>(qf, Ks, l31, half);
    lds_frag<DH>(dof, Vs, l31, half);
    __syncthreads();
    tile_lstore<DH>(Ks, kr, 1.f, lane);
    __syncthreads();
    float dsum = 0.f;
#pragma unroll
    for (int jj = 0; jj < DH / 8; ++jj) {
        const float4 ov = *reinterpret_cast<const float4*>(&Ks[l31 * LD + 8 * jj + 4 * half]);
        dsum += (dof[jj].x * ov.x + dof[jj].y * ov.y) + (dof[jj].z * ov.z + dof[jj].w * ov.w);
    }
    dsum += __shfl_xor(dsum, 32, 64);
    const float lse = query < p.Tq ? p.lse[(size_t)bh * p.Tq + query] : INFINITY;
    if (lane < 32 && query < p.Tq) p.delta[(size_t)bh * p.Tq + query] = dsum;
    if (nkt > 0) {
        tile_gload<DH>(kr, kg, p.ldk, 0, p.Tk, lane, b16);
        tile_gload<DH>(vr, vg, p.ldv, 0, p.Tk, lane, b16);
    }

    f32x16 dqacc[DH / 32];
#pragma unroll
    for (int dt = 0; dt < DH / 32; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dqacc[dt][r] = 0.f;
    for (int kt = 0; kt < nkt; ++kt) {
        const int k0 = kt * 32;
        __syncthreads();
        tile_lstore<DH>(Ks, kr, 1.f, lane, b16);
        tile_lstore<DH>(Vs, vr, 1.f, lane, b16);
        __syncthreads();
        if (kt + 1 < nkt) {
            tile_gload<DH>(kr, kg, p.ldk, k0 + 32, p.Tk, lane, b16);
            tile_gload<DH>(vr, vg, p.ldv, k0 + 32, p.Tk, lane, b16);
        }
        f32x16 s = mma_lds_reg<DH>(Ks, qf, l31, half);    // S^T[key][query]
        f32x16 dp = mma_lds_reg<DH>(Vs, dof, l31, half);  // dP^T[key][query]
        float ds[16];
        const uint32_t base = ((uint32_t)bh * (uint32_t)p.Tq + (uint32_t)query) * (uint32_t)p.Tk + (uint32_t)k0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + row_of(r, half);
            const bool vis = key < p.Tk && (!p.causal || key <= query + off);
            const float pv = vis ? __expf(s[r] - lse) : 0.f;
            float g = dp[r];
            if (p.drop.on()) g = p.drop.apply(g, base + (uint32_t)row_of(r, half));
            ds[r] = pv * (g - dsum);
        }
        mma_acc_lds<DH>(dqacc, ds, Ks, l31, half);
    }
    float* dqg = p.dq + (size_t)b * p.Tq * p.lddq + h * DH;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int qq = q0 + row_of(r, half);
        if (qq < p.Tq) {
#pragma unroll
            for (int dt = 0; dt < DH / 32; ++dt) dqg[(size_t)qq * p.lddq + dt * 32 + l31] = dqacc[dt][r] * p.scale;
        }
    }
}

// dK, dV; wave owns 32 keys.  Must run after attn_bwd_dq_kernel (reads delta).
// The next query tile's Q / dO rows are NOT prefetched into registers here: the 64 registers that
// costs take away the second wave per SIMD at dh = 64, and the second resident wave hides the loads
// better than the prefetch did (measured at every shape: C2 52 -> 38 us, C1 and C3 3-6 % faster).
template <int DH>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) void attn_bwd_dkv_kernel(AttnParams p) {
    constexpr int LD = DH + 4;
    __shared__ __attribute__((aligned(16))) float Qs[32 * LD];
    __shared__ __attribute__((aligned(16))) float dOs[32 * LD];
    __shared__ __attribute__((aligned(16))) float lse_s[32];
    __shared__ __attribute__((aligned(16))) float del_s[32];
    const int lane = threadIdx.x, l31 = lane & 31, half = lane >> 5;
    const int nkt = (p.Tk + 31) / 32;
    int bh, kt;  // causal: key tile 0 meets every query tile, the last one few -> longest first
    attn_item(p.B * p.H, nkt, p.causal != 0, bh, kt);
    const int b = bh / p.H, h = bh % p.H;
    const int k0 = kt * 32, off = p.Tk - p.Tq;
    const int key = k0 + l31;
    const float* qg = p.q + (size_t)b * p.Tq * p.ldq + h * DH;
    const bool b16 = p.kv_bf16 != 0;
    const float* kg = kv_base(p.k, (size_t)b * p.Tk * p.ldk + h * DH, b16);
    const float* vg = kv_base(p.v, (size_t)b * p.Tk * p.ldv + h * DH, b16);
    const float* dog = p.dout + (size_t)b * p.Tq * p.lddo + h * DH;

    float4 kf[DH / 8], vf[DH / 8], qr[DH / 8], dr[DH / 8];
    // this wave's K and V tile: full rows -> LDS -> fragments (kept in registers for the whole q loop)
    tile_gload<DH>(kf, kg, p.ldk, k0, p.Tk, lane, b16);
    tile_gload<DH>(vf, vg, p.ldv, k0, p.Tk, lane, b16);
    tile_lstore<DH>(Qs, kf, 1.f, lane, b16);
    tile_lstore<DH>(dOs, vf, 1.f, lane, b16);
    __syncthreads();
    lds_frag<DH>(kf, Qs, l31, half);
    lds_frag<DH>(vf, dOs, l31, half);
    f32x16 dkacc[DH / 32], dvacc[DH / 32];
#pragma unroll
    for (int dt = 0; dt < DH / 32; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dkacc[dt][r] = dvacc[dt][r] = 0.f;
    // causal: queries q with key <= q + off  <=>  q >= key - off; first useful tile holds q = k0 - off
    int qstart = 0;
    if (p.causal) {
        const int qmin = k0 - off;
        qstart = qmin > 0 ? (qmin / 32) : 0;
    }
    const int nqt = (p.Tq + 31) / 32;
    for (int qt = qstart; qt < nqt; ++qt) {
        const int q0 = qt * 32;
        tile_gload<DH>(qr, qg, p.ldq, q0, p.Tq, lane);
        tile_gload<DH>(dr, dog, p.lddo, q0, p.Tq, lane);
        __syncthreads();
        tile_lstore<DH>(Qs, qr, p.scale, lane);
        tile_lstore<DH>(dOs, dr, 1.f, lane);
        if (lane < 32) {
            const int qq = q0 + lane;
            lse_s[lane] = qq < p.Tq ? p.lse[(size_t)bh * p.Tq + qq] : INFINITY;
            del_s[lane] = qq < p.Tq ? p.delta[(size_t)bh * p.Tq + qq] : 0.f;
        }
        __syncthreads();
        f32x16 s = mma_lds_reg<DH>(Qs, kf, l31, half);    // S[query][key]
        f32x16 dp = mma_lds_reg<DH>(dOs, vf, l31, half);  // dP[query][key]
        float pd[16], ds[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 lv = *reinterpret_cast<const float4*>(&lse_s[8 * g + 4 * half]);
            const float4 dv = *reinterpret_cast<const float4*>(&del_s[8 * g + 4 * half]);
            const float la[4] = {lv.x, lv.y, lv.z, lv.w};
            const float da[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int r = 4 * g + c;
                const int qq = q0 + row_of(r, half);
                const bool vis = key < p.Tk && qq < p.Tq && (!p.causal || key <= qq + off);
                const float pv = vis ? __expf(s[r] - la[c]) : 0.f;
                float pdr = pv, gr = dp[r];
                if (p.drop.on()) {
                    const uint32_t idx = ((uint32_t)bh * (uint32_t)p.Tq + (uint32_t)qq) * (uint32_t)p.Tk + (uint32_t)key;
                    const bool keep = p.drop.keep(idx);
                    pdr = keep ? pv * p.drop.scale : 0.f;
                    gr = keep ? gr * p.drop.scale : 0.f;
                }
                pd[r] = pdr;
                ds[r] = pv * (gr - da[c]);
            }
        }
        mma_acc_lds<DH>(dvacc, pd, dOs, l31, half);
        mma_acc_lds<DH>(dkacc, ds, Qs, l31, half);
    }
    if (p.dkv_bf16) {  // the gradients only feed matrix products that read bf16 copies: write those directly
        __bf16* dkg = reinterpret_cast<__bf16*>(p.dk) + (size_t)b * p.Tk * p.lddk + h * DH;
        __bf16* dvg = reinterpret_cast<__bf16*>(p.dv) + (size_t)b * p.Tk * p.lddv + h * DH;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kk = k0 + row_of(r, half);
            if (kk < p.Tk) {
#pragma unroll
                for (int dt = 0; dt < DH / 32; ++dt) {
                    dkg[(size_t)kk * p.lddk + dt * 32 + l31] = (__bf16)dkacc[dt][r];
                    dvg[(size_t)kk * p.lddv + dt * 32 + l31] = (__bf16)dvacc[dt][r];
                }
            }
        }
        return;
    }
    float* dkg = p.dk + (size_t)b * p.Tk * p.lddk + h * DH;
    float* dvg = p.dv + (size_t)b * p.Tk * p.lddv + h * DH;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int kk = k0 + row_of(r, half);
        if (kk < p.Tk) {
#pragma unroll
            for (int dt = 0; dt < DH / 32; ++dt) {
                dkg[(size_t)kk * p.lddk + dt * 32 + l31] = dkacc[dt][r];
                dvg[(size_t)kk * p.lddv + dt * 32 + l31] = dvacc[dt][r];
            }
        }
    }
}

// ---- Tk == 1 (the reference's own configuration: the memory is ONE projected ViT vector, ref
// models.py:104): softmax over a single key is exactly 1 whatever the score, so
//   out[b,t,h,:] = dropmask(b,h,t) * v[b,h,:],   dq = dk = 0 exactly,   dv[b,h,:] = sum_t dropmask * dout[b,t,h,:]
// (torch computes the same exact zeros: p * (g - sum(p g)) with p = 1).  Streaming kernels instead of
// 32 x 64 MFMA tiles that would be 1/64 full; same dropout element index as the general kernels.
__global__ __launch_bounds__(256) void attn_s1_fwd_kernel(AttnParams p, int dh) {
    const int c4 = p.H * dh / 4;  // float4 columns per row
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)p.B * p.Tq * c4;
    const bool live = idx < total;
    const int m = live ? (int)(idx / c4) : 0, c = live ? (int)(idx % c4) : 0;
    const int b = m / p.Tq, t = m % p.Tq, h = (c * 4) / dh;
    const int bh = b * p.H + h;
    float part = 0.f;
    if (live) {
        float4 v4 = *reinterpret_cast<const float4*>(p.v + (size_t)b * p.ldv + c * 4);
        if (p.drop.on()) {
            const float w = p.drop.keep((uint32_t)bh * (uint32_t)p.Tq + (uint32_t)t) ? p.drop.scale : 0.f;
            v4.x *= w; v4.y *= w; v4.z *= w; v4.w *= w;
        }
        *reinterpret_cast<float4*>(p.out + (size_t)m * p.ldo + c * 4) = v4;
        if (p.lse) {
            const float4 q4 = *reinterpret_cast<const float4*>(p.q + (size_t)m * p.ldq + c * 4);
            const float4 k4 = *reinterpret_cast<const float4*>(p.k + (size_t)b * p.ldk + c * 4);
            part = (q4.x * k4.x + q4.y * k4.y) + (q4.z * k4.z + q4.w * k4.w);
        }
    }
    if (p.lse) {  // lse of one key = its scaled score; dh/4 consecutive lanes hold one head
        for (int o = 1; o < dh / 4; o <<= 1) part += __shfl_xor(part, o, 64);
        if (live && (c % (dh / 4)) == 0) p.lse[(size_t)bh * p.Tq + t] = part * p.scale;
    }
}

// block = (b, 16 float4 columns) x 16 query lanes; fixed summation order (bitwise reproducible)
__global__ __launch_bounds__(256) void attn_s1_bwd_kernel(AttnParams p, int dh) {
    __shared__ float4 red[16][17];
    const int c4 = p.H * dh / 4;
    const int groups = c4 / 16;
    const int b = blockIdx.x / groups, c = (blockIdx.x % groups) * 16 + (threadIdx.x & 15);
    const int tl = threadIdx.x >> 4;
    const int bh = b * p.H + (c * 4) / dh;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = tl; t < p.Tq; t += 16) {
        const size_t m = (size_t)b * p.Tq + t;
        float4 g = *reinterpret_cast<const float4*>(p.dout + m * p.lddo + c * 4);
        if (p.drop.on()) {
            const float w = p.drop.keep((uint32_t)bh * (uint32_t)p.Tq + (uint32_t)t) ? p.drop.scale : 0.f;
            g.x *= w; g.y *= w; g.z *= w; g.w *= w;
        }
        acc.x += g.x; acc.y += g.y; acc.z += g.z; acc.w += g.w;
        if (p.dq) *reinterpret_cast<float4*>(p.dq + m * p.lddq + c * 4) = zero;
    }
    red[tl][threadIdx.x & 15] = acc;
    __syncthreads();
    if (tl == 0) {
        float4 s = red[0][threadIdx.x];
        for (int j = 1; j < 16; ++j) {
            const float4 a = red[j][threadIdx.x];
            s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
        }
        *reinterpret_cast<float4*>(p.dv + (size_t)b * p.lddv + c * 4) = s;
        *reinterpret_cast<float4*>(p.dk + (size_t)b * p.lddk + c * 4) = zero;
    }
}

static int check_common(const char* what, int B, int H, int Tq, int Tk, int dh) {
    CAPDEC_REQUIRE(B > 0 && H > 0 && Tq > 0 && Tk > 0, "%s: bad shape B=%d H=%d Tq=%d Tk=%d", what, B, H, Tq, Tk);
    CAPDEC_REQUIRE(dh == 32 || dh == 64 || dh == 128, "%s: head dim %d not in {32,64,128}", what, dh);
    CAPDEC_REQUIRE((uint64_t)B * H * Tq * (uint64_t)Tk < (1ull << 32), "%s: B*H*Tq*Tk exceeds 2^32", what);
    return 0;
}
static bool ok_mat(const void* p, int ld, int H, int dh) { return p && aligned16(p) && ld % 4 == 0 && ld >= H * dh; }

}  // namespace capdec

using namespace capdec;

// 0 = the ring kernels (attention_f32_ring.hip) wherever they cover the call, 1 = the legacy one-item-per-wave kernels
// everywhere (tests compare the two bit for bit; measurements)
static int g_attn_mode = 0;
extern "C" int capdec_attn_tune(int mode) {
    CAPDEC_REQUIRE(mode == 0 || mode == 1, "capdec_attn_tune: mode %d not in {0 = auto, 1 = legacy kernels}", mode);
    g_attn_mode = mode;
    return CAPDEC_OK;
}

static int attn_fwd_impl(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* o, int ldo,
                         void* lse, int B, int H, int Tq, int Tk, int dh, int causal, float scale, float drop_p,
                         uint64_t drop_seed, capdec_stream_t stream, int kv_bf16) {
    if (int rc = check_common("capdec_attn_fwd", B, H, Tq, Tk, dh)) return rc;
    CAPDEC_REQUIRE(!kv_bf16 || Tk > 1, "capdec_attn_fwd_kv16: needs Tk > 1");
    CAPDEC_REQUIRE(ok_mat(q, ldq, H, dh) && ok_mat(k, ldk, H, dh) && ok_mat(v, ldv, H, dh) && ok_mat(o, ldo, H, dh),
                   "capdec_attn_fwd: operands must be non-null, 16-byte aligned, ld %% 4 == 0, ld >= H*dh");
    AttnParams p;
    memset(&p, 0, sizeof(p));
    p.q = (const float*)q; p.k = (const float*)k; p.v = (const float*)v; p.out = (float*)o; p.lse = (float*)lse;
    p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo;
    p.B = B; p.H = H; p.Tq = Tq; p.Tk = Tk; p.causal = causal; p.scale = scale;
    p.kv_bf16 = kv_bf16;
    p.drop = make_dropout(drop_p, drop_seed);
    const dim3 grid(B * H * ((Tq + 31) / 32));
    hipStream_t st = (hipStream_t)stream;
    if (Tk == 1) {
        const size_t total = (size_t)B * Tq * (H * dh / 4);
        hipLaunchKernelGGL(attn_s1_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, p, dh);
        return check_launch("capdec_attn_fwd");
    }
    if (g_attn_mode == 0 && attn_ring_fwd(p, dh, st)) return check_launch("capdec_attn_fwd");
    if (dh == 32) hipLaunchKernelGGL(attn_fwd_kernel<32>, grid, dim3(64), 0, st, p);
    else if (dh == 64) hipLaunchKernelGGL(attn_fwd_kernel<64>, grid, dim3(64), 0, st, p);
    else hipLaunchKernelGGL(attn_fwd_kernel<128>, grid, dim3(64), 0, st, p);
    return check_launch("capdec_attn_fwd");
}

extern "C" int capdec_attn_fwd(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* o, int ldo,
                               void* lse, int B, int H, int Tq, int Tk, int dh, int causal, float scale, float drop_p,
                               uint64_t drop_seed, capdec_stream_t stream) {
    return attn_fwd_impl(q, ldq, k, ldk, v, ldv, o, ldo, lse, B, H, Tq, Tk, dh, causal, scale, drop_p, drop_seed, stream, 0);
}

extern "C" int capdec_attn_fwd_kv16(const void* q, int ldq, const void* k16, int ldk, const void* v16, int ldv, void* o,
                                    int ldo, void* lse, int B, int H, int Tq, int Tk, int dh, int causal, float scale,
                                    float drop_p, uint64_t drop_seed, capdec_stream_t stream) {
    return attn_fwd_impl(q, ldq, k16, ldk, v16, ldv, o, ldo, lse, B, H, Tq, Tk, dh, causal, scale, drop_p, drop_seed, stream, 1);
}

static int attn_bwd_impl(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const void* o,
                         int ldo, const void* dout, int lddo, const void* lse, void* delta, void* dq, int lddq,
                         void* dk, int lddk, void* dv, int lddv, int B, int H, int Tq, int Tk, int dh, int causal,
                         float scale, float drop_p, uint64_t drop_seed, capdec_stream_t stream, int dkv_bf16,
                         int kv_bf16 = 0) {
    if (int rc = check_common("capdec_attn_bwd", B, H, Tq, Tk, dh)) return rc;
    CAPDEC_REQUIRE(ok_mat(q, ldq, H, dh) && ok_mat(k, ldk, H, dh) && ok_mat(v, ldv, H, dh) && ok_mat(o, ldo, H, dh) &&
                       ok_mat(dout, lddo, H, dh) && (ok_mat(dq, lddq, H, dh) || (Tk == 1 && !dq)) &&
                       ok_mat(dk, lddk, H, dh) && ok_mat(dv, lddv, H, dh) && (Tk == 1 || (lse && delta)),
                   "capdec_attn_bwd: operands must be non-null, 16-byte aligned, ld %% 4 == 0, ld >= H*dh");
    CAPDEC_REQUIRE(!dkv_bf16 || (Tk > 1 && lddk % 8 == 0 && lddv % 8 == 0),
                   "capdec_attn_bwd_dkv16: needs Tk > 1 and lddk / lddv (bf16 elements) that are multiples of 8");
    AttnParams p;
    memset(&p, 0, sizeof(p));
    p.dkv_bf16 = dkv_bf16;
    p.kv_bf16 = kv_bf16;
    CAPDEC_REQUIRE(!kv_bf16 || Tk > 1, "capdec_attn_bwd_kv16: needs Tk > 1");
    p.q = (const float*)q; p.k = (const float*)k; p.v = (const float*)v; p.o = (const float*)o;
    p.dout = (const float*)dout; p.lse = (float*)lse; p.delta = (float*)delta;
    p.dq = (float*)dq; p.dk = (float*)dk; p.dv = (float*)dv;
    p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo; p.lddo = lddo; p.lddq = lddq; p.lddk = lddk; p.lddv = lddv;
    p.B = B; p.H = H; p.Tq = Tq; p.Tk = Tk; p.causal = causal; p.scale = scale;
    p.drop = make_dropout(drop_p, drop_seed);
    hipStream_t st = (hipStream_t)stream;
    if (Tk == 1) {
        hipLaunchKernelGGL(attn_s1_bwd_kernel, dim3(B * (H * dh / 64)), dim3(256), 0, st, p, dh);
        return check_launch("capdec_attn_bwd");
    }
    if (g_attn_mode == 0 && attn_ring_bwd(p, dh, st)) return check_launch("capdec_attn_bwd");
    const dim3 gq(B * H * ((Tq + 31) / 32)), gk(B * H * ((Tk + 31) / 32));
    if (dh == 32) {
        hipLaunchKernelGGL(attn_bwd_dq_kernel<32>, gq, dim3(64), 0, st, p);
        hipLaunchKernelGGL(attn_bwd_dkv_kernel<32>, gk, dim3(64), 0, st, p);
    } else if (dh == 64) {
        hipLaunchKernelGGL(attn_bwd_dq_kernel<64>, gq, dim3(64), 0, st, p);
        hipLaunchKernelGGL(attn_bwd_dkv_kernel<64>, gk, dim3(64), 0, st, p);
    } else {
        hipLaunchKernelGGL(attn_bwd_dq_kernel<128>, gq, dim3(64), 0, st, p);
        hipLaunchKernelGGL(attn_bwd_dkv_kernel<128>, gk, dim3(64), 0, st, p);
    }
    return check_launch("capdec_attn_bwd");
}

extern "C" int capdec_attn_bwd(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const void* o,
                               int ldo, const void* dout, int lddo, const void* lse, void* delta, void* dq, int lddq,
                               void* dk, int lddk, void* dv, int lddv, int B, int H, int Tq, int Tk, int dh, int causal,
                               float scale, float drop_p, uint64_t drop_seed, capdec_stream_t stream) {
    return attn_bwd_impl(q, ldq, k, ldk, v, ldv, o, ldo, dout, lddo, lse, delta, dq, lddq, dk, lddk, dv, lddv, B, H, Tq, Tk,
                         dh, causal, scale, drop_p, drop_seed, stream, 0);
}

extern "C" int capdec_attn_bwd_dkv16(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const void* o,
                                     int ldo, const void* dout, int lddo, const void* lse, void* delta, void* dq, int lddq,
                                     void* dk16, int lddk, void* dv16, int lddv, int B, int H, int Tq, int Tk, int dh,
                                     int causal, float scale, float drop_p, uint64_t drop_seed, capdec_stream_t stream) {
    return attn_bwd_impl(q, ldq, k, ldk, v, ldv, o, ldo, dout, lddo, lse, delta, dq, lddq, dk16, lddk, dv16, lddv, B, H, Tq,
                         Tk, dh, causal, scale, drop_p, drop_seed, stream, 1);
}

extern "C" int capdec_attn_bwd_kv16(const void* q, int ldq, const void* k16, int ldk, const void* v16, int ldv, const void* o,
                                    int ldo, const void* dout, int lddo, const void* lse, void* delta, void* dq, int lddq,
                                    void* dk16, int lddk, void* dv16, int lddv, int B, int H, int Tq, int Tk, int dh,
                                    int causal, float scale, float drop_p, uint64_t drop_seed, capdec_stream_t stream) {
    return attn_bwd_impl(q, ldq, k16, ldk, v16, ldv, o, ldo, dout, lddo, lse, delta, dq, lddq, dk16, lddk, dv16, lddv, B, H,
                         Tq, Tk, dh, causal, scale, drop_p, drop_seed, stream, 1, 1);
}
