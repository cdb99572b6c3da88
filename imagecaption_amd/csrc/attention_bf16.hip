// Attention on bf16 matrices (the CAPDEC_BF16S storage mode): scores + softmax + context and the flash-style
// backward on v_mfma_f32_32x32x16_bf16, with Q, K, V, O, dO, dQ, dK, dV all bf16 and the softmax statistics fp32.
// See include/capdec.h (capdec_attn_fwd_bf16 / capdec_attn_bwd_bf16).
//
// Same decomposition as attention_f32.hip -- one wave (64-thread workgroup) per (batch, head, 32-row tile), nothing
// crosses workgroups, no atomics, bitwise reproducible, the same dropout element index -- but built for streaming:
//   * the walked tiles (K and V of 32 keys; Q and dO of 32 queries in the dK/dV kernel) go global -> LDS by DMA
//     (global_load_lds_dwordx4: no staging registers, no ds_write) into a ring of NST image pairs, so a wave keeps
//     NST tiles in flight while it multiplies the current one, paced by s_waitcnt vmcnt alone (loads return in
//     order; a one-wave workgroup needs no barrier);
//   * the wave's own rows (Q / dO in forward and dQ, K / V in dK/dV) are the MFMA B operand and are read straight
//     from global in fragment form (16 bytes per lane and 16-deep k-step), once;
//   * S^T = K Q^T puts one query on a lane (its partner lane^32 holds the other 16 keys), so the softmax row
//     reduction is 15 fmax + one shuffle and the rescale of O^T is a per-lane scalar; the accumulator registers of
//     S^T, rounded to bf16, ARE the B operand of O^T += V^T P^T -- the k index of that product (the key) is simply
//     taken in the accumulator's row order, and the A operand V^T is fetched from the [key][dh] image in the same
//     order by the transposing LDS read (ds_read_b64_tr_b16), so P never moves between lanes or through LDS.
// An image is [32 rows][DH] bf16, unpadded (a DMA writes 1 KiB lane-linear); the 16-byte piece c of row r sits at
// slot c ^ swz(r), applied on the source address of the DMA and again in the reads.  One swizzle serves both read
// kinds: ds_read_b128 fragments (16 consecutive rows, one piece -> 16 distinct slots) and transposing reads (4
// consecutive rows x 4 pieces per 32 lanes -> 32 distinct 8-byte bank pairs).
// A DMA cannot mask: rows past the end of a matrix are clamped to its last row and the scores they produce are masked.
#include "common.h"
#include "attention_common.h"

namespace capdec {

typedef __bf16 bf16x8a __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4a __attribute__((ext_vector_type(4)));
typedef short s16x4a __attribute__((ext_vector_type(4)));
typedef short s16x8a __attribute__((ext_vector_type(8)));

struct Attn16Params {
    const __bf16* q; const __bf16* k; const __bf16* v; const __bf16* o; const __bf16* dout;
    __bf16* out; __bf16* dq; __bf16* dk; __bf16* dv;
    float* lse; float* delta;
    int ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv;
    int B, H, Tq, Tk, causal;
    float scale;
    Dropout drop;
};

#define CAPDEC_A16_GLDS(gptr, lptr) \
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(gptr), \
                                     (void __attribute__((address_space(3)))*)(lptr), 16, 0, 0)

template <int DH>
struct Img {
    static constexpr int ROWB = DH * 2;     // bytes per image row
    static constexpr int PPR = DH / 8;      // 16-byte pieces per row
    static constexpr int NI = DH / 16;      // DMA instructions (1 KiB each) per image
    static constexpr int BYTES = 32 * ROWB;
    static constexpr int NST = 256 / DH;    // image pairs in the ring (32 KB of LDS per wave)
    __device__ static __forceinline__ int swz(int r) {
        return DH == 128 ? (((r & 3) << 2) | ((r >> 2) & 3)) : ((((r >> 1) & 1) << 2) | ((r >> 2) & 3));
    }
};

// rows row0 .. row0 + 31 of a [T, ld] bf16 matrix (head slice at g) -> image at `img` (wave-uniform)
template <int DH>
__device__ __forceinline__ void dma_image(unsigned char* img, const __bf16* g, int ld, int row0, int T, int lane) {
    using L = Img<DH>;
#pragma unroll
    for (int j = 0; j < L::NI; ++j) {
        const int r = j * (64 / L::PPR) + lane / L::PPR;
        const int piece = (lane % L::PPR) ^ L::swz(r);
        const unsigned off = (unsigned)min(row0 + r, T - 1) * (unsigned)ld + (unsigned)(piece * 8);
        CAPDEC_A16_GLDS(g + off, img + j * 1024);
    }
}

// at most `images` DMA'd images (the youngest ones) may still be in flight
template <int NI>
__device__ __forceinline__ void wait_images(int images) {
    switch (images) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NI) : "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NI) : "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * NI) : "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * NI) : "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 * NI) : "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(7 * NI) : "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * NI <= 63 ? 8 * NI : 63) : "memory"); break;
    }
}

// the wave's own row (B operand: column = lane & 31, k = 16 ks + 8 half + 0..7), straight from global
template <int DH>
__device__ __forceinline__ void row_frags(bf16x8a* f, const __bf16* g, int ld, int row, int T, int half) {
    const __bf16* src = g + (unsigned)min(row, T - 1) * (unsigned)ld + 8 * half;
#pragma unroll
    for (int ks = 0; ks < DH / 16; ++ks) {
        bf16x8a v = *reinterpret_cast<const bf16x8a*>(src + 16 * ks);
        if (row >= T) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (__bf16)0.f;
        }
        f[ks] = v;
    }
}

// byte offset (k-step 0) of the "row on lane" fragment of an image: row = lane & 31, piece 2 ks + half
template <int DH>
__device__ __forceinline__ int frag_off(int l31, int half) {
    return l31 * Img<DH>::ROWB + ((half ^ Img<DH>::swz(l31)) << 4);
}
// acc = image(rows on lanes) x bfrag^T over the head dimension
template <int DH>
__device__ __forceinline__ f32x16 mma_img_reg(const unsigned char* img, int foff, const bf16x8a* bfrag) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < DH / 16; ++ks) {
        const bf16x8a a = *reinterpret_cast<const bf16x8a*>(img + (foff ^ (ks << 5)));
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bfrag[ks], acc, 0, 0, 0);
    }
    return acc;
}

// accT += image x frag^T (image rows on the accumulator rows) and acc += frag x image^T (image rows on the lanes) from
// ONE read of each image fragment: the A and B register layouts of this MFMA are the same
template <int DH>
__device__ __forceinline__ void mma_img_reg_both(f32x16& accT, f32x16& acc, const unsigned char* img, int foff,
                                                 const bf16x8a* frag) {
#pragma unroll
    for (int r = 0; r < 16; ++r) accT[r] = acc[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < DH / 16; ++ks) {
        const bf16x8a a = *reinterpret_cast<const bf16x8a*>(img + (foff ^ (ks << 5)));
        accT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, frag[ks], accT, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag[ks], a, acc, 0, 0, 0);
    }
}

// byte offsets of the two transposing reads that make the A fragment "column block 0 of the image, transposed":
// rows (k) in accumulator-row order, i.e. k slot s of a lane of half-wave `half` = image row 8 (s / 4) + 4 half + s % 4
// of the 16-row run.  Lane 4 q + p of 16-lane group g addresses row 8 rd + 4 half + q, 8-byte piece p of the 16
// columns (g & 1) * 16 ..; the hardware hands every lane its column's four k values.
template <int DH>
__device__ __forceinline__ void tr_offs(int lane, int* toff) {
    using L = Img<DH>;
    const int g4 = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3, half = g4 >> 1;
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
        const int kr = 8 * rd + 4 * half + q4;
        const int low = (g4 & 1) * 2 + (p4 >> 1);
        toff[rd] = kr * L::ROWB + ((low ^ L::swz(kr)) << 4) + 8 * (p4 & 1);
    }
}
// The transposing reads are issued as inline assembly: the compiler puts s_waitcnt vmcnt(0) in front of every
// ds_read_b64_tr_b16 BUILTIN while an LDS DMA is in flight (it cannot tell that the DMA targets another ring slot; the
// plain ds_read_b128 loads get no such wait), which drains the ring -- every tile then pays a full memory latency.
// The pacing is explicit anyway (wait_images), so the reads below are invisible to that pass and the lgkmcnt wait
// that their results need is issued here.
__device__ __forceinline__ unsigned lds_addr(const unsigned char* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)p;
}
__device__ __forceinline__ s16x4a lds_read_tr16(unsigned addr) {
    s16x4a v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
// out[dt] += image^T (column block dt, rows 16 m ..) x bfrag  for the two 16-row runs m of the image
template <int DH>
__device__ __forceinline__ void mma_trimg_acc(f32x16* out, const unsigned char* img, const int* toff, const bf16x8a* bfrag) {
    using L = Img<DH>;
    constexpr int NDT = DH / 32;
    const unsigned base = lds_addr(img);
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        s16x4a lo[NDT], hi[NDT];
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
            lo[dt] = lds_read_tr16(base + m * 16 * L::ROWB + (toff[0] ^ (dt << 6)));
            hi[dt] = lds_read_tr16(base + m * 16 * L::ROWB + (toff[1] ^ (dt << 6)));
        }
        if constexpr (NDT == 4)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[0]), "+v"(hi[0]), "+v"(lo[1]), "+v"(hi[1]), "+v"(lo[2]), "+v"(hi[2]),
                         "+v"(lo[3]), "+v"(hi[3]));
        else
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[0]), "+v"(hi[0]), "+v"(lo[1]), "+v"(hi[1]));
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
            const s16x8a v = {lo[dt][0], lo[dt][1], lo[dt][2], lo[dt][3], hi[dt][0], hi[dt][1], hi[dt][2], hi[dt][3]};
            out[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8a, v), bfrag[m], out[dt], 0, 0, 0);
        }
    }
}
// the same transposed fragments kept in registers (frag[dt][m]) for an image that does not change during a kernel
template <int DH>
__device__ __forceinline__ void read_trimg(bf16x8a (*frag)[2], const unsigned char* img, const int* toff) {
    using L = Img<DH>;
    constexpr int NDT = DH / 32;
    const unsigned base = lds_addr(img);
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        s16x4a lo[NDT], hi[NDT];
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
            lo[dt] = lds_read_tr16(base + m * 16 * L::ROWB + (toff[0] ^ (dt << 6)));
            hi[dt] = lds_read_tr16(base + m * 16 * L::ROWB + (toff[1] ^ (dt << 6)));
        }
        if constexpr (NDT == 4)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[0]), "+v"(hi[0]), "+v"(lo[1]), "+v"(hi[1]), "+v"(lo[2]), "+v"(hi[2]),
                         "+v"(lo[3]), "+v"(hi[3]));
        else
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[0]), "+v"(hi[0]), "+v"(lo[1]), "+v"(hi[1]));
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
            const s16x8a v = {lo[dt][0], lo[dt][1], lo[dt][2], lo[dt][3], hi[dt][0], hi[dt][1], hi[dt][2], hi[dt][3]};
            frag[dt][m] = __builtin_bit_cast(bf16x8a, v);
        }
    }
}
template <int DH>
__device__ __forceinline__ void mma_trfrag_acc(f32x16* out, const bf16x8a (*frag)[2], const bf16x8a* bfrag) {
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int dt = 0; dt < DH / 32; ++dt)
            out[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag[dt][m], bfrag[m], out[dt], 0, 0, 0);
}
// accumulator-layout values x[16] -> the two B fragments (k = accumulator row, 16 per fragment)
__device__ __forceinline__ void pack_rows(bf16x8a* f, const float* x) {
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int e = 0; e < 8; ++e) f[m][e] = (__bf16)x[8 * m + e];
}
// accumulators (rows = dh, column = this lane's matrix row) -> bf16 matrix row, 8 bytes at a time
template <int DH>
__device__ __forceinline__ void store_row(__bf16* dst, const f32x16* acc, float mul, int half) {
#pragma unroll
    for (int dt = 0; dt < DH / 32; ++dt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bf16x4a v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (__bf16)(acc[dt][4 * i + j] * mul);
            *reinterpret_cast<bf16x4a*>(dst + dt * 32 + 8 * i + 4 * half) = v;
        }
}

// The same rows through a (free) LDS image, so that global memory sees whole rows: the accumulator layout gives a
// lane 8-byte pieces of ITS row (32 rows per store instruction, 16 bytes each: a quarter of every 64-byte sector
// per request), the image gives 16 consecutive lanes one 256- or 128-byte row (the DMA's mapping, run backwards).
// One wave, LDS operations execute in order; the wait between the two phases is for the compiler's benefit.
template <int DH>
__device__ __forceinline__ void store_rows_via_lds(unsigned char* img, __bf16* g, int ld, int row0, int T, const f32x16* acc,
                                                   float mul, int lane) {
    using L = Img<DH>;
    const int l31 = lane & 31, half = lane >> 5;
#pragma unroll
    for (int dt = 0; dt < DH / 32; ++dt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bf16x4a v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (__bf16)(acc[dt][4 * i + j] * mul);
            *reinterpret_cast<bf16x4a*>(img + l31 * L::ROWB + (((4 * dt + i) ^ L::swz(l31)) << 4) + 8 * half) = v;
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (not __syncthreads(): its fence would drain vmcnt as well)
#pragma unroll
    for (int j = 0; j < L::NI; ++j) {
        const int r = j * (64 / L::PPR) + lane / L::PPR;
        const int piece = (lane % L::PPR) ^ L::swz(r);
        const uint4 v = *reinterpret_cast<const uint4*>(img + j * 1024 + lane * 16);
        if (row0 + r < T) *reinterpret_cast<uint4*>(g + (size_t)(row0 + r) * ld + piece * 8) = v;
    }
}

template <int DH>
__global__ __launch_bounds__(64) void attn16_fwd_kernel(Attn16Params p) {
    using L = Img<DH>;
    constexpr int NST = L::NST, NI = L::NI;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NST * 2 * L::BYTES];
    const int lane = threadIdx.x, l31 = lane & 31, half = lane >> 5;
    const int nqt = (p.Tq + 31) / 32;
    int bh, qt;
    attn_item(p.B * p.H, nqt, p.causal != 0, bh, qt);
    if (p.causal) qt = nqt - 1 - qt;  // longest tiles first
    const int b = bh / p.H, h = bh % p.H;
    const int q0 = qt * 32, off = p.Tk - p.Tq;
    const int query = q0 + l31;
    const __bf16* qg = p.q + (size_t)b * p.Tq * p.ldq + h * DH;
    const __bf16* kg = p.k + (size_t)b * p.Tk * p.ldk + h * DH;
    const __bf16* vg = p.v + (size_t)b * p.Tk * p.ldv + h * DH;
    int kmax = p.Tk;
    if (p.causal) kmax = min(p.Tk, q0 + 32 + off);
    const int nkt = kmax > 0 ? (kmax + 31) / 32 : 0;

    bf16x8a qf[DH / 16];
    row_frags<DH>(qf, qg, p.ldq, query, p.Tq, half);
#pragma unroll
    for (int s = 0; s < NST; ++s) {
        if (s < nkt) {
            dma_image<DH>(smem + s * 2 * L::BYTES, kg, p.ldk, s * 32, p.Tk, lane);
            dma_image<DH>(smem + s * 2 * L::BYTES + L::BYTES, vg, p.ldv, s * 32, p.Tk, lane);
        }
    }
    const int foff = frag_off<DH>(l31, half);
    int toff[2];
    tr_offs<DH>(lane, toff);

    f32x16 oacc[DH / 32];
#pragma unroll
    for (int dt = 0; dt < DH / 32; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    for (int kt = 0; kt < nkt; ++kt) {
        const int k0 = kt * 32;
        unsigned char* Ks = smem + (kt % NST) * 2 * L::BYTES;
        unsigned char* Vs = Ks + L::BYTES;
        const int ahead = min(nkt - 1 - kt, NST - 1);  // tiles requested after this one
        wait_images<NI>(1 + 2 * ahead);                // K of this tile has landed
        f32x16 s = mma_img_reg<DH>(Ks, foff, qf);      // S^T[key][query]
        float mloc = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + row_of(r, half);
            const bool vis = key < p.Tk && (!p.causal || key <= query + off);
            s[r] = vis ? s[r] * p.scale : -INFINITY;
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
        bf16x8a pb[2];
        pack_rows(pb, pr);
#pragma unroll
        for (int dt = 0; dt < DH / 32; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
        wait_images<NI>(2 * ahead);  // V of this tile has landed
        mma_trimg_acc<DH>(oacc, Vs, toff, pb);
        if (kt + NST < nkt) {  // this ring slot is free again (its reads have returned: the MFMAs above consumed them)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            dma_image<DH>(Ks, kg, p.ldk, k0 + NST * 32, p.Tk, lane);
            dma_image<DH>(Vs, vg, p.ldv, k0 + NST * 32, p.Tk, lane);
        }
    }
    if (query < p.Tq) {
        const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
        store_row<DH>(p.out + ((size_t)b * p.Tq + query) * p.ldo + h * DH, oacc, inv, half);
        if (p.lse && lane < 32) p.lse[(size_t)bh * p.Tq + query] = l_run > 0.f ? m_run + __logf(l_run) : INFINITY;
    }
}

// dQ (and delta = rowsum(dO * O)); the wave owns 32 queries and walks the key tiles.
template <int DH>
__global__ __launch_bounds__(64) void attn16_bwd_dq_kernel(Attn16Params p) {
    using L = Img<DH>;
    constexpr int NST = L::NST, NI = L::NI;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NST * 2 * L::BYTES];
    const int lane = threadIdx.x, l31 = lane & 31, half = lane >> 5;
    const int nqt = (p.Tq + 31) / 32;
    int bh, qt;
    attn_item(p.B * p.H, nqt, p.causal != 0, bh, qt);
    if (p.causal) qt = nqt - 1 - qt;
    const int b = bh / p.H, h = bh % p.H;
    const int q0 = qt * 32, off = p.Tk - p.Tq;
    const int query = q0 + l31;
    const __bf16* kg = p.k + (size_t)b * p.Tk * p.ldk + h * DH;
    const __bf16* vg = p.v + (size_t)b * p.Tk * p.ldv + h * DH;
    int kmax = p.Tk;
    if (p.causal) kmax = min(p.Tk, q0 + 32 + off);
    const int nkt = kmax > 0 ? (kmax + 31) / 32 : 0;

    bf16x8a qf[DH / 16], dof[DH / 16];
    row_frags<DH>(qf, p.q + (size_t)b * p.Tq * p.ldq + h * DH, p.ldq, query, p.Tq, half);
    row_frags<DH>(dof, p.dout + (size_t)b * p.Tq * p.lddo + h * DH, p.lddo, query, p.Tq, half);
#pragma unroll
    for (int s = 0; s < NST; ++s) {
        if (s < nkt) {
            dma_image<DH>(smem + s * 2 * L::BYTES, kg, p.ldk, s * 32, p.Tk, lane);
            dma_image<DH>(smem + s * 2 * L::BYTES + L::BYTES, vg, p.ldv, s * 32, p.Tk, lane);
        }
    }
    float dsum = 0.f;
    {
        bf16x8a of[DH / 16];
        row_frags<DH>(of, p.o + (size_t)b * p.Tq * p.ldo + h * DH, p.ldo, query, p.Tq, half);
#pragma unroll
        for (int ks = 0; ks < DH / 16; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) dsum = __fmaf_rn((float)dof[ks][e], (float)of[ks][e], dsum);
    }
    dsum += __shfl_xor(dsum, 32, 64);
    const float lse = query < p.Tq ? p.lse[(size_t)bh * p.Tq + query] : INFINITY;
    if (lane < 32 && query < p.Tq) p.delta[(size_t)bh * p.Tq + query] = dsum;
    const int foff = frag_off<DH>(l31, half);
    int toff[2];
    tr_offs<DH>(lane, toff);

    f32x16 dqacc[DH / 32];
#pragma unroll
    for (int dt = 0; dt < DH / 32; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dqacc[dt][r] = 0.f;
    for (int kt = 0; kt < nkt; ++kt) {
        const int k0 = kt * 32;
        unsigned char* Ks = smem + (kt % NST) * 2 * L::BYTES;
        unsigned char* Vs = Ks + L::BYTES;
        const int ahead = min(nkt - 1 - kt, NST - 1);
        wait_images<NI>(1 + 2 * ahead);
        f32x16 s = mma_img_reg<DH>(Ks, foff, qf);    // S^T[key][query]
        wait_images<NI>(2 * ahead);
        f32x16 dp = mma_img_reg<DH>(Vs, foff, dof);  // dP^T[key][query]
        float ds[16];
        const uint32_t base = ((uint32_t)bh * (uint32_t)p.Tq + (uint32_t)query) * (uint32_t)p.Tk + (uint32_t)k0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + row_of(r, half);
            const bool vis = key < p.Tk && (!p.causal || key <= query + off);
            const float pv = vis ? __expf(s[r] * p.scale - lse) : 0.f;
            float g = dp[r];
            if (p.drop.on()) g = p.drop.apply(g, base + (uint32_t)row_of(r, half));
            ds[r] = pv * (g - dsum);
        }
        bf16x8a db[2];
        pack_rows(db, ds);
        mma_trimg_acc<DH>(dqacc, Ks, toff, db);  // dQ^T[dh][query] += K^T dS^T
        if (kt + NST < nkt) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            dma_image<DH>(Ks, kg, p.ldk, k0 + NST * 32, p.Tk, lane);
            dma_image<DH>(Vs, vg, p.ldv, k0 + NST * 32, p.Tk, lane);
        }
    }
    if (query < p.Tq) store_row<DH>(p.dq + ((size_t)b * p.Tq + query) * p.lddq + h * DH, dqacc, p.scale, half);
}

// dK, dV; the wave owns 32 keys and walks the query tiles.  Must run after attn16_bwd_dq_kernel (reads delta).
template <int DH>
__global__ __launch_bounds__(64) void attn16_bwd_dkv_kernel(Attn16Params p) {
    using L = Img<DH>;
    constexpr int NST = L::NST, NI = L::NI;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NST * 2 * L::BYTES];
    const int lane = threadIdx.x, l31 = lane & 31, half = lane >> 5;
    const int nkt = (p.Tk + 31) / 32;
    int bh, kt;  // causal: key tile 0 meets every query tile, the last one few -> longest first
    attn_item(p.B * p.H, nkt, p.causal != 0, bh, kt);
    const int b = bh / p.H, h = bh % p.H;
    const int k0 = kt * 32, off = p.Tk - p.Tq;
    const int key = k0 + l31;
    const __bf16* qg = p.q + (size_t)b * p.Tq * p.ldq + h * DH;
    const __bf16* dog = p.dout + (size_t)b * p.Tq * p.lddo + h * DH;
    const int nqt = (p.Tq + 31) / 32;
    // causal: queries q with key <= q + off  <=>  q >= key - off; the first useful tile holds q = k0 - off
    int qstart = 0;
    if (p.causal) {
        const int qmin = k0 - off;
        qstart = qmin > 0 ? (qmin / 32) : 0;
    }
    const int nt = nqt - qstart;

    bf16x8a kf[DH / 16], vf[DH / 16];
    row_frags<DH>(kf, p.k + (size_t)b * p.Tk * p.ldk + h * DH, p.ldk, key, p.Tk, half);
    row_frags<DH>(vf, p.v + (size_t)b * p.Tk * p.ldv + h * DH, p.ldv, key, p.Tk, half);
#pragma unroll
    for (int s = 0; s < NST; ++s) {
        if (s < nt) {
            dma_image<DH>(smem + s * 2 * L::BYTES, qg, p.ldq, (qstart + s) * 32, p.Tq, lane);
            dma_image<DH>(smem + s * 2 * L::BYTES + L::BYTES, dog, p.lddo, (qstart + s) * 32, p.Tq, lane);
        }
    }
    const int foff = frag_off<DH>(l31, half);
    int toff[2];
    tr_offs<DH>(lane, toff);

    f32x16 dkacc[DH / 32], dvacc[DH / 32];
#pragma unroll
    for (int dt = 0; dt < DH / 32; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dkacc[dt][r] = dvacc[dt][r] = 0.f;
    for (int it = 0; it < nt; ++it) {
        const int q0 = (qstart + it) * 32;
        unsigned char* Qs = smem + (it % NST) * 2 * L::BYTES;
        unsigned char* dOs = Qs + L::BYTES;
        // per-query statistics in accumulator-row order (one address per half-wave: a broadcast load)
        float la[16], da[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qq = q0 + row_of(r, half);
            la[r] = qq < p.Tq ? p.lse[(size_t)bh * p.Tq + qq] : INFINITY;
            da[r] = qq < p.Tq ? p.delta[(size_t)bh * p.Tq + qq] : 0.f;
        }
        const int ahead = min(nt - 1 - it, NST - 1);
        wait_images<NI>(1 + 2 * ahead);
        f32x16 s = mma_img_reg<DH>(Qs, foff, kf);    // S[query][key]
        wait_images<NI>(2 * ahead);
        f32x16 dp = mma_img_reg<DH>(dOs, foff, vf);  // dP[query][key]
        float pd[16], ds[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qq = q0 + row_of(r, half);
            const bool vis = key < p.Tk && qq < p.Tq && (!p.causal || key <= qq + off);
            const float pv = vis ? __expf(s[r] * p.scale - la[r]) : 0.f;
            float pdr = pv, gr = dp[r];
            if (p.drop.on()) {
                const uint32_t idx = ((uint32_t)bh * (uint32_t)p.Tq + (uint32_t)qq) * (uint32_t)p.Tk + (uint32_t)key;
                const bool keep = p.drop.keep(idx);
                pdr = keep ? pv * p.drop.scale : 0.f;
                gr = keep ? gr * p.drop.scale : 0.f;
            }
            pd[r] = pdr;
            ds[r] = pv * (gr - da[r]);
        }
        bf16x8a pb[2], sb[2];
        pack_rows(pb, pd);
        pack_rows(sb, ds);
        mma_trimg_acc<DH>(dvacc, dOs, toff, pb);  // dV^T[dh][key] += dO^T Pd
        mma_trimg_acc<DH>(dkacc, Qs, toff, sb);   // dK^T[dh][key] += Q^T dS
        if (it + NST < nt) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            dma_image<DH>(Qs, qg, p.ldq, q0 + NST * 32, p.Tq, lane);
            dma_image<DH>(dOs, dog, p.lddo, q0 + NST * 32, p.Tq, lane);
        }
    }
    // every DMA has landed and been consumed: the ring is free
    __syncthreads();
    store_rows_via_lds<DH>(smem, p.dk + (size_t)b * p.Tk * p.lddk + h * DH, p.lddk, k0, p.Tk, dkacc, p.scale, lane);
    store_rows_via_lds<DH>(smem + L::BYTES, p.dv + (size_t)b * p.Tk * p.lddv + h * DH, p.lddv, k0, p.Tk, dvacc, 1.f, lane);
}

// Backward when all queries of a (b, h) fit one tile (Tq <= 32: the training shapes C2 / C3): ONE wave per (b, h) walks
// the key tiles once and produces dQ, dK and dV together -- K and V are read once (the two-kernel form reads them
// twice, and its dK/dV waves re-read Q and dO once per key tile), dK / dV of a tile are complete when the tile is
// done.  dO stays in LDS for the whole walk and Q in registers (row fragments and their transposes: dK^T += Q^T dS,
// dV^T += dO^T Pd); the scores are computed in both orientations from one read of the K / V fragments
// (query on the lanes for dQ, key on the lanes for dK / dV): 16 extra MFMAs per tile instead of moving P between lanes.
template <int DH>
__global__ __launch_bounds__(64) void attn16_bwd_fused_kernel(Attn16Params p) {
    using L = Img<DH>;
    constexpr int NST = L::NST, NI = L::NI;
    // dO image + the K / V ring: 40 KB (dh 128) / 36 KB (dh 64), four workgroups per CU.  Q passes through the last
    // ring slot's V image once, on its way into registers (row fragments and their transposes).
    __shared__ __attribute__((aligned(1024))) unsigned char smem[(1 + NST * 2) * L::BYTES];
    unsigned char* dOs = smem;
    unsigned char* ring = smem + L::BYTES;
    unsigned char* Qs = ring + (NST - 1) * 2 * L::BYTES + L::BYTES;
    const int lane = threadIdx.x, l31 = lane & 31, half = lane >> 5;
    int bh, qt;
    attn_item(p.B * p.H, 1, false, bh, qt);
    const int b = bh / p.H, h = bh % p.H;
    const int off = p.Tk - p.Tq;
    const int query = l31;
    const __bf16* kg = p.k + (size_t)b * p.Tk * p.ldk + h * DH;
    const __bf16* vg = p.v + (size_t)b * p.Tk * p.ldv + h * DH;
    int kmax = p.Tk;
    if (p.causal) kmax = min(p.Tk, 32 + off);
    const int nkt = kmax > 0 ? (kmax + 31) / 32 : 0;

    // register loads first (they are then older than every DMA and the image counts below are exact)
    bf16x8a of[DH / 16];
    row_frags<DH>(of, p.o + (size_t)b * p.Tq * p.ldo + h * DH, p.ldo, query, p.Tq, half);
    const float lse = query < p.Tq ? p.lse[(size_t)bh * p.Tq + query] : INFINITY;
    float la[16], da[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int qq = row_of(r, half);
        la[r] = qq < p.Tq ? p.lse[(size_t)bh * p.Tq + qq] : INFINITY;
    }
    dma_image<DH>(Qs, p.q + (size_t)b * p.Tq * p.ldq + h * DH, p.ldq, 0, p.Tq, lane);
    dma_image<DH>(dOs, p.dout + (size_t)b * p.Tq * p.lddo + h * DH, p.lddo, 0, p.Tq, lane);
#pragma unroll
    for (int s = 0; s < NST - 1; ++s) {
        if (s < nkt) {
            dma_image<DH>(ring + s * 2 * L::BYTES, kg, p.ldk, s * 32, p.Tk, lane);
            dma_image<DH>(ring + s * 2 * L::BYTES + L::BYTES, vg, p.ldv, s * 32, p.Tk, lane);
        }
    }
    const int foff = frag_off<DH>(l31, half);
    int toff[2];
    tr_offs<DH>(lane, toff);
    wait_images<NI>(2 * max(min(nkt, NST - 1) - 1, 0));  // Q, dO and the first key tile have landed
    bf16x8a qf[DH / 16], dof[DH / 16], qT[DH / 32][2];
#pragma unroll
    for (int ks = 0; ks < DH / 16; ++ks) {
        qf[ks] = *reinterpret_cast<const bf16x8a*>(Qs + (foff ^ (ks << 5)));
        dof[ks] = *reinterpret_cast<const bf16x8a*>(dOs + (foff ^ (ks << 5)));
    }
    read_trimg<DH>(qT, Qs, toff);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // Q is in registers: its slot joins the ring
    if (NST - 1 < nkt) {
        dma_image<DH>(ring + (NST - 1) * 2 * L::BYTES, kg, p.ldk, (NST - 1) * 32, p.Tk, lane);
        dma_image<DH>(ring + (NST - 1) * 2 * L::BYTES + L::BYTES, vg, p.ldv, (NST - 1) * 32, p.Tk, lane);
    }
    float dsum = 0.f;
#pragma unroll
    for (int ks = 0; ks < DH / 16; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) dsum = __fmaf_rn((float)dof[ks][e], (float)of[ks][e], dsum);
    dsum += __shfl_xor(dsum, 32, 64);
    if (lane < 32 && query < p.Tq) p.delta[(size_t)bh * p.Tq + query] = dsum;
#pragma unroll
    for (int r = 0; r < 16; ++r) da[r] = __shfl(dsum, row_of(r, half), 64);  // per-query value -> accumulator-row order

    f32x16 dqacc[DH / 32];
#pragma unroll
    for (int dt = 0; dt < DH / 32; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dqacc[dt][r] = 0.f;
    for (int kt = 0; kt < nkt; ++kt) {
        const int k0 = kt * 32;
        const int key = k0 + l31;
        unsigned char* Ks = ring + (kt % NST) * 2 * L::BYTES;
        unsigned char* Vs = Ks + L::BYTES;
        f32x16 sT, s, dpT, dp;  // (this tile's images have landed: prologue / end of the previous iteration)
        mma_img_reg_both<DH>(sT, s, Ks, foff, qf);     // S^T[key][query], S[query][key]
        mma_img_reg_both<DH>(dpT, dp, Vs, foff, dof);  // dP^T[key][query], dP[query][key]
        {   // query on the lanes: dQ^T[dh][query] += K^T dS^T
            float ds[16];
            const uint32_t base = ((uint32_t)bh * (uint32_t)p.Tq + (uint32_t)query) * (uint32_t)p.Tk + (uint32_t)k0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kk = k0 + row_of(r, half);
                const bool vis = kk < p.Tk && (!p.causal || kk <= query + off);
                const float pv = vis ? __expf(sT[r] * p.scale - lse) : 0.f;
                float g = dpT[r];
                if (p.drop.on()) g = p.drop.apply(g, base + (uint32_t)row_of(r, half));
                ds[r] = pv * (g - dsum);
            }
            bf16x8a db[2];
            pack_rows(db, ds);
            mma_trimg_acc<DH>(dqacc, Ks, toff, db);
        }
        {   // key on the lanes: dV^T[dh][key] = dO^T Pd, dK^T[dh][key] = Q^T dS -- complete for this tile
            float pd[16], ds[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qq = row_of(r, half);
                const bool vis = key < p.Tk && qq < p.Tq && (!p.causal || key <= qq + off);
                const float pv = vis ? __expf(s[r] * p.scale - la[r]) : 0.f;
                float pdr = pv, gr = dp[r];
                if (p.drop.on()) {
                    const uint32_t idx = ((uint32_t)bh * (uint32_t)p.Tq + (uint32_t)qq) * (uint32_t)p.Tk + (uint32_t)key;
                    const bool keep = p.drop.keep(idx);
                    pdr = keep ? pv * p.drop.scale : 0.f;
                    gr = keep ? gr * p.drop.scale : 0.f;
                }
                pd[r] = pdr;
                ds[r] = pv * (gr - da[r]);
            }
            bf16x8a pb[2], sb[2];
            pack_rows(pb, pd);
            pack_rows(sb, ds);
            f32x16 dkacc[DH / 32], dvacc[DH / 32];
#pragma unroll
            for (int dt = 0; dt < DH / 32; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) dkacc[dt][r] = dvacc[dt][r] = 0.f;
            mma_trimg_acc<DH>(dvacc, dOs, toff, pb);
            mma_trfrag_acc<DH>(dkacc, qT, sb);
            // The NEXT tile is awaited here, BEFORE this tile's stores are issued: loads and stores retire through one
            // in-order counter, so a wait placed after the stores would also wait for their acknowledgements (a
            // store round trip per tile: 211 -> 172 us was all the one-walk form gained with the wait at the top).
            // Requested so far: tiles up to kt + NST - 1; everything older than the tiles after kt + 1 must be done.
            if (kt + 1 < nkt) wait_images<NI>(2 * min(nkt - 2 - kt, NST - 2));
            // through this tile's own K / V images (every read of them is done)
            store_rows_via_lds<DH>(Ks, p.dk + (size_t)b * p.Tk * p.lddk + h * DH, p.lddk, k0, p.Tk, dkacc, p.scale, lane);
            store_rows_via_lds<DH>(Vs, p.dv + (size_t)b * p.Tk * p.lddv + h * DH, p.lddv, k0, p.Tk, dvacc, 1.f, lane);
        }
        if (kt + NST < nkt) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            dma_image<DH>(Ks, kg, p.ldk, k0 + NST * 32, p.Tk, lane);
            dma_image<DH>(Vs, vg, p.ldv, k0 + NST * 32, p.Tk, lane);
        }
    }
    // causal with fewer visible keys than Tk: the key tiles past the last visible one get zero gradients
    for (int k0 = nkt * 32; k0 < p.Tk; k0 += 32) {
        const int key = k0 + l31;
        if (key < p.Tk) {
            f32x16 z[DH / 32];
#pragma unroll
            for (int dt = 0; dt < DH / 32; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) z[dt][r] = 0.f;
            store_row<DH>(p.dk + ((size_t)b * p.Tk + key) * p.lddk + h * DH, z, 0.f, half);
            store_row<DH>(p.dv + ((size_t)b * p.Tk + key) * p.lddv + h * DH, z, 0.f, half);
        }
    }
    if (query < p.Tq) store_row<DH>(p.dq + ((size_t)b * p.Tq + query) * p.lddq + h * DH, dqacc, p.scale, half);
}

static bool ok_mat16(const void* p, int ld, int H, int dh, int rows) {
    return p && aligned16(p) && ld % 8 == 0 && ld >= H * dh && (uint64_t)rows * (uint64_t)ld < (1ull << 31);
}

}  // namespace capdec

using namespace capdec;

// CAPDEC_ATTN16_SPLIT_BWD=1 (read once): capdec_attn_bwd_bf16 takes the two-kernel form at every shape -- for the
// tests that compare the two forms and for measurements
static const bool g_attn16_split_bwd = [] {
    const char* e = getenv("CAPDEC_ATTN16_SPLIT_BWD");
    return e && e[0] == '1';
}();

static int attn16_check(const char* what, int B, int H, int Tq, int Tk, int dh) {
    CAPDEC_REQUIRE(B > 0 && H > 0 && Tq > 0 && Tk > 0, "%s: bad shape B=%d H=%d Tq=%d Tk=%d", what, B, H, Tq, Tk);
    CAPDEC_REQUIRE(dh == 64 || dh == 128, "%s: head dim %d not in {64,128}", what, dh);
    CAPDEC_REQUIRE((uint64_t)B * H * Tq * (uint64_t)Tk < (1ull << 32), "%s: B*H*Tq*Tk exceeds 2^32", what);
    return 0;
}

extern "C" int capdec_attn_fwd_bf16(const void* q16, int ldq, const void* k16, int ldk, const void* v16, int ldv, void* o16,
                                    int ldo, void* lse, int B, int H, int Tq, int Tk, int dh, int causal, float scale,
                                    float drop_p, uint64_t drop_seed, capdec_stream_t stream) {
    if (int rc = attn16_check("capdec_attn_fwd_bf16", B, H, Tq, Tk, dh)) return rc;
    CAPDEC_REQUIRE(ok_mat16(q16, ldq, H, dh, B * Tq) && ok_mat16(k16, ldk, H, dh, B * Tk) && ok_mat16(v16, ldv, H, dh, B * Tk) &&
                       ok_mat16(o16, ldo, H, dh, B * Tq),
                   "capdec_attn_fwd_bf16: operands must be non-null, 16-byte aligned bf16 matrices with ld %% 8 == 0, "
                   "ld >= H*dh and fewer than 2^31 elements");
    Attn16Params p;
    memset(&p, 0, sizeof(p));
    p.q = (const __bf16*)q16; p.k = (const __bf16*)k16; p.v = (const __bf16*)v16; p.out = (__bf16*)o16; p.lse = (float*)lse;
    p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo;
    p.B = B; p.H = H; p.Tq = Tq; p.Tk = Tk; p.causal = causal; p.scale = scale;
    p.drop = make_dropout(drop_p, drop_seed);
    const dim3 grid(B * H * ((Tq + 31) / 32));
    hipStream_t st = (hipStream_t)stream;
    if (dh == 64) hipLaunchKernelGGL(attn16_fwd_kernel<64>, grid, dim3(64), 0, st, p);
    else hipLaunchKernelGGL(attn16_fwd_kernel<128>, grid, dim3(64), 0, st, p);
    return check_launch("capdec_attn_fwd_bf16");
}

extern "C" int capdec_attn_bwd_bf16(const void* q16, int ldq, const void* k16, int ldk, const void* v16, int ldv,
                                    const void* o16, int ldo, const void* do16, int lddo, const void* lse, void* delta,
                                    void* dq16, int lddq, void* dk16, int lddk, void* dv16, int lddv, int B, int H, int Tq,
                                    int Tk, int dh, int causal, float scale, float drop_p, uint64_t drop_seed,
                                    capdec_stream_t stream) {
    if (int rc = attn16_check("capdec_attn_bwd_bf16", B, H, Tq, Tk, dh)) return rc;
    CAPDEC_REQUIRE(ok_mat16(q16, ldq, H, dh, B * Tq) && ok_mat16(k16, ldk, H, dh, B * Tk) && ok_mat16(v16, ldv, H, dh, B * Tk) &&
                       ok_mat16(o16, ldo, H, dh, B * Tq) && ok_mat16(do16, lddo, H, dh, B * Tq) &&
                       ok_mat16(dq16, lddq, H, dh, B * Tq) && ok_mat16(dk16, lddk, H, dh, B * Tk) &&
                       ok_mat16(dv16, lddv, H, dh, B * Tk) && lse && delta,
                   "capdec_attn_bwd_bf16: operands must be non-null, 16-byte aligned bf16 matrices with ld %% 8 == 0, "
                   "ld >= H*dh and fewer than 2^31 elements; lse / delta non-null");
    Attn16Params p;
    memset(&p, 0, sizeof(p));
    p.q = (const __bf16*)q16; p.k = (const __bf16*)k16; p.v = (const __bf16*)v16; p.o = (const __bf16*)o16;
    p.dout = (const __bf16*)do16; p.lse = (float*)lse; p.delta = (float*)delta;
    p.dq = (__bf16*)dq16; p.dk = (__bf16*)dk16; p.dv = (__bf16*)dv16;
    p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo; p.lddo = lddo; p.lddq = lddq; p.lddk = lddk; p.lddv = lddv;
    p.B = B; p.H = H; p.Tq = Tq; p.Tk = Tk; p.causal = causal; p.scale = scale;
    p.drop = make_dropout(drop_p, drop_seed);
    hipStream_t st = (hipStream_t)stream;
    const dim3 gq(B * H * ((Tq + 31) / 32)), gk(B * H * ((Tk + 31) / 32));
    if (Tq <= 32 && !g_attn16_split_bwd) {  // one query tile: dQ, dK and dV in one walk over the keys
        if (dh == 64) hipLaunchKernelGGL(attn16_bwd_fused_kernel<64>, dim3(B * H), dim3(64), 0, st, p);
        else hipLaunchKernelGGL(attn16_bwd_fused_kernel<128>, dim3(B * H), dim3(64), 0, st, p);
        return check_launch("capdec_attn_bwd_bf16");
    }
    if (dh == 64) {
        hipLaunchKernelGGL(attn16_bwd_dq_kernel<64>, gq, dim3(64), 0, st, p);
        hipLaunchKernelGGL(attn16_bwd_dkv_kernel<64>, gk, dim3(64), 0, st, p);
    } else {
        hipLaunchKernelGGL(attn16_bwd_dq_kernel<128>, gq, dim3(64), 0, st, p);
        hipLaunchKernelGGL(attn16_bwd_dkv_kernel<128>, gk, dim3(64), 0, st, p);
    }
    return check_launch("capdec_attn_bwd_bf16");
}
