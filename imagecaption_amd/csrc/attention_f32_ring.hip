// fp32 attention, streaming form: the kernels of attention_f32.hip rebuilt on the structure that made the bf16
// kernels (attention_bf16.hip) fast -- tiles move global -> LDS by DMA into a ring, several tiles in flight per wave,
// paced by counted s_waitcnt vmcnt alone -- and made PERSISTENT: one-wave workgroups (one per SIMD) walk a sequence of
// (batch, head, query tile) items and keep requesting the NEXT item's tiles while they multiply the current one.
//
// Why (profiles/r02_attention_evidence.txt): the one-item-per-wave kernels run as ONE round of waves that all load,
// then all multiply, then all store -- the memory system idles while the matrix pipes work and the other way round
// (fp32 forward at C2: 24 us for 72 MB, of which 6 us is every wave waiting for its first tile).  Here a wave's ring
// always holds the next tiles, across item boundaries, so memory and MFMA time overlap.
//
// Same arithmetic as attention_f32.hip, in the same order (exact fp32 v_mfma_f32_32x32x2_f32, S^T = K Q^T with one
// query per lane, online softmax, the accumulator registers fed back as the next product's operand, the same dropout
// element index): results are BIT-IDENTICAL to the legacy kernels, which stay for head dim 32, one-key memories and the
// multi-tile backward, and as the reference the tests compare against (capdec_attn_tune).
//
// Image = 32 rows x DH floats, unpadded (a DMA instruction writes 1 KiB lane-linear); the 16-byte piece c of row r
// sits at slot c ^ (r & 15), applied on the SOURCE address of the DMA and again in the reads: the ds_read_b128
// fragment reads (16 rows, one piece) and the ds_read_b32 row reads (one row, 32 consecutive floats) are both
// bank-conflict free.  Only DMA instructions that hold a valid row are issued (a partial tile of 17 rows costs 5
// instructions, not 8); rows past the end inside an issued instruction are clamped to the last row, rows of skipped
// instructions hold stale data -- every operand read from such a row is replaced by zero / masked before use.
//
// Pacing: vector-memory operations of a wave (DMA, stores) retire in issue order through one counter.  The wave counts
// every operation it issues (`issued`) and remembers the count at the end of each image's request (`mark[slot]`):
// image n has landed once at most issued - mark[slot(n)] younger operations are outstanding.  The output stores are
// buffer stores whose out-of-range rows are dropped by the descriptor (so their NUMBER is fixed) and are counted too:
// a wave never waits for a store to be acknowledged.
#include "common.h"
#include "attention_common.h"

namespace capdec {

// Diagnostic builds only (CAPDEC_EXTRA_FLAGS=-DCAPDEC_RING_STAMPS, tools/attn_ring_stamps.py; never in the shipped library):
// per workgroup, shader cycles spent waiting for images, requesting images, and in total.
#ifdef CAPDEC_RING_STAMPS
__device__ unsigned long long g_ring_dbg[4096 * 4];  // fwd: 4 words per workgroup; bwd: 8 words per workgroup
#define RING_T0(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#define RING_ACC(acc, v) acc += __builtin_amdgcn_s_memtime() - v
#else
#define RING_T0(v) do { } while (0)
#define RING_ACC(acc, v) do { } while (0)
#endif

#define CAPDEC_RING_GLDS(gptr, lptr) \
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(gptr), \
                                     (void __attribute__((address_space(3)))*)(lptr), 16, 0, 0)

template <int DH>
struct RImg {
    static constexpr int ROWB = DH * 4;      // bytes per image row
    static constexpr int PPR = DH / 4;       // 16-byte pieces per row
    static constexpr int RPI = 64 / PPR;     // rows per DMA instruction
    static constexpr int NI = 32 / RPI;      // DMA instructions per full image
    static constexpr int BYTES = 32 * ROWB;
};

// wait until at most `allowed` (wave-uniform, >= 0) of this wave's youngest vector-memory operations are outstanding
__device__ __forceinline__ void vm_wait(int allowed) {
    allowed = __builtin_amdgcn_readfirstlane(allowed);
#define CAPDEC_VMW(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
    switch (allowed > 63 ? 63 : allowed) {
        CAPDEC_VMW(0) CAPDEC_VMW(1) CAPDEC_VMW(2) CAPDEC_VMW(3) CAPDEC_VMW(4) CAPDEC_VMW(5) CAPDEC_VMW(6) CAPDEC_VMW(7)
        CAPDEC_VMW(8) CAPDEC_VMW(9) CAPDEC_VMW(10) CAPDEC_VMW(11) CAPDEC_VMW(12) CAPDEC_VMW(13) CAPDEC_VMW(14) CAPDEC_VMW(15)
        CAPDEC_VMW(16) CAPDEC_VMW(17) CAPDEC_VMW(18) CAPDEC_VMW(19) CAPDEC_VMW(20) CAPDEC_VMW(21) CAPDEC_VMW(22) CAPDEC_VMW(23)
        CAPDEC_VMW(24) CAPDEC_VMW(25) CAPDEC_VMW(26) CAPDEC_VMW(27) CAPDEC_VMW(28) CAPDEC_VMW(29) CAPDEC_VMW(30) CAPDEC_VMW(31)
        CAPDEC_VMW(32) CAPDEC_VMW(33) CAPDEC_VMW(34) CAPDEC_VMW(35) CAPDEC_VMW(36) CAPDEC_VMW(37) CAPDEC_VMW(38) CAPDEC_VMW(39)
        CAPDEC_VMW(40) CAPDEC_VMW(41) CAPDEC_VMW(42) CAPDEC_VMW(43) CAPDEC_VMW(44) CAPDEC_VMW(45) CAPDEC_VMW(46) CAPDEC_VMW(47)
        CAPDEC_VMW(48) CAPDEC_VMW(49) CAPDEC_VMW(50) CAPDEC_VMW(51) CAPDEC_VMW(52) CAPDEC_VMW(53) CAPDEC_VMW(54) CAPDEC_VMW(55)
        CAPDEC_VMW(56) CAPDEC_VMW(57) CAPDEC_VMW(58) CAPDEC_VMW(59) CAPDEC_VMW(60) CAPDEC_VMW(61) CAPDEC_VMW(62)
        default: asm volatile("s_waitcnt vmcnt(63)" ::: "memory"); break;
    }
#undef CAPDEC_VMW
}

// rows row0 .. row0 + 31 of a [T, ld] fp32 matrix (head slice at g) -> image at `img` (wave-uniform); returns the
// number of DMA instructions issued (those that hold at least one valid row)
template <int DH>
__device__ __forceinline__ int dma_rows(unsigned char* img, const float* g, int ld, int row0, int T, int lane) {
    using L = RImg<DH>;
    const int valid = T - row0;  // >= 1 by construction
    int n = 0;
#pragma unroll
    for (int j = 0; j < L::NI; ++j) {
        if (j * L::RPI < valid) {
            const int r = j * L::RPI + lane / L::PPR;
            const int piece = (lane % L::PPR) ^ (r & 15);
            const unsigned off = (unsigned)min(row0 + r, T - 1) * (unsigned)ld + (unsigned)(piece * 4);
            CAPDEC_RING_GLDS(g + off, img + j * 1024);
            ++n;
        }
    }
    return n;
}

// ---- LDS access by inline assembly ------------------------------------------------------------------------------------
// Every LDS read / write of these kernels is issued as inline assembly: while an LDS DMA is in flight the compiler puts
// s_waitcnt vmcnt(0) in front of ordinary LDS accesses (it cannot tell that the DMA targets another ring slot), which
// drains the ring -- every tile would pay a full memory latency.  The pacing is explicit anyway (vm_wait), so these
// accesses are invisible to that pass; the lgkmcnt wait their results need is issued here too, tied to the result
// registers so that no use can be scheduled ahead of it.
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)p;
}
__device__ __forceinline__ f32x4 lds_read128(unsigned addr) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
template <int OFF>
__device__ __forceinline__ float lds_read32(unsigned addr) {
    float v;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
__device__ __forceinline__ void lds_write32(unsigned addr, float v) { asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void tie4(f32x4* x) { asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3])); }
__device__ __forceinline__ void tie16(float* x) {
    asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]),
                 "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]));
}
__device__ __forceinline__ void lgkm_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// byte offset (k-step 0) of the "row on lane" fragment: row = lane & 31, 16-byte piece 2 jj + half
template <int DH>
__device__ __forceinline__ int rfrag_off(int l31, int half) {
    return l31 * RImg<DH>::ROWB + ((half ^ (l31 & 15)) << 4);
}
// issue the fragment reads of an image (LDS address `img`, 1 KiB aligned); results valid after frags_arrived
template <int DH>
__device__ __forceinline__ void read_frags(f32x4* f, unsigned img, int foff) {
#pragma unroll
    for (int jj = 0; jj < DH / 8; ++jj) f[jj] = lds_read128((img + foff) ^ (jj << 5));
}
template <int DH>
__device__ __forceinline__ void frags_arrived(f32x4* f) {
#pragma unroll
    for (int i = 0; i < DH / 32; ++i) tie4(f + 4 * i);
}
// acc = A x B^T over the head dimension from two fragment sets (the k pairing of attention_f32.hip's mma_lds_reg)
template <int DH>
__device__ __forceinline__ f32x16 mma_frags(const f32x4* a, const f32x4* b) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int jj = 0; jj < DH / 8; ++jj) {
#pragma unroll
        for (int c = 0; c < 4; ++c) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[jj][c], b[jj][c], acc, 0, 0, 0);
    }
    return acc;
}
// per-lane part of the address of element (row, 32 dt + l31) of an image; the row part is added by read_rows
template <int DH>
__device__ __forceinline__ int rowread_base(int l31, int half) {
    return (((l31 >> 2) << 4) ^ (half << 6)) + (l31 & 3) * 4 + 4 * half * RImg<DH>::ROWB;
}
// issue x[r * (DH/32) + dt] = image[row_of(r, half)][32 dt + l31]; after rows_arrived, entries of rows >= nvalid are zero
// (stale LDS, see the header)
template <int DH, int R>
__device__ __forceinline__ void read_rows_r(float* x, unsigned img_rbase) {
    if constexpr (R < 16) {
        constexpr int rc = (R & 3) + 8 * (R >> 2);                     // row without the half-wave term
        constexpr int cr = ((R & 3) << 4) ^ (((R >> 2) & 1) << 7);     // swizzle of that row: (row & 15) << 4
#pragma unroll
        for (int dt = 0; dt < DH / 32; ++dt)
            x[R * (DH / 32) + dt] = lds_read32<rc * RImg<DH>::ROWB>(img_rbase ^ (cr ^ (dt << 7)));
        read_rows_r<DH, R + 1>(x, img_rbase);
    }
}
template <int DH>
__device__ __forceinline__ void read_rows(float* x, unsigned img, int rbase) { read_rows_r<DH, 0>(x, img + rbase); }
template <int DH>
__device__ __forceinline__ void rows_arrived(float* x, int half, int nvalid) {
#pragma unroll
    for (int i = 0; i < DH / 32; ++i) tie16(x + 16 * i);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const bool ok = (r & 3) + 8 * (r >> 2) + 4 * half < nvalid;
#pragma unroll
        for (int dt = 0; dt < DH / 32; ++dt) x[r * (DH / 32) + dt] = ok ? x[r * (DH / 32) + dt] : 0.f;
    }
}
// a per-query scalar (lanes 0..31, row = lane) -> accumulator-row order, through a 32-float line (one wave: LDS
// operations execute in order, no barrier -- a __syncthreads() would drain the DMA ring)
__device__ __forceinline__ void line_put(unsigned line, float v, int lane) {
    if (lane < 32) lds_write32(line + 4 * lane, v);
}
__device__ __forceinline__ void line_get(f32x4* a, unsigned line, int half) {
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) a[g4] = lds_read128(line + 4 * (8 * g4 + 4 * half));
}


// dsT[r] = T[key = row_of(r, half)][query = lane & 31] of the [32][36]-float dS tile (t_rd: per-lane base address)
template <int R>
__device__ __forceinline__ void ts_read_r(float* x, unsigned t_rd) {
    if constexpr (R < 16) {
        x[R] = lds_read32<((R & 3) + 8 * (R >> 2)) * 144>(t_rd);
        ts_read_r<R + 1>(x, t_rd);
    }
}
__device__ __forceinline__ void ts_read(float* x, unsigned t_rd) { ts_read_r<0>(x, t_rd); }

// small uniform arrays indexed by a uniform, run-time slot number: unrolled selects keep them in scalar registers
template <int NS>
__device__ __forceinline__ int sel_get(const int (&a)[NS], int i) {
    int v = a[0];
#pragma unroll
    for (int k = 1; k < NS; ++k) v = i == k ? a[k] : v;
    return v;
}
template <int NS>
__device__ __forceinline__ void sel_set(int (&a)[NS], int i, int v) {
#pragma unroll
    for (int k = 0; k < NS; ++k) a[k] = i == k ? v : a[k];
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t out_rsrc(const void* base) {
    // raw buffer: byte offsets below num_records are stored, anything else is dropped by the hardware
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, base ? 0x7fffffff : 0, 0x00020000);
}
constexpr unsigned OOB = 0xfffffff0u;
constexpr unsigned OOB_S = 0x80000000u;  // with a scalar offset below 2^30 added: still past num_records, no 32-bit wrap

// ---- the stream of images a wave walks in the forward kernel: per item  Q, K0, V0, K1, V1, ... ----------------------
template <int DH, int NS>
__global__ __launch_bounds__(64) void attn_fwd_ring_kernel(AttnParams p, int n_items) {
    using L = RImg<DH>;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NS * L::BYTES];
    __shared__ __attribute__((aligned(16))) float line[32];
    const int lane = threadIdx.x, l31 = lane & 31, half = lane >> 5;
    const int nqt = (p.Tq + 31) / 32, n_bh = p.B * p.H, off = p.Tk - p.Tq;
    const int G = gridDim.x;
    const bool causal = p.causal != 0;

    auto tiles_of = [&](int q0) {
        int kmax = p.Tk;
        if (causal) kmax = min(p.Tk, q0 + 31 + off + 1);
        return kmax > 0 ? (kmax + 31) / 32 : 0;
    };
    auto decode = [&](int item, int& bh, int& q0) {
        int qt;
        attn_item_at(item, n_bh, nqt, causal, bh, qt);
        if (causal) qt = nqt - 1 - qt;  // longest tiles first
        q0 = qt * 32;
    };

    // ---- requester: which image comes next, and the accounting of what has been issued
    int issued = 0, reqs = 0, mark[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) mark[k] = 0;
    int r_item = blockIdx.x, r_pos = 0, r_nimg = 0, r_q0 = 0;
    const float *r_q = nullptr, *r_k = nullptr, *r_v = nullptr;
    auto r_decode = [&]() {
        if (r_item < n_items) {
            int bh;
            decode(r_item, bh, r_q0);
            const int b = bh / p.H, h = bh % p.H;
            r_q = p.q + (size_t)b * p.Tq * p.ldq + h * DH;
            r_k = p.k + (size_t)b * p.Tk * p.ldk + h * DH;
            r_v = p.v + (size_t)b * p.Tk * p.ldv + h * DH;
            r_nimg = 1 + 2 * tiles_of(r_q0);
            r_pos = 0;
        }
    };
#ifdef CAPDEC_RING_STAMPS
    unsigned long long t_wait = 0, t_req = 0, t_lgkm = 0;
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
#endif
    auto request = [&]() {
        if (r_item >= n_items) return;
        RING_T0(tr0);
        const int slot = reqs % NS;
        unsigned char* img = smem + slot * L::BYTES;
        int n;
        if (r_pos == 0) {
            n = dma_rows<DH>(img, r_q, p.ldq, r_q0, p.Tq, lane);
        } else {
            const int t = (r_pos - 1) >> 1;
            if ((r_pos - 1) & 1) n = dma_rows<DH>(img, r_v, p.ldv, t * 32, p.Tk, lane);
            else n = dma_rows<DH>(img, r_k, p.ldk, t * 32, p.Tk, lane);
        }
        issued += n;
        sel_set<NS>(mark, slot, issued);
        ++reqs;
        if (++r_pos == r_nimg) {
            r_item += G;
            r_decode();
        }
        RING_ACC(t_req, tr0);
    };
    auto wait_image = [&](int slot) {
        RING_T0(tw0);
        vm_wait(issued - sel_get<NS>(mark, slot));
        RING_ACC(t_wait, tw0);
    };
    // the reads of the image just consumed have returned: its slot takes the next request
    auto release_and_request = [&]() {
        RING_T0(tl0);
        lgkm_wait();
        RING_ACC(t_lgkm, tl0);
        request();
        asm volatile("" ::: "memory");
    };
    r_decode();
#pragma unroll
    for (int s = 0; s < NS; ++s) request();

    const int foff = rfrag_off<DH>(l31, half), rbase = rowread_base<DH>(l31, half);
    const unsigned smem_a = lds_addr(smem), line_a = lds_addr(line);
    const __amdgpu_buffer_rsrc_t lse_rs = out_rsrc(p.lse);
    int cons = 0;
    for (int item = blockIdx.x; item < n_items; item += G) {
        int bh, q0;
        decode(item, bh, q0);
        const int b = bh / p.H, h = bh % p.H;
        const int query = q0 + l31;
        const int nkt = tiles_of(q0);

        f32x4 qf[DH / 8];
        {
            const int slot = cons % NS;
            wait_image(slot);
            read_frags<DH>(qf, smem_a + slot * L::BYTES, foff);
            ++cons;
            release_and_request();
            frags_arrived<DH>(qf);
#pragma unroll
            for (int jj = 0; jj < DH / 8; ++jj) qf[jj] *= p.scale;
        }
        f32x16 oacc[DH / 32];
#pragma unroll
        for (int dt = 0; dt < DH / 32; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
        float m_run = -INFINITY, l_run = 0.f;
        for (int kt = 0; kt < nkt; ++kt) {
            const int k0 = kt * 32;
            f32x16 s;
            {
                f32x4 kf[DH / 8];
                const int slot = cons % NS;
                wait_image(slot);
                read_frags<DH>(kf, smem_a + slot * L::BYTES, foff);
                ++cons;
                release_and_request();
                frags_arrived<DH>(kf);
                s = mma_frags<DH>(kf, qf);  // S^T[key][query]
            }
            float mloc = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = k0 + row_of(r, half);
                const bool vis = key < p.Tk && (!causal || key <= query + off);
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
            line_put(line_a, alpha, lane);  // per-query rescale factors -> accumulator-row order
            float vv[16 * (DH / 32)];
            f32x4 al[4];
            {
                const int slot = cons % NS;
                wait_image(slot);
                read_rows<DH>(vv, smem_a + slot * L::BYTES, rbase);
                line_get(al, line_a, half);
                ++cons;
                release_and_request();
                rows_arrived<DH>(vv, half, p.Tk - k0);
                tie4(al);
            }
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
                for (int dt = 0; dt < DH / 32; ++dt)
#pragma unroll
                    for (int c = 0; c < 4; ++c) oacc[dt][4 * g4 + c] *= al[g4][c];
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int dt = 0; dt < DH / 32; ++dt)
                    oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(pr[r], vv[r * (DH / 32) + dt], oacc[dt], 0, 0, 0);
        }
        // ---- epilogue: O rows (two 128-byte row segments per store instruction), lse
        const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
        line_put(line_a, inv, lane);
        f32x4 iv[4];
        line_get(iv, line_a, half);
        lgkm_wait();
        tie4(iv);
        float irow[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) irow[r] = iv[r >> 2][r & 3];
        const __amdgpu_buffer_rsrc_t o_rs = out_rsrc(p.out + (size_t)b * p.Tq * p.ldo + h * DH);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qq = q0 + row_of(r, half);
#pragma unroll
            for (int dt = 0; dt < DH / 32; ++dt) {
                const unsigned vo = qq < p.Tq ? ((unsigned)qq * (unsigned)p.ldo + (unsigned)(dt * 32 + l31)) * 4u : OOB;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(oacc[dt][r] * irow[r]), o_rs, vo, 0, 0);
            }
        }
        {
            const unsigned vo = (lane < 32 && query < p.Tq) ? ((unsigned)bh * (unsigned)p.Tq + (unsigned)query) * 4u : OOB;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(l_run > 0.f ? m_run + __logf(l_run) : INFINITY), lse_rs, vo, 0, 0);
        }
        issued += 16 * (DH / 32) + 1;
        asm volatile("" ::: "memory");
    }
#ifdef CAPDEC_RING_STAMPS
    if (lane == 0 && blockIdx.x < 4096) {
        g_ring_dbg[blockIdx.x * 4 + 0] = __builtin_amdgcn_s_memtime() - t_begin;
        g_ring_dbg[blockIdx.x * 4 + 1] = t_wait;
        g_ring_dbg[blockIdx.x * 4 + 2] = t_req;
        g_ring_dbg[blockIdx.x * 4 + 3] = t_lgkm;
    }
#endif
}


// ---- backward, one query tile (Tq <= 32: the training shapes C2 / C3) ---------------------------------------------------
// ONE wave per (b, h) walks the key tiles once and produces dQ, dK and dV together: K and V are read once (the
// two-kernel form reads them twice and its dK / dV waves re-read Q and dO once per key tile), and the scores and dP are
// computed ONCE per tile -- S[query][key] with the key on the lanes, which is the operand layout of dV += Pd^T dO and
// dK += dS^T Q as it stands; dQ += dS K needs the query on the lanes, and gets it by passing dS through a 4.5 KB LDS
// tile (4 ds_write_b128 + 16 ds_read_b32 per lane) instead of 64 more MFMAs (the two-kernel form computes S and dP in
// both orientations: 224 MFMAs per key tile, here 160).  Same values, products and summation orders as
// attn_bwd_dq_kernel / attn_bwd_dkv_kernel: bit-identical results.
// LDS: K and V images of the current tile (the V image is dead once its fragments are in registers and then holds the
// dS tile), the Q (pre-scaled in place) and dO images of QR = Tq rounded up to 4 rows, two 32-float lines.
template <int DH>
__global__ __launch_bounds__(64) void attn_bwd_ring_kernel(AttnParams p, int qr) {
    using L = RImg<DH>;
    extern __shared__ __attribute__((aligned(1024))) unsigned char dsm[];
    unsigned char* Ks = dsm;
    unsigned char* Vs = dsm + L::BYTES;
    unsigned char* Qs = dsm + 2 * L::BYTES;
    unsigned char* dOs = Qs + qr * L::ROWB;
    const unsigned ks_a = lds_addr(Ks), vs_a = lds_addr(Vs), qs_a = lds_addr(Qs), dos_a = lds_addr(dOs);
    const unsigned line_a = dos_a + qr * L::ROWB, lse_line = line_a + 128;  // delta line, lse line
    const int lane = threadIdx.x, l31 = lane & 31, half = lane >> 5;
    int bh, qt;
    attn_item(p.B * p.H, 1, false, bh, qt);
    const int b = bh / p.H, h = bh % p.H;
    const int off = p.Tk - p.Tq;
    const bool causal = p.causal != 0;
    const float* kg = p.k + (size_t)b * p.Tk * p.ldk + h * DH;
    const float* vg = p.v + (size_t)b * p.Tk * p.ldv + h * DH;
    int kmax = p.Tk;
    if (causal) kmax = min(p.Tk, 31 + off + 1);
    const int nkt = kmax > 0 ? (kmax + 31) / 32 : 0;

    // ---- plain loads first (older than every DMA): O fragments of this lane's query row, lse in accumulator-row order
    f32x4 of[DH / 8];
    {
        const float* orow = p.o + ((size_t)b * p.Tq + min(l31, p.Tq - 1)) * p.ldo + h * DH + 4 * half;
#pragma unroll
        for (int jj = 0; jj < DH / 8; ++jj) of[jj] = *reinterpret_cast<const f32x4*>(orow + 8 * jj);
    }
    // lse of this lane's query (lanes 0..31); it reaches the accumulator-row order through an LDS line, per tile
    const float lse_q = p.lse[(size_t)bh * p.Tq + min(l31, p.Tq - 1)];
    asm volatile("" ::: "memory");
    int issued = 0;
    const int n_q = dma_rows<DH>(Qs, p.q + (size_t)b * p.Tq * p.ldq + h * DH, p.ldq, 0, p.Tq, lane);
    const int n_do = dma_rows<DH>(dOs, p.dout + (size_t)b * p.Tq * p.lddo + h * DH, p.lddo, 0, p.Tq, lane);
    issued = n_q + n_do;
    const int mark_qdo = issued;
    int mark_k = 0, mark_v = 0;
    if (nkt > 0) {
        issued += dma_rows<DH>(Ks, kg, p.ldk, 0, p.Tk, lane);
        mark_k = issued;
        issued += dma_rows<DH>(Vs, vg, p.ldv, 0, p.Tk, lane);
        mark_v = issued;
    }
    asm volatile("" ::: "memory");

    const int foff = rfrag_off<DH>(l31, half), rbase = rowread_base<DH>(l31, half);
#ifdef CAPDEC_RING_STAMPS
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
    unsigned long long t_last = t_begin;
#define PHASE(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); ph[i] += n_ - t_last; t_last = n_; } while (0)
#else
#define PHASE(i) do { } while (0)
#endif
    // ---- Q (scaled in place, as the legacy kernels store it) and dO fragments, delta
    vm_wait(issued - mark_qdo);
    PHASE(0);
    {
        // every lane scales the pieces of the DMA instructions that were issued: instruction j wrote 1 KiB lane-linear
        f32x4 t[L::NI];
#pragma unroll
        for (int j = 0; j < L::NI; ++j) t[j] = lds_read128(qs_a + j * 1024 + lane * 16);  // (past the image: harmless)
        lgkm_wait();
#pragma unroll
        for (int j = 0; j < L::NI / 4; ++j) tie4(t + 4 * j);
#pragma unroll
        for (int j = 0; j < L::NI; ++j) {
            if (j < n_q) {
                const f32x4 u = t[j] * p.scale;
                asm volatile("ds_write_b128 %0, %1" ::"v"(qs_a + j * 1024 + lane * 16), "v"(u) : "memory");
            }
        }
    }
    // (measured: re-reading these fragments from the images per tile gets the kernel under 256 registers -- six instead of
    // four workgroups per CU -- and runs SLOWER: 59.9 vs 52.5 us on the C2 cross shape)
    f32x4 qf[DH / 8], dof[DH / 8];
    read_frags<DH>(qf, qs_a, foff);
    read_frags<DH>(dof, dos_a, foff);
    lgkm_wait();
    frags_arrived<DH>(qf);
    frags_arrived<DH>(dof);
    float dsum = 0.f;
#pragma unroll
    for (int jj = 0; jj < DH / 8; ++jj)
        dsum += (dof[jj].x * of[jj].x + dof[jj].y * of[jj].y) + (dof[jj].z * of[jj].z + dof[jj].w * of[jj].w);
    dsum += __shfl_xor(dsum, 32, 64);
    {
        const __amdgpu_buffer_rsrc_t d_rs = out_rsrc(p.delta);
        const unsigned vo = (lane < 32 && l31 < p.Tq) ? ((unsigned)bh * (unsigned)p.Tq + (unsigned)l31) * 4u : OOB;
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dsum), d_rs, vo, 0, 0);
        issued += 1;
    }
    line_put(line_a, l31 < p.Tq ? dsum : 0.f, lane);
    line_put(lse_line, l31 < p.Tq ? lse_q : INFINITY, lane);
    PHASE(1);

    f32x16 dqacc[DH / 32];
#pragma unroll
    for (int dt = 0; dt < DH / 32; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dqacc[dt][r] = 0.f;
    const __amdgpu_buffer_rsrc_t dk_rs = out_rsrc(p.dk + (size_t)b * p.Tk * p.lddk + h * DH);
    const __amdgpu_buffer_rsrc_t dv_rs = out_rsrc(p.dv + (size_t)b * p.Tk * p.lddv + h * DH);
    // store addressing: per-lane part (its half-wave's row offset, its column) + a scalar part per (row, column block)
    const unsigned dk_lane = ((unsigned)(4 * half) * (unsigned)p.lddk + (unsigned)l31) * 4u;
    const unsigned dv_lane = ((unsigned)(4 * half) * (unsigned)p.lddv + (unsigned)l31) * 4u;
    // the dS tile inside the (dead) V image: [key][36] floats, written from the key-on-lane layout, read query-on-lane
    const unsigned t_wr = vs_a + l31 * 144 + 16 * half, t_rd = vs_a + l31 * 4 + half * 4 * 144;

    for (int kt = 0; kt < nkt; ++kt) {
        const int k0 = kt * 32, key = k0 + l31;
        float pd[16], ds[16];
        {
            f32x4 ll[4], dl[4];
            f32x16 s, dp;
            {
                f32x4 kf[DH / 8], vf[DH / 8];
                vm_wait(issued - mark_k);
                PHASE(2);
                read_frags<DH>(kf, ks_a, foff);
                vm_wait(issued - mark_v);
                read_frags<DH>(vf, vs_a, foff);
                line_get(ll, lse_line, half);
                line_get(dl, line_a, half);
                lgkm_wait();
                frags_arrived<DH>(kf);
                frags_arrived<DH>(vf);
                tie4(ll);
                tie4(dl);
                s = mma_frags<DH>(qf, kf);     // S[query][key]
                dp = mma_frags<DH>(dof, vf);   // dP[query][key]
                PHASE(3);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qq = row_of(r, half);
                const bool vis = key < p.Tk && qq < p.Tq && (!causal || key <= qq + off);
                const float pv = vis ? __expf(s[r] - ll[r >> 2][r & 3]) : 0.f;
                float pdr = pv, gr = dp[r];
                if (p.drop.on()) {
                    const uint32_t idx = ((uint32_t)bh * (uint32_t)p.Tq + (uint32_t)qq) * (uint32_t)p.Tk + (uint32_t)key;
                    const bool keep = p.drop.keep(idx);
                    pdr = keep ? pv * p.drop.scale : 0.f;
                    gr = keep ? gr * p.drop.scale : 0.f;
                }
                pd[r] = pdr;
                ds[r] = vis ? pv * (gr - dl[r >> 2][r & 3]) : 0.f;  // (dP of an invalid row / column is stale LDS: may be NaN)
            }
        }
        PHASE(4);
        // dS -> query-on-lane layout through the V image (its fragments are in registers)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const f32x4 t = {ds[4 * g4], ds[4 * g4 + 1], ds[4 * g4 + 2], ds[4 * g4 + 3]};
            asm volatile("ds_write_b128 %0, %1" ::"v"(t_wr + 32 * g4), "v"(t) : "memory");
        }
        {
            float dsT[16], krow[16 * (DH / 32)];
            read_rows<DH>(krow, ks_a, rbase);
            ts_read(dsT, t_rd);
            lgkm_wait();
            rows_arrived<DH>(krow, half, p.Tk - k0);
            tie16(dsT);
            // K and V images are free: the next tile is requested now and lands under the 96 products below
            if (kt + 1 < nkt) {
                issued += dma_rows<DH>(Ks, kg, p.ldk, k0 + 32, p.Tk, lane);
                mark_k = issued;
                issued += dma_rows<DH>(Vs, vg, p.ldv, k0 + 32, p.Tk, lane);
                mark_v = issued;
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int dt = 0; dt < DH / 32; ++dt)
                    dqacc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(dsT[r], krow[r * (DH / 32) + dt], dqacc[dt], 0, 0, 0);
        }
        PHASE(5);
        // dV, dK of this tile: complete after one pass; rows = keys (two 128-byte row segments per store instruction)
        auto tile_product = [&](const float* a, unsigned img_a, const __amdgpu_buffer_rsrc_t& rs, unsigned lane_off, int ldd) {
            float rows[16 * (DH / 32)];
            read_rows<DH>(rows, img_a, rbase);
            lgkm_wait();
            rows_arrived<DH>(rows, half, p.Tq);
            f32x16 acc[DH / 32];
#pragma unroll
            for (int dt = 0; dt < DH / 32; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[dt][r] = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int dt = 0; dt < DH / 32; ++dt)
                    acc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r], rows[r * (DH / 32) + dt], acc[dt], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kr = k0 + (r & 3) + 8 * (r >> 2);  // this register's key row in half-wave 0
                const unsigned vo = kr + 4 * half < p.Tk ? lane_off : OOB_S;
#pragma unroll
                for (int dt = 0; dt < DH / 32; ++dt)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[dt][r]), rs, vo, (kr * ldd + dt * 32) * 4, 0);
            }
            issued += 16 * (DH / 32);
            asm volatile("" ::: "memory");
        };
        tile_product(pd, dos_a, dv_rs, dv_lane, p.lddv);  // dV[key][:] = sum_q Pd[q][key] dO[q][:]
        tile_product(ds, qs_a, dk_rs, dk_lane, p.lddk);   // dK[key][:] = sum_q dS[q][key] (scale Q)[q][:]
        PHASE(6);
    }
    // causal with fewer visible keys than Tk: the key tiles past the last visible one get zero gradients
    for (int k0 = nkt * 32; k0 < p.Tk; k0 += 32) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kk = k0 + row_of(r, half);
#pragma unroll
            for (int dt = 0; dt < DH / 32; ++dt) {
                const unsigned ok = kk < p.Tk;
                const unsigned vk = ok ? ((unsigned)kk * (unsigned)p.lddk + (unsigned)(dt * 32 + l31)) * 4u : OOB;
                const unsigned vv = ok ? ((unsigned)kk * (unsigned)p.lddv + (unsigned)(dt * 32 + l31)) * 4u : OOB;
                __builtin_amdgcn_raw_buffer_store_b32(0u, dk_rs, vk, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(0u, dv_rs, vv, 0, 0);
            }
        }
    }
    const __amdgpu_buffer_rsrc_t dq_rs = out_rsrc(p.dq + (size_t)b * p.Tq * p.lddq + h * DH);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int qq = row_of(r, half);
#pragma unroll
        for (int dt = 0; dt < DH / 32; ++dt) {
            const unsigned vo = qq < p.Tq ? ((unsigned)qq * (unsigned)p.lddq + (unsigned)(dt * 32 + l31)) * 4u : OOB;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dqacc[dt][r] * p.scale), dq_rs, vo, 0, 0);
        }
    }
#ifdef CAPDEC_RING_STAMPS
    PHASE(7);
    if (lane == 0 && blockIdx.x < 2048) {
#pragma unroll
        for (int i = 0; i < 8; ++i) g_ring_dbg[blockIdx.x * 8 + i] = ph[i];
    }
#endif
#undef PHASE
}

// workgroups per CU the ring kernels are launched with: what the LDS footprint allows, at most 8
template <typename K>
static int resident_per_cu(K kernel) {
    // one-wave workgroups: what the LDS footprint and the register allocation allow (512 registers per lane and SIMD,
    // four SIMDs), at most 8
    hipFuncAttributes a;
    int n = 4;
    if (hipFuncGetAttributes(&a, reinterpret_cast<const void*>(kernel)) == hipSuccess) {
        const int lds = (int)a.sharedSizeBytes > 0 ? 163840 / (int)a.sharedSizeBytes : 8;
        const int regs = a.numRegs > 0 ? 4 * (512 / ((a.numRegs + 7) / 8 * 8)) : 8;
        n = lds < regs ? lds : regs;
        if (getenv("CAPDEC_DEBUG_ATTN"))
            fprintf(stderr, "[capdec] ring kernel: %d B of LDS, %d registers -> %d workgroups per CU\n", (int)a.sharedSizeBytes,
                    a.numRegs, n);
    }
    return n < 1 ? 1 : (n > 8 ? 8 : n);
}
static int cu_count() {
    static int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
        return n;
    }();
    return cus;
}

template <int DH, int NS>
static void launch_fwd_ring(const AttnParams& p, hipStream_t st) {
    static const int per_cu = resident_per_cu(attn_fwd_ring_kernel<DH, NS>);
    const int n_items = p.B * p.H * ((p.Tq + 31) / 32);
    int grid = cu_count() * per_cu;
    if (n_items % 8 == 0 && (p.B * p.H) % 8 == 0) grid -= grid % 8;  // items keep their XCD from step to step
    if (grid > n_items) grid = n_items;
    hipLaunchKernelGGL((attn_fwd_ring_kernel<DH, NS>), dim3(grid), dim3(64), 0, st, p, n_items);
}

static bool ring_mat_ok(int rows, int ld) { return (uint64_t)rows * (uint64_t)ld < (1ull << 30); }  // 32-bit byte offsets

bool attn_ring_fwd(const AttnParams& p, int dh, hipStream_t st) {
    if (p.Tk <= 1 || p.kv_bf16 || (dh != 64 && dh != 128)) return false;
    if (!ring_mat_ok(p.Tq, p.ldq) || !ring_mat_ok(p.Tk, p.ldk) || !ring_mat_ok(p.Tk, p.ldv) || !ring_mat_ok(p.Tq, p.ldo)) return false;
    if ((uint64_t)p.B * p.H * p.Tq >= (1ull << 29)) return false;
    static const int ns = [] { const char* e = getenv("CAPDEC_RING_NS"); return e ? atoi(e) : 0; }();  // experiments
    if (dh == 64) {
        if (ns == 4) launch_fwd_ring<64, 4>(p, st);
        else if (ns == 3) launch_fwd_ring<64, 3>(p, st);
        else launch_fwd_ring<64, 2>(p, st);
    } else {
        return false;  // (a 16 KB image per tile: the ring form measured 2.2-2.6x slower than the legacy kernel at head dim 128)
    }
    return true;
}

bool attn_ring_bwd(const AttnParams& p, int dh, hipStream_t st) {
    if (p.Tk <= 1 || p.Tq > 32 || p.kv_bf16 || p.dkv_bf16 || dh != 64) return false;
    if (!ring_mat_ok(p.Tq, p.ldq) || !ring_mat_ok(p.Tk, p.ldk) || !ring_mat_ok(p.Tk, p.ldv) || !ring_mat_ok(p.Tq, p.ldo) ||
        !ring_mat_ok(p.Tq, p.lddo) || !ring_mat_ok(p.Tq, p.lddq) || !ring_mat_ok(p.Tk, p.lddk) || !ring_mat_ok(p.Tk, p.lddv))
        return false;
    if ((uint64_t)p.B * p.H * p.Tq >= (1ull << 29)) return false;
    const int qr = (p.Tq + 3) / 4 * 4;
    const size_t lds = 2 * RImg<64>::BYTES + 2 * (size_t)qr * RImg<64>::ROWB + 256;
    hipLaunchKernelGGL((attn_bwd_ring_kernel<64>), dim3(p.B * p.H), dim3(64), lds, st, p, qr);
    return true;
}

}  // namespace capdec

#ifdef CAPDEC_RING_STAMPS
extern "C" int capdec_debug_ring_stamps(unsigned long long* host_out, int n) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(capdec::g_ring_dbg), (size_t)n * 4 * sizeof(unsigned long long)) == hipSuccess ? 0 : -2;
}
#endif
