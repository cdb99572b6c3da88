// bf16-storage GEMM, 256 x 256 block, eight phases per pair of k-tiles ("p8" kernel; plan 256 x 256, k-tile field 32).
//
// Same block as the pair kernel of gemm_bf16s_dma.hip (eight waves as 2 x 4, 128 x 64 outputs each, 128 KB of LDS, one
// workgroup per CU, two wave groups one barrier apart so that on every SIMD one wave multiplies while the other reads),
// but staged and consumed in QUARTERS of a k-tile, which is what lets loads stay in flight across barriers:
//   * a k-tile (64 deep) is four half-tiles of 16 KB -- A rows 0-127 / 128-255 of the tile, B columns likewise -- and
//     eight half-tile slots (two k-tiles) make up the LDS;
//   * every wave takes 64 rows from EACH A half and 32 columns from EACH B half: wave (wr, wc) owns tile rows
//     128 mq + 64 wr + (0..63) and tile columns 128 nq + 32 wc + (0..31), mq, nq = 0, 1.  A phase multiplies one
//     64 x 32 quadrant (mq, nq) of that over the whole k-tile (8 MFMAs 32 x 32 x 16), so each half-tile is read in
//     exactly one phase of the four:
//         phase 0: read B half 0 (4 ds_read_b128) + A half 0 (8)   -> quadrant (0, 0)
//         phase 1: read B half 1 (4)                               -> quadrant (0, 1)
//         phase 2: read A half 1 (8)                               -> quadrant (1, 1)
//         phase 3: (B half 0 is still in registers)                -> quadrant (1, 0)
//   * so a slot is free one to two phases after it was read, and every phase re-fills one: phase p requests the
//     half-tile that is needed 7 phases later (need order B0, A0, B1, A1).  Seven half-tiles are in flight or landed
//     ahead of the reader; once per k-tile (phase 3) a wave waits with vmcnt(6): everything but its last three requests
//     (two DMA instructions each) has landed, which is exactly the next k-tile.  The counter is never drained in the loop.
// Order rules (the two wave groups run one barrier apart; barrier n is the n-th s_barrier every wave executes):
//   RAW  a half-tile is read one phase (two barriers) after the vmcnt wait that covers it: both groups have then waited.
//   WAR  B half 0's slot is re-requested in the phase after its reads, so those reads are retired (a counted lgkmcnt: the
//        B reads are issued first) BEFORE the barrier that ends their read phase; the other slots are re-requested two
//        phases after their reads, which are retired by the lgkmcnt(0) in front of the MFMAs.
// All LDS reads are inline assembly (an ordinary LDS read beside an LDS DMA in flight draws a compiler-inserted
// s_waitcnt vmcnt(0): see attention_f32_ring.hip), barriers are bare s_barrier.
// Operand layouts as in gemm_bf16s_dma.hip: a k-contiguous half-tile is [128 rows][64 k] (128-byte rows, 16-byte pieces
// XOR-swizzled on the source side, fragments by ds_read_b128); a k-major one is [64 k][128 rows] (256-byte k-rows, its own
// swizzle, fragments by two transposing ds_read_b64_tr_b16).  Bias-gradient sums (colsum, k-major A) are taken from the A
// half-tiles in the phases that read them.  The epilogue is the vector form (gemm_common.h) with this kernel's wave geometry.
#include "common.h"
#include <type_traits>
#include <cstdlib>

#include "gemm_common.h"

namespace capdec {

typedef __bf16 bf16x8d __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

#define CAPDEC_GLDS16(gptr, lptr) \
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(gptr), \
                                     (void __attribute__((address_space(3)))*)(lptr), 16, 0, 0)

constexpr int P8_BK = 64;
constexpr int P8_HALF = 128 * P8_BK * 2;  // bytes of a half-tile
constexpr int P8_B0 = 4 * P8_HALF;        // B half-tiles start here (A: [buf][half], then B: [buf][half])
#ifndef CAPDEC_P8_MGROUP
#define CAPDEC_P8_MGROUP 8
#endif

template <int OFF>
__device__ __forceinline__ i32x4 p8_read128(unsigned addr) {
    i32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int OFF>
__device__ __forceinline__ i32x2 p8_read_tr(unsigned addr) {  // 4 k-rows x 16 columns -> each lane its column's 4 k values
    i32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}

// one operand's fragments of a half-tile: NB row blocks (32 rows) x 4 k-steps (16 deep)
template <bool KC, int NB>
struct P8Frag {
    i32x4 q[NB][4];      // k-contiguous: one ds_read_b128 each
    i32x2 t[NB][4][2];   // k-major: two transposing reads each
    __device__ __forceinline__ bf16x8d get(int b, int ks) const {
        if constexpr (KC) {
            return __builtin_bit_cast(bf16x8d, q[b][ks]);
        } else {
            const i32x4 v = {t[b][ks][0][0], t[b][ks][0][1], t[b][ks][1][0], t[b][ks][1][1]};
            return __builtin_bit_cast(bf16x8d, v);
        }
    }
};

// The vocabulary product's epilogue on this kernel: logits = acc + bias stored as fp32 (the vector form's loop), and on the
// way the softmax partials of the fused cross-entropy (ref train.py:41-50; gemm_common.h ce_tile_partials is the generic
// form: 8 chunks of 32 rows through LDS with two workgroup barriers each, ~125 us of an 870 us launch at one workgroup per
// CU).  Here a row of the wave's block is 16 lanes x 4 columns in the vector loop: (max, sum exp(z - max), sum z) over
// the wave's 64 columns by four shuffle steps each, left in LDS per (row, wave column); after ONE barrier 256 threads
// combine the four wave columns of their row in a fixed order.  Deterministic; the partials' rounding differs from the
// generic form's in the last bits (another summation order), which ce_finalize's own sum over the tiles absorbs.
__device__ __forceinline__ void p8_epilogue_ce(const GemmParams& p, const f32x16 (&acc)[4][2], int m0, int n0, int wr, int wc,
                                               int lane, float* wl, float4* part, int tile_n, int tid) {
    const int l31 = lane & 31, half = lane >> 5, c4 = (lane & 15) * 4, rsub = lane >> 4;
    const int n = n0 + wc * 32 + (c4 & 31) + (c4 >> 5) * 128;
    const bool n_ok = n < p.N;  // N % 4 == 0 (epilogue_vec_ok): all four columns or none
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias && n_ok) b4 = *reinterpret_cast<const float4*>(p.bias + n);
    // with the targets at hand the partial's fourth component carries the row's target logit (if its column is in this
    // tile): the kernel staged the tile's 256 targets behind the partials (yt; -1 for rows past M)
    const int* yt = reinterpret_cast<const int*>(part + 256 * 4);
    const bool ZY = p.ce_y != nullptr, C16 = p.c16 != nullptr;
    // which of this lane's 32 row groups hold their row's target column: one batch of LDS reads here instead of a
    // dependent read in every turn of the store loop (a hit is rare: 4 columns out of V)
    unsigned hit = 0;
    if (ZY) {
#pragma unroll
        for (int g = 0; g < 32; ++g)
            hit |= ((unsigned)(yt[(g >> 4) * 128 + wr * 64 + rsub + (g & 15) * 4] - n) < 4u ? 1u : 0u) << g;
    }
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    wl[(ii * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * 64 + j * 32 + l31] = acc[ps * 2 + ii][j][r];
        const int row0 = ps * 128 + wr * 64 + rsub;  // tile row of this lane's first row group
#pragma unroll 1
        for (int it = 0; it < 16; ++it) {
            const int trow = row0 + it * 4, m = m0 + trow;
            const float4 v = *reinterpret_cast<const float4*>(&wl[(it * 4 + rsub) * 64 + c4]);
            const float e[4] = {v.x + b4.x, v.y + b4.y, v.z + b4.z, v.w + b4.w};
            float mx = n_ok ? fmaxf(fmaxf(e[0], e[1]), fmaxf(e[2], e[3])) : -INFINITY;
            mx = group_max<16>(mx);
            const float mref = mx == -INFINITY ? 0.f : mx;  // a wave block wholly past column N: no exp(-inf + inf)
            float se = n_ok ? (__expf(e[0] - mref) + __expf(e[1] - mref)) + (__expf(e[2] - mref) + __expf(e[3] - mref)) : 0.f;
            float sz = n_ok ? (e[0] + e[1]) + (e[2] + e[3]) : 0.f;
            se = group_sum<16>(se);
            sz = group_sum<16>(sz);
            if ((lane & 15) == 0) part[trow * 4 + wc] = make_float4(mx, se, sz, 0.f);
            if ((hit >> (ps * 16 + it)) & 1u) {  // the one lane that holds the row's target column overwrites the zero (a wave's LDS writes land in order)
                const int yr = yt[trow] - n;     // 0..3 (then n_ok: targets are < N)
                reinterpret_cast<float*>(&part[trow * 4 + wc])[3] = yr == 0 ? e[0] : yr == 1 ? e[1] : yr == 2 ? e[2] : e[3];
            }
            if (m < p.M && n_ok) ce_store4(p, m, n, e, C16);
        }
    }
    __syncthreads();
    if (tid < 256 && m0 + tid < p.M) {
        const float4 q0 = part[tid * 4], q1 = part[tid * 4 + 1], q2 = part[tid * 4 + 2], q3 = part[tid * 4 + 3];
        const float mx = fmaxf(fmaxf(q0.x, q1.x), fmaxf(q2.x, q3.x));  // finite: column 0 of the tile is inside the matrix
        const float se = ((q0.y * __expf(q0.x - mx) + q1.y * __expf(q1.x - mx)) + q2.y * __expf(q2.x - mx)) + q3.y * __expf(q3.x - mx);
        const float sz = ((q0.z + q1.z) + q2.z) + q3.z;
        p.ce_part[(size_t)(m0 + tid) * p.tiles_n + tile_n] = make_float4(mx, se, sz, ((q0.w + q1.w) + q2.w) + q3.w);
    }
}

#ifdef CAPDEC_P8_DIAG
// diagnostic builds only (tools/p8_stamps.py): s_memtime of wave 0 of every workgroup at kernel entry, first k-tile landed,
// main loop done, stores drained
__device__ unsigned long long g_p8_stamps[4096][4];
#define P8_STAMP(i) do { if (tid == 0 && blockIdx.x < 4096) g_p8_stamps[blockIdx.x][i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define P8_STAMP(i) do { } while (0)
#endif

// ABL (diagnostic builds only): 1 = no DMA after the prologue, 2 = no MFMA, 3 = no fragment reads, 4 = no epilogue store
template <bool AKC, bool BKC, bool SPLITK, int ABL = 0>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2)))
void gemm_bf16s_p8_kernel(GemmParams p) {
    constexpr int BM = 256, BN = 256, TM = 4, TN = 2, NT = 512, WN = 4;
    // eight half-tile slots; behind them 16 KB where the waves leave their per-row cross-entropy partials and 1 KB for the
    // tile's targets (epilogue only)
    __shared__ __attribute__((aligned(1024))) unsigned char smem[8 * P8_HALF + 256 * 4 * 16 + 256 * 4];

    const int tid = threadIdx.x;
    P8_STAMP(0);
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int wr = wave / WN, wc = wave % WN;
    const bool late = wave >= WN;  // wave-uniform: the group that runs one barrier behind
    const int nwg = gridDim.x;
    int wg = blockIdx.x;
    {   // XCD-aware tile order: a contiguous run of tiles per XCD (blocks b and b + 8 share one)
        const int q = nwg >> 3, r = nwg & 7, xcd = wg & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wg >> 3);
    }
    int tile_m, tile_n;
    {
        const int tiles_m = nwg / p.tiles_n;
        const int per_group = CAPDEC_P8_MGROUP * p.tiles_n;
        const int gid = wg / per_group, first_m = gid * CAPDEC_P8_MGROUP;
        const int gsz = min(tiles_m - first_m, CAPDEC_P8_MGROUP);
        const int in_g = wg - gid * per_group;
        tile_m = first_m + in_g % gsz;
        tile_n = in_g / gsz;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kbeg = blockIdx.z * p.ksplit;
    const int kend = min(p.K, kbeg + p.ksplit);
    const int nk = (kend - kbeg + P8_BK - 1) / P8_BK;
    const __bf16* A16 = reinterpret_cast<const __bf16*>(p.A);
    const __bf16* B16 = reinterpret_cast<const __bf16*>(p.B);

    // DMA sources: this wave's two 1 KiB pieces of each half-tile.
    //   k-contiguous: piece c = local rows 8 c .. 8 c + 7 (128 B each); 16-byte piece s of local row lr goes to slot
    //       s ^ ((lr >> 1) & 7).  off = row * ld + 8 s (+ k0 per k-tile).
    //   k-major: piece c = k-rows 4 c .. 4 c + 3 (256 B each); 16-byte piece s (8 columns) of k-row kr goes to slot
    //       s ^ (((kr & 3) << 2) | ((kr >> 2) & 3)).  off = column (+ (k0 + kr) * ld per k-tile, k-row clamped to K - 1:
    //       the partner's zero padding kills the rows past K).
    unsigned offa[2][2], offb[2][2];  // [half][piece]
    int kra[2], krb[2];               // k-major: the k-row inside the tile of piece j
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = wave * 2 + j;
        kra[j] = krb[j] = c * 4 + (lane >> 4);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int lr = c * 8 + (lane >> 3), src_kc = (lane & 7) ^ ((lr >> 1) & 7);
            const int kr = c * 4 + (lane >> 4), src_km = (lane & 15) ^ (((kr & 3) << 2) | ((kr >> 2) & 3));
            if constexpr (AKC) offa[h][j] = (unsigned)min(m0 + h * 128 + lr, p.M - 1) * (unsigned)p.lda + src_kc * 8;
            else offa[h][j] = (unsigned)min(m0 + h * 128 + src_km * 8, p.lda - 8);
            if constexpr (BKC) offb[h][j] = (unsigned)min(n0 + h * 128 + lr, p.N - 1) * (unsigned)p.ldb + src_kc * 8;
            else offb[h][j] = (unsigned)min(n0 + h * 128 + src_km * 8, p.ldb - 8);
        }
    }
    unsigned char* dma_dst = smem + wave * 2048;  // wave-uniform
    const unsigned lds0 = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)smem;
    // fragment read addresses.
    //   k-contiguous: one per 16-deep k-step: row * 128 + (((2 ks + half) ^ ((l31 >> 1) & 7)) * 16); row blocks and
    //       slots by immediate offsets.
    //   k-major (transposing read; lane 4 q + p of 16-lane group g): k-row (g >> 1) * 8 + rd * 4 + q (+ 16 ks),
    //       piece ch = r0 / 8 + (g & 1) * 2 + (p >> 1), slot ch ^ ((q << 2) | (((g >> 1) * 2 + rd) & 3)), + 8 (p & 1);
    //       one address per (row block, rd), k-steps and slots by immediate offsets (4096 B per k-step).
    const int g4 = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
    unsigned ra[4], rb[4];  // k-contiguous: [ks];  k-major: A [ii * 2 + rd], B [rd]
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        const unsigned sl = (unsigned)(((2 * x + half) ^ ((l31 >> 1) & 7)) * 16);
        const int rd = x & 1;
        const int kr = (g4 >> 1) * 8 + rd * 4 + q4;
        const int swz = (q4 << 2) | (((g4 >> 1) * 2 + rd) & 3);
        if constexpr (AKC) {
            ra[x] = lds0 + (wr * 64 + l31) * 128 + sl;
        } else {
            const int r0 = wr * 64 + (x >> 1) * 32;
            const int ch = r0 / 8 + (g4 & 1) * 2 + (p4 >> 1);
            ra[x] = lds0 + kr * 256 + ((ch ^ swz) * 16) + 8 * (p4 & 1);
        }
        if constexpr (BKC) {
            rb[x] = lds0 + P8_B0 + (wc * 32 + l31) * 128 + sl;
        } else {
            const int ch = (wc * 32) / 8 + (g4 & 1) * 2 + (p4 >> 1);
            rb[x] = lds0 + P8_B0 + kr * 256 + ((ch ^ swz) * 16) + 8 * (p4 & 1);  // x = 0, 1 used
        }
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    P8Frag<AKC, 2> fa;        // A fragments of the current half (2 row blocks x 4 k-steps)
    P8Frag<BKC, 1> fb0, fb1;  // B halves 0 / 1

    const bool do_colsum = !AKC && p.colsum != nullptr && tile_n == 0;  // colsum[m] = sum_k A(m, k): the bias gradient
    float cs8[2][8];  // k-major A: per half, 8 columns x this thread's two k-rows of every k-tile
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 8; ++e) cs8[h][e] = 0.f;

#define P8_BAR() asm volatile("s_barrier" ::: "memory")
    // request half-tile (operand, half h) of k-tile kt into buffer BUF
    auto stage_a = [&](auto buf_c, auto h_c, int kt) {
        constexpr int SLOT = (decltype(buf_c)::value * 2 + decltype(h_c)::value) * P8_HALF;
        constexpr int H = decltype(h_c)::value;
        if (ABL == 1 && kt >= 2) return;
        const int k0 = kbeg + kt * P8_BK;
        if constexpr (AKC) {
            CAPDEC_GLDS16(A16 + k0 + offa[H][0], dma_dst + SLOT);
            CAPDEC_GLDS16(A16 + k0 + offa[H][1], dma_dst + SLOT + 1024);
        } else {
            CAPDEC_GLDS16(A16 + ((unsigned)min(k0 + kra[0], p.K - 1) * (unsigned)p.lda + offa[H][0]), dma_dst + SLOT);
            CAPDEC_GLDS16(A16 + ((unsigned)min(k0 + kra[1], p.K - 1) * (unsigned)p.lda + offa[H][1]), dma_dst + SLOT + 1024);
        }
    };
    auto stage_b = [&](auto buf_c, auto h_c, int kt) {
        constexpr int SLOT = P8_B0 + (decltype(buf_c)::value * 2 + decltype(h_c)::value) * P8_HALF;
        constexpr int H = decltype(h_c)::value;
        if (ABL == 1 && kt >= 2) return;
        const int k0 = kbeg + kt * P8_BK;
        if constexpr (BKC) {
            CAPDEC_GLDS16(B16 + k0 + offb[H][0], dma_dst + SLOT);
            CAPDEC_GLDS16(B16 + k0 + offb[H][1], dma_dst + SLOT + 1024);
        } else {
            CAPDEC_GLDS16(B16 + ((unsigned)min(k0 + krb[0], p.K - 1) * (unsigned)p.ldb + offb[H][0]), dma_dst + SLOT);
            CAPDEC_GLDS16(B16 + ((unsigned)min(k0 + krb[1], p.K - 1) * (unsigned)p.ldb + offb[H][1]), dma_dst + SLOT + 1024);
        }
    };
    auto read_a = [&](auto buf_c, auto h_c) {
        constexpr int SLOT = (decltype(buf_c)::value * 2 + decltype(h_c)::value) * P8_HALF;
        if (ABL == 3) return;
        if constexpr (AKC) {
            fa.q[0][0] = p8_read128<SLOT>(ra[0]);          fa.q[0][1] = p8_read128<SLOT>(ra[1]);
            fa.q[0][2] = p8_read128<SLOT>(ra[2]);          fa.q[0][3] = p8_read128<SLOT>(ra[3]);
            fa.q[1][0] = p8_read128<SLOT + 4096>(ra[0]);   fa.q[1][1] = p8_read128<SLOT + 4096>(ra[1]);
            fa.q[1][2] = p8_read128<SLOT + 4096>(ra[2]);   fa.q[1][3] = p8_read128<SLOT + 4096>(ra[3]);
        } else {
            fa.t[0][0][0] = p8_read_tr<SLOT>(ra[0]);           fa.t[0][0][1] = p8_read_tr<SLOT>(ra[1]);
            fa.t[0][1][0] = p8_read_tr<SLOT + 4096>(ra[0]);    fa.t[0][1][1] = p8_read_tr<SLOT + 4096>(ra[1]);
            fa.t[0][2][0] = p8_read_tr<SLOT + 8192>(ra[0]);    fa.t[0][2][1] = p8_read_tr<SLOT + 8192>(ra[1]);
            fa.t[0][3][0] = p8_read_tr<SLOT + 12288>(ra[0]);   fa.t[0][3][1] = p8_read_tr<SLOT + 12288>(ra[1]);
            fa.t[1][0][0] = p8_read_tr<SLOT>(ra[2]);           fa.t[1][0][1] = p8_read_tr<SLOT>(ra[3]);
            fa.t[1][1][0] = p8_read_tr<SLOT + 4096>(ra[2]);    fa.t[1][1][1] = p8_read_tr<SLOT + 4096>(ra[3]);
            fa.t[1][2][0] = p8_read_tr<SLOT + 8192>(ra[2]);    fa.t[1][2][1] = p8_read_tr<SLOT + 8192>(ra[3]);
            fa.t[1][3][0] = p8_read_tr<SLOT + 12288>(ra[2]);   fa.t[1][3][1] = p8_read_tr<SLOT + 12288>(ra[3]);
        }
    };
    auto read_b = [&](auto buf_c, auto h_c, P8Frag<BKC, 1>& fb) {
        constexpr int SLOT = (decltype(buf_c)::value * 2 + decltype(h_c)::value) * P8_HALF;  // rb already holds P8_B0
        if (ABL == 3) return;
        if constexpr (BKC) {
            fb.q[0][0] = p8_read128<SLOT>(rb[0]);   fb.q[0][1] = p8_read128<SLOT>(rb[1]);
            fb.q[0][2] = p8_read128<SLOT>(rb[2]);   fb.q[0][3] = p8_read128<SLOT>(rb[3]);
        } else {
            fb.t[0][0][0] = p8_read_tr<SLOT>(rb[0]);           fb.t[0][0][1] = p8_read_tr<SLOT>(rb[1]);
            fb.t[0][1][0] = p8_read_tr<SLOT + 4096>(rb[0]);    fb.t[0][1][1] = p8_read_tr<SLOT + 4096>(rb[1]);
            fb.t[0][2][0] = p8_read_tr<SLOT + 8192>(rb[0]);    fb.t[0][2][1] = p8_read_tr<SLOT + 8192>(rb[1]);
            fb.t[0][3][0] = p8_read_tr<SLOT + 12288>(rb[0]);   fb.t[0][3][1] = p8_read_tr<SLOT + 12288>(rb[1]);
        }
    };
    // bias-gradient sums over A half h of the k-tile in buffer BUF ([64 k][128]: thread t owns piece t & 15 = 8 columns
    // of k-rows t >> 4 and 32 + (t >> 4)); inline-assembly reads like all the others of the loop
    auto colsum_half = [&](auto buf_c, auto h_c) {
        constexpr int SLOT = (decltype(buf_c)::value * 2 + decltype(h_c)::value) * P8_HALF;
        constexpr int H = decltype(h_c)::value;
        if constexpr (!AKC) {
            if (do_colsum) {
                const int r0 = tid >> 4, r1 = 32 + (tid >> 4);
                const unsigned a0 = lds0 + r0 * 256 + (((tid & 15) ^ (((r0 & 3) << 2) | ((r0 >> 2) & 3))) * 16);
                const unsigned a1 = lds0 + r1 * 256 + (((tid & 15) ^ (((r1 & 3) << 2) | ((r1 >> 2) & 3))) * 16);
                i32x4 v0 = p8_read128<SLOT>(a0), v1 = p8_read128<SLOT>(a1);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v0), "+v"(v1));
                const bf16x8d x0 = __builtin_bit_cast(bf16x8d, v0), x1 = __builtin_bit_cast(bf16x8d, v1);
#pragma unroll
                for (int e = 0; e < 8; ++e) cs8[H][e] += (float)x0[e] + (float)x1[e];
            }
        }
    };
    auto multiply = [&](auto mq_c, const P8Frag<BKC, 1>& fb, auto nq_c) {
        constexpr int MQ = decltype(mq_c)::value, NQ = decltype(nq_c)::value;
        __builtin_amdgcn_sched_barrier(0);
        if (ABL == 2) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8d x = fa.get(0, ks), y = fa.get(1, ks), z = fb.get(0, ks);
                asm volatile("" ::"v"(x), "v"(y), "v"(z));
            }
            return;
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
                acc[MQ * 2 + ii][NQ] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa.get(ii, ks), fb.get(0, ks), acc[MQ * 2 + ii][NQ], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto retire_a = [&]() {  // every LDS read issued so far has returned
        if constexpr (AKC) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa.q[0][0]), "+v"(fa.q[0][1]), "+v"(fa.q[0][2]), "+v"(fa.q[0][3]),
                         "+v"(fa.q[1][0]), "+v"(fa.q[1][1]), "+v"(fa.q[1][2]), "+v"(fa.q[1][3]));
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa.t[0][0][0]), "+v"(fa.t[0][0][1]), "+v"(fa.t[0][1][0]), "+v"(fa.t[0][1][1]),
                         "+v"(fa.t[0][2][0]), "+v"(fa.t[0][2][1]), "+v"(fa.t[0][3][0]), "+v"(fa.t[0][3][1]),
                         "+v"(fa.t[1][0][0]), "+v"(fa.t[1][0][1]), "+v"(fa.t[1][1][0]), "+v"(fa.t[1][1][1]),
                         "+v"(fa.t[1][2][0]), "+v"(fa.t[1][2][1]), "+v"(fa.t[1][3][0]), "+v"(fa.t[1][3][1]));
        }
    };
    // the B reads (issued first) have returned; N = LDS reads issued after them that may still be in flight
    auto retire_b = [&](P8Frag<BKC, 1>& fb, auto n_c) {
        constexpr int N = decltype(n_c)::value;
        static_assert(N <= 15, "lgkmcnt is a 4-bit counter");
        if constexpr (BKC) {
            asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(fb.q[0][0]), "+v"(fb.q[0][1]), "+v"(fb.q[0][2]), "+v"(fb.q[0][3]) : "n"(N));
        } else {
            asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(fb.t[0][0][0]), "+v"(fb.t[0][0][1]), "+v"(fb.t[0][1][0]), "+v"(fb.t[0][1][1]),
                         "+v"(fb.t[0][2][0]), "+v"(fb.t[0][2][1]), "+v"(fb.t[0][3][0]), "+v"(fb.t[0][3][1]) : "n"(N));
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    constexpr int A_READS = AKC ? 8 : 16;  // LDS reads of an A half
    // the four phases of k-tile t, which lives in buffer BUF
    auto k_tile = [&](auto buf_c, int t) {
        constexpr int BUF = decltype(buf_c)::value;
        using B_ = std::integral_constant<int, BUF>;
        using O_ = std::integral_constant<int, BUF ^ 1>;
        const bool has1 = t + 1 < nk, has2 = t + 2 < nk;
        // phase 0
        read_b(B_{}, I0{}, fb0);
        read_a(B_{}, I0{});
        if (has1) stage_a(O_{}, I1{}, t + 1);
        retire_b(fb0, std::integral_constant<int, (A_READS < 15 ? A_READS : 15)>{});
        P8_BAR();
        retire_a();
        colsum_half(B_{}, I0{});
        multiply(I0{}, fb0, I0{});
        P8_BAR();
        // phase 1
        read_b(B_{}, I1{}, fb1);
        if (has2) stage_b(B_{}, I0{}, t + 2);
        P8_BAR();
        retire_b(fb1, I0{});
        multiply(I0{}, fb1, I1{});
        P8_BAR();
        // phase 2
        read_a(B_{}, I1{});
        if (has2) stage_a(B_{}, I0{}, t + 2);
        P8_BAR();
        retire_a();
        colsum_half(B_{}, I1{});
        multiply(I1{}, fb1, I1{});
        P8_BAR();
        // phase 3
        if (has2) {
            stage_b(B_{}, I1{}, t + 2);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // all but the last three requests: k-tile t + 1 has landed
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        P8_BAR();
        multiply(I1{}, fb0, I0{});
        P8_BAR();
    };

    if (nk > 0) {
        stage_b(I0{}, I0{}, 0);
        stage_a(I0{}, I0{}, 0);
        stage_b(I0{}, I1{}, 0);
        stage_a(I0{}, I1{}, 0);
    }
    if (nk > 1) {
        stage_b(I1{}, I0{}, 1);
        stage_a(I1{}, I0{}, 1);
        stage_b(I1{}, I1{}, 1);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    P8_BAR();            // every wave's pieces of k-tile 0 have landed
    P8_STAMP(1);
    if (late) P8_BAR();  // the stagger
    for (int t = 0; t < nk; t += 2) {
        k_tile(I0{}, t);
        if (t + 1 < nk) k_tile(I1{}, t + 1);
    }
    // the logits product with targets: this tile's target ids, requested here so that the closing barriers hide the latency
    constexpr bool CE = AKC && !SPLITK;  // capdec_gemm: ce_partials only with A row-major and without a K split
    int ytgt = -1;
    if constexpr (CE) {
        if (p.ce_part && p.ce_y && tid < BM && m0 + tid < p.M) ytgt = ce_target(p, m0 + tid, p.N);
    }
    if (!late) P8_BAR();  // the stagger, paid back
    P8_BAR();             // everyone is done with the images: the epilogue reuses the LDS
#undef P8_BAR
    P8_STAMP(2);

    float csum = 0.f;
    if constexpr (!AKC) {
        if (do_colsum) {  // 32 threads hold partial sums of each column of a half: combine through LDS in a fixed order
            float* red = reinterpret_cast<float*>(smem);  // [half][32][128]
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int e = 0; e < 8; ++e) red[(h * 32 + (tid >> 4)) * 128 + (tid & 15) * 8 + e] = cs8[h][e];
            __syncthreads();
            if (tid < BM) {
#pragma unroll
                for (int u = 0; u < 32; ++u) csum += red[((tid >> 7) * 32 + u) * 128 + (tid & 127)];
            }
            __syncthreads();
        }
    }
    if constexpr (SPLITK) {
        if (gridDim.z > 1 && !splitk_combine<TM, TN, NT>(p, acc, csum, do_colsum, tile_m * p.tiles_n + tile_n, m0, BM, tid,
                                                          reinterpret_cast<int*>(smem)))
            return;
    }
    if (do_colsum && tid < BM && m0 + tid < p.M) p.colsum[m0 + tid] = csum;
    if constexpr (CE) {
        if (p.ce_part && p.ce_y && tid < BM)  // the tile's targets, for p8_epilogue_ce
            reinterpret_cast<int*>(smem + 8 * P8_HALF + 256 * 4 * 16)[tid] = ytgt;
    }
    __syncthreads();  // whatever used the LDS last is done before the waves write their private regions below
    if (ABL == 4) {
        if (acc[0][0][0] == 123.456f) p.C[0] = acc[1][1][3] + acc[2][0][5] + acc[3][1][7];
        return;
    }
    if constexpr (CE) {
        if (p.ce_part) {  // the logits product: bias only (capdec_gemm checks), softmax partials per row and tile
            p8_epilogue_ce(p, acc, m0, n0, wr, wc, lane, reinterpret_cast<float*>(smem + wave * 16384),
                           reinterpret_cast<float4*>(smem + 8 * P8_HALF), tile_n, tid);
            return;
        }
    }
    // one workgroup per CU: nothing else hides the epilogue, so it is the vector form (gemm_common.h); a pass = the
    // wave's 64 rows of one A half (pass stride 128 rows), block columns 128 apart
    epilogue_store_vec<TM, TN, 2>(p, acc, m0 + wr * 64, n0 + wc * 32, lane, reinterpret_cast<float*>(smem + wave * 16384), 128, 128);
#ifdef CAPDEC_P8_DIAG
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    P8_STAMP(3);
#endif
}

// the products the p8 kernel takes (beyond gemm_bf16s_dma_ok): a result the vector epilogue can store
bool gemm_bf16s_p8_ok(const GemmParams& p, bool AKC, bool BKC) { return p.vec_epi != 0; }

template <bool AKC, bool BKC>
static void launch_p8_layout(const GemmParams& p, dim3 grid, hipStream_t st) {
#ifdef CAPDEC_P8_DIAG
    static const int abl = getenv("CAPDEC_P8_ABL") ? atoi(getenv("CAPDEC_P8_ABL")) : 0;
    if (abl == 1) { hipLaunchKernelGGL((gemm_bf16s_p8_kernel<AKC, BKC, false, 1>), grid, dim3(512), 0, st, p); return; }
    if (abl == 2) { hipLaunchKernelGGL((gemm_bf16s_p8_kernel<AKC, BKC, false, 2>), grid, dim3(512), 0, st, p); return; }
    if (abl == 3) { hipLaunchKernelGGL((gemm_bf16s_p8_kernel<AKC, BKC, false, 3>), grid, dim3(512), 0, st, p); return; }
    if (abl == 4) { hipLaunchKernelGGL((gemm_bf16s_p8_kernel<AKC, BKC, false, 4>), grid, dim3(512), 0, st, p); return; }
#endif
    if (grid.z > 1) hipLaunchKernelGGL((gemm_bf16s_p8_kernel<AKC, BKC, true>), grid, dim3(512), 0, st, p);
    else hipLaunchKernelGGL((gemm_bf16s_p8_kernel<AKC, BKC, false>), grid, dim3(512), 0, st, p);
}

void launch_gemm_bf16s_p8(const GemmParams& p, bool AKC, bool BKC, dim3 grid, hipStream_t st) {
    if (AKC && BKC) launch_p8_layout<true, true>(p, grid, st);
    else if (AKC) launch_p8_layout<true, false>(p, grid, st);
    else if (BKC) launch_p8_layout<false, true>(p, grid, st);
    else launch_p8_layout<false, false>(p, grid, st);
}

}  // namespace capdec

#ifdef CAPDEC_P8_DIAG
extern "C" int capdec_p8_stamps(void* dst, int n) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(capdec::g_p8_stamps), (size_t)n * 4 * sizeof(unsigned long long));
}
#endif
