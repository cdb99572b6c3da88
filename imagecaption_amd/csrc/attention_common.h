// Pieces shared by the attention kernels (attention_f32.hip, attention_bf16.hip).
#pragma once
#include "common.h"

namespace capdec {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// row of a 32 x 32 MFMA accumulator held in register r of a lane of half-wave `half`
__device__ __forceinline__ int row_of(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// parameters of the fp32 attention kernels (attention_f32.hip, attention_f32_ring.hip)
struct AttnParams {
    const float* q; const float* k; const float* v; const float* o; const float* dout;
    float* out; float* lse; float* delta; float* dq; float* dk; float* dv;
    int ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv;
    int B, H, Tq, Tk, causal;
    int dkv_bf16;  // dk / dv point to bf16 matrices (lddk / lddv in bf16 elements): capdec_attn_bwd_dkv16
    int kv_bf16;   // k / v point to bf16 matrices (ldk / ldv in bf16 elements): capdec_attn_*_kv16
    float scale;
    Dropout drop;
};

// ring kernels (attention_f32_ring.hip): persistent one-wave workgroups streaming their tiles global -> LDS by DMA;
// head dims 64 / 128, Tk > 1; the backward covers one query tile (Tq <= 32).  `false` = not covered (legacy kernels).
bool attn_ring_fwd(const AttnParams& p, int dh, hipStream_t st);
bool attn_ring_bwd(const AttnParams& p, int dh, hipStream_t st);

// Workgroup -> (batch*head, tile).  One workgroup reads 256-byte head slices out of rows that hold all H heads side
// by side; which XCD (and so which L2) serves a workgroup is blockIdx % 8 (round-robin dispatch).  With the natural
// order, head h of every image lands on XCD h % 8 and each L2 / memory channel sees nothing but isolated 256-byte
// pieces.  Here each XCD gets a CONTIGUOUS run of (b, h) items in its own dispatch order, so the H heads of one
// image run on one XCD at about the same time and its L2 is asked for whole rows (speed only, never correctness).
// Causal launches keep their longest-tiles-first order inside that run.
__device__ __forceinline__ void attn_item_at(int blk, int n_bh, int n_tiles, bool longest_first, int& bh, int& tile);
__device__ __forceinline__ void attn_item(int n_bh, int n_tiles, bool longest_first, int& bh, int& tile) {
    attn_item_at(blockIdx.x, n_bh, n_tiles, longest_first, bh, tile);
}
// the same map for work item `blk` of a persistent grid (item = workgroup + k * grid: the XCD stays the workgroup's)
__device__ __forceinline__ void attn_item_at(int blk, int n_bh, int n_tiles, bool longest_first, int& bh, int& tile) {
    if (n_bh % 8 == 0) {
        const int xcd = blk & 7, idx = blk >> 3, per = n_bh >> 3;  // per-XCD share of the (b, h) items
        if (longest_first) {
            tile = idx / per;
            bh = xcd * per + idx % per;
        } else {
            tile = idx % n_tiles;
            bh = xcd * per + idx / n_tiles;
        }
        return;
    }
    if (longest_first) {
        tile = blk / n_bh;
        bh = blk % n_bh;
    } else {
        tile = blk % n_tiles;
        bh = blk / n_tiles;
    }
}

}  // namespace capdec
