// One decoder layer of a TRAINING step behind one C-ABI call per direction (capdec_train_layer_fwd / _bwd): the ~12
// forward and ~25 backward launches that engine.layer_fwd / layer_bwd issue op by op -- projections, attention,
// out-projections with dropout + residual, LayerNorms, FFN, their input- and weight-gradient products, the hand-over
// of the weight gradients to the side lane -- enqueued from C over a per-layer arena of fixed layout.  Same entry
// points, same arguments, same order on each stream as the Python plan (engine.py cites the reference lines:
// torch/nn/modules/transformer.py:1143-1199 as built at ref models.py:53-65): bit-identical results.  What it removes
// is the host cost per launch (a Python -> ctypes round trip, a torch allocation per result, an event pair per
// hand-over), which bounds the bf16-storage step at C2.  No kernels live in this file.
#include <math.h>
#include <string.h>

#include "common.h"

using namespace capdec;

namespace {

// ---- arena layout ------------------------------------------------------------------------------------------------
// One slot per tensor the layer keeps between forward and backward or hands between launches of one direction.  fp32
// compute types: every slot is an fp32 matrix of its natural width.  CAPDEC_BF16S: the slots marked "16" are bf16
// matrices with the row pitch of capdec_cast_bf16's callers (round up to 64, one more line where that is a multiple
// of 1 KiB), the others stay fp32.  Backward buffers are per layer as well: the side lane reads them (weight
// gradients) while the critical path has already moved on to the layer below.
enum Slot {
    SL_QKV, SL_A, SL_LSE1, SL_S1, SL_X1, SL_X1_16, SL_MEAN1, SL_RSTD1, SL_Q, SL_C, SL_LSE2, SL_S2, SL_X2, SL_X2_16, SL_MEAN2,
    SL_RSTD2, SL_U, SL_H, SL_S3, SL_X3, SL_X3_16, SL_MEAN3, SL_RSTD3,
    SL_G3, SL_G3M, SL_DU, SL_DX2, SL_G2, SL_G2M, SL_DC, SL_DQ, SL_DELTA, SL_DX1, SL_G1, SL_G1M, SL_DA, SL_DQKV, SL_DX,
    SL_PART1, SL_PART2, SL_PART3, SL_COUNT
};

static_assert(SL_X3 == CAPDEC_TL_X3 && SL_X3_16 == CAPDEC_TL_X3_16 && SL_DX == CAPDEC_TL_DX && SL_COUNT == CAPDEC_TL_SLOTS,
              "slot ids published in capdec.h");

int bf16_pitch(int cols) {
    int ld = (cols + 63) / 64 * 64;
    if (ld % 512 == 0) ld += 64;
    return ld;
}

struct SlotInfo {
    size_t offset, bytes;
    int ld;    // row stride in elements (0: a vector)
    int bf16;  // element type of the slot
};

struct Layout {
    SlotInfo s[SL_COUNT];
    size_t total;
};

bool geom_ok(const capdec_train_geom* g) {
    return g && g->B > 0 && g->T > 0 && g->S > 1 && g->H > 0 && g->d > 0 && g->dff > 0 && g->d % g->H == 0 && g->d % 4 == 0 &&
           g->dff % 4 == 0 && g->drop_p >= 0.f && g->drop_p < 1.f && g->dtype >= CAPDEC_F32 && g->dtype <= CAPDEC_BF16S &&
           (g->dtype != CAPDEC_BF16S || (g->d % 64 == 0 && g->dff % 64 == 0 && (g->d / g->H == 64 || g->d / g->H == 128))) &&
           (long long)g->B * g->T * (long long)(3 * g->d + 64) < (1ll << 31) && (long long)g->B * g->T * (long long)(g->dff + 64) < (1ll << 31);
}

Layout make_layout(const capdec_train_geom* g) {
    Layout L;
    memset(&L, 0, sizeof(L));
    const bool s16 = g->dtype == CAPDEC_BF16S;
    const size_t M = (size_t)g->B * g->T;
    const int d = g->d, dff = g->dff;
    size_t off = 0;
    auto put = [&](int slot, int cols, bool bf16, bool present) {
        SlotInfo& si = L.s[slot];
        si.bf16 = bf16;
        si.ld = bf16 ? bf16_pitch(cols) : cols;
        si.offset = off;
        si.bytes = present ? M * (size_t)si.ld * (bf16 ? 2 : 4) : 0;
        off += (si.bytes + 255) / 256 * 256;
    };
    auto vec = [&](int slot, size_t n, bool present) {
        SlotInfo& si = L.s[slot];
        si.ld = 0;
        si.offset = off;
        si.bytes = present ? n * 4 : 0;
        off += (si.bytes + 255) / 256 * 256;
    };
    const size_t bht = (size_t)g->B * g->H * g->T;
    const bool tr = g->drop_p > 0.f;
    put(SL_QKV, 3 * d, s16, true);
    put(SL_A, d, s16, true);
    vec(SL_LSE1, bht, true);
    put(SL_S1, d, false, true);
    put(SL_X1, d, false, true);
    put(SL_X1_16, d, true, s16);
    vec(SL_MEAN1, M, true);
    vec(SL_RSTD1, M, true);
    put(SL_Q, d, s16, true);
    put(SL_C, d, s16, true);
    vec(SL_LSE2, bht, true);
    put(SL_S2, d, false, true);
    put(SL_X2, d, false, true);
    put(SL_X2_16, d, true, s16);
    vec(SL_MEAN2, M, true);
    vec(SL_RSTD2, M, true);
    put(SL_U, dff, false, true);
    put(SL_H, dff, s16, true);
    put(SL_S3, d, false, true);
    put(SL_X3, d, false, true);
    put(SL_X3_16, d, true, s16);
    vec(SL_MEAN3, M, true);
    vec(SL_RSTD3, M, true);
    put(SL_G3, d, false, true);
    put(SL_G3M, d, s16, s16 || tr);
    put(SL_DU, dff, s16, true);
    put(SL_DX2, d, false, true);
    put(SL_G2, d, false, true);
    put(SL_G2M, d, s16, s16 || tr);
    put(SL_DC, d, s16, true);
    put(SL_DQ, d, s16, true);
    vec(SL_DELTA, bht, true);
    put(SL_DX1, d, false, true);
    put(SL_G1, d, false, true);
    put(SL_G1M, d, s16, s16 || tr);
    put(SL_DA, d, s16, true);
    put(SL_DQKV, 3 * d, s16, true);
    put(SL_DX, d, false, true);
    const size_t part = capdec_layernorm_bwd_workspace_bytes((int)M, d);
    for (int s = SL_PART1; s <= SL_PART3; ++s) {
        L.s[s].offset = off;
        L.s[s].bytes = part;
        off += (part + 255) / 256 * 256;
    }
    L.total = off;
    return L;
}

// ---- dropout seeds: engine.DropoutPlan.seed(16 * layer + site) ------------------------------------------------------
uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
enum { S_SA_P = 0, S_SA_OUT, S_CA_P, S_CA_OUT, S_FFN_IN, S_FFN_OUT };
uint64_t site_seed(uint64_t key, int layer, int site) {
    return splitmix64(key ^ ((uint64_t)(16 * layer + site) * 0xD6E8FEB86659FD93ull));
}

// ---- hand-over events: the lane waits for what the caller's stream has enqueued so far --------------------------------
// Reused in rotation (a wait refers to the record that preceded it; re-recording an event later is harmless).
const int kMaxDev = 16, kEvents = 64;
hipEvent_t g_events[kMaxDev][kEvents];
bool g_events_made[kMaxDev];
unsigned g_event_next[kMaxDev];

int hand_over(hipStream_t from, hipStream_t to) {
    if (from == to) return CAPDEC_OK;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) {
        set_error("capdec_train_layer: cannot tell the current device (or more than %d devices)", kMaxDev);
        return CAPDEC_ERR_LAUNCH;
    }
    if (!g_events_made[dev]) {
        for (int i = 0; i < kEvents; ++i)
            if (hipEventCreateWithFlags(&g_events[dev][i], hipEventDisableTiming) != hipSuccess) {
                set_error("capdec_train_layer: hipEventCreateWithFlags failed");
                return CAPDEC_ERR_LAUNCH;
            }
        g_events_made[dev] = true;
    }
    hipEvent_t ev = g_events[dev][g_event_next[dev]++ % kEvents];
    if (hipEventRecord(ev, from) != hipSuccess || hipStreamWaitEvent(to, ev, 0) != hipSuccess) {
        set_error("capdec_train_layer: event hand-over between the two streams failed");
        return CAPDEC_ERR_LAUNCH;
    }
    return CAPDEC_OK;
}

struct Mat {
    const void* p;
    int ld;
};

// One product with ops.gemm's conventions: a bf16 result goes to ep.c16 (C = NULL, ldc strides pre / aux or is N).
struct Gemm {
    int dtype;
    void* ws;
    size_t ws_bytes;
    capdec_stream_t stream;
    int run(int ta, int tb, int M, int N, int K, Mat a, Mat b, void* out, int ldo, bool out16, capdec_epilogue ep) const {
        void* C = out;
        int ldc = ldo;
        if (out16) {
            ep.c16 = out;
            ep.ldc16 = ldo;
            C = nullptr;
            ldc = N;  // (pre / aux of this file are dense [M, N] blocks)
        }
        return capdec_gemm(ta, tb, M, N, K, a.p, a.ld, b.p, b.ld, C, ldc, &ep, dtype, ws, ws_bytes, stream);
    }
};

capdec_epilogue ep0() {
    capdec_epilogue ep;
    memset(&ep, 0, sizeof(ep));
    return ep;
}

const float* f32(const void* p, size_t off = 0) { return (const float*)p + off; }
const void* col(const void* p, bool bf16, size_t elems) { return bf16 ? (const void*)((const uint16_t*)p + elems) : (const void*)((const float*)p + elems); }
void* col(void* p, bool bf16, size_t elems) { return bf16 ? (void*)((uint16_t*)p + elems) : (void*)((float*)p + elems); }

bool layer_ok(const capdec_train_layer* P, bool s16, bool bwd) {
    if (!P) return false;
    const bool params = P->w_in && P->b_in && P->w_out && P->b_out && P->ln1_w && P->ln1_b && P->wc_in && P->bc_in && P->wc_out &&
                        P->bc_out && P->ln2_w && P->ln2_b && P->w1 && P->b1 && P->w2 && P->b2 && P->ln3_w && P->ln3_b && P->arena;
    const bool grads = !bwd || (P->g_w_in && P->g_b_in && P->g_w_out && P->g_b_out && P->g_ln1_w && P->g_ln1_b && P->g_wc_in &&
                                P->g_bc_in && P->g_wc_out && P->g_bc_out && P->g_ln2_w && P->g_ln2_b && P->g_w1 && P->g_b1 && P->g_w2 &&
                                P->g_b2 && P->g_ln3_w && P->g_ln3_b);
    const bool copies = !s16 || (P->w_in16 && P->w_out16 && P->wcq16 && P->wc_out16 && P->w1_16 && P->w2_16 && P->ld_w_in16 > 0 &&
                                 P->ld_w_out16 > 0 && P->ld_wcq16 > 0 && P->ld_wc_out16 > 0 && P->ld_w1_16 > 0 && P->ld_w2_16 > 0);
    return params && grads && copies;
}

bool groupable(const void* p, int ld) { return aligned16(p) && ld % 4 == 0; }

}  // namespace

extern "C" size_t capdec_train_layer_arena_bytes(const capdec_train_geom* g) {
    if (!geom_ok(g)) return 0;
    return make_layout(g).total;
}

extern "C" int capdec_train_layer_slot(const capdec_train_geom* g, int slot, size_t* offset, size_t* bytes, int* ld, int* is_bf16) {
    CAPDEC_REQUIRE(geom_ok(g), "capdec_train_layer_slot: geometry not covered by the one-call layer plans");
    CAPDEC_REQUIRE(slot >= 0 && slot < SL_COUNT, "capdec_train_layer_slot: slot %d outside 0..%d", slot, SL_COUNT - 1);
    const Layout L = make_layout(g);
    if (offset) *offset = L.s[slot].offset;
    if (bytes) *bytes = L.s[slot].bytes;
    if (ld) *ld = L.s[slot].ld;
    if (is_bf16) *is_bf16 = L.s[slot].bf16;
    return CAPDEC_OK;
}

#define STEP(call) do { const int rc_ = (call); if (rc_) return rc_; } while (0)

extern "C" int capdec_train_layer_fwd(const capdec_train_layer* P, const capdec_train_geom* g, int layer, uint64_t dropout_key,
                                      const void* x, const void* x16, int ldx16, const void* kv, int ldkv, void* gemm_workspace,
                                      size_t gemm_workspace_bytes, capdec_stream_t stream) {
    CAPDEC_REQUIRE(geom_ok(g), "capdec_train_layer_fwd: geometry not covered (memory of one position, odd widths, or a head "
                               "dim the bf16 attention kernels do not have): use the entry points one by one");
    const bool s16 = g->dtype == CAPDEC_BF16S;
    CAPDEC_REQUIRE(layer_ok(P, s16, false), "capdec_train_layer_fwd: a parameter, bf16-copy or arena pointer is null");
    CAPDEC_REQUIRE(x && kv && (!s16 || x16), "capdec_train_layer_fwd: x / kv (and x16 with CAPDEC_BF16S) are required");
    const Layout L = make_layout(g);
    CAPDEC_REQUIRE(P->arena_bytes >= L.total && ((uintptr_t)P->arena & 255u) == 0,
                   "capdec_train_layer_fwd: the arena needs %zu bytes, 256-byte aligned (capdec_train_layer_arena_bytes)", L.total);
    char* base = (char*)P->arena;
    auto at = [&](int slot) -> void* { return base + L.s[slot].offset; };
    auto ld = [&](int slot) { return L.s[slot].ld; };
    const int B = g->B, T = g->T, S = g->S, H = g->H, d = g->d, dff = g->dff, M = B * T, dh = d / H;
    const float p = g->drop_p, scale = 1.0f / sqrtf((float)dh), eps = 1e-5f;
    const Gemm G{g->dtype, gemm_workspace, gemm_workspace_bytes, stream};
    auto seed = [&](int site) { return site_seed(dropout_key, layer, site); };
    auto W = [&](const void* w32, const void* w16, int ld16, int ld32) { return s16 ? Mat{w16, ld16} : Mat{w32, ld32}; };
    auto act_in = [&](const void* a32, const void* a16, int ld16) { return s16 ? Mat{a16, ld16} : Mat{a32, d}; };
    auto attn_fwd = s16 ? capdec_attn_fwd_bf16 : capdec_attn_fwd;
    auto ln = [&](int s_in, const void* gamma, const void* beta, int s_y, int s_y16, int s_mean, int s_rstd) {
        if (s16)
            return capdec_layernorm_fwd_y16(at(s_in), gamma, beta, at(s_y), at(s_y16), ld(s_y16), at(s_mean), at(s_rstd), M, d, eps,
                                            CAPDEC_ACT_NONE, 0.f, 0, stream);
        return capdec_layernorm_fwd(at(s_in), gamma, beta, at(s_y), at(s_mean), at(s_rstd), M, d, eps, CAPDEC_ACT_NONE, 0.f, 0,
                                    stream);
    };

    // self-attention block
    capdec_epilogue ep = ep0();
    ep.bias = P->b_in;
    STEP(G.run(0, 1, M, 3 * d, d, act_in(x, x16, ldx16), W(P->w_in, P->w_in16, P->ld_w_in16, d), at(SL_QKV), ld(SL_QKV), s16, ep));
    void* qkv = at(SL_QKV);
    STEP(attn_fwd(qkv, ld(SL_QKV), col(qkv, s16, d), ld(SL_QKV), col(qkv, s16, 2 * d), ld(SL_QKV), at(SL_A), ld(SL_A), at(SL_LSE1),
                  B, H, T, T, dh, g->causal, scale, p, seed(S_SA_P), stream));
    ep = ep0();
    ep.bias = P->b_out; ep.drop_p = p; ep.drop_seed = seed(S_SA_OUT); ep.addend = x; ep.ld_addend = d;
    STEP(G.run(0, 1, M, d, d, Mat{at(SL_A), ld(SL_A)}, W(P->w_out, P->w_out16, P->ld_w_out16, d), at(SL_S1), d, false, ep));
    STEP(ln(SL_S1, P->ln1_w, P->ln1_b, SL_X1, SL_X1_16, SL_MEAN1, SL_RSTD1));
    // cross-attention block over the image memory (its K | V projection comes from the caller's batched product)
    ep = ep0();
    ep.bias = P->bc_in;
    STEP(G.run(0, 1, M, d, d, act_in(at(SL_X1), at(SL_X1_16), ld(SL_X1_16)), W(P->wc_in, P->wcq16, P->ld_wcq16, d), at(SL_Q), ld(SL_Q),
               s16, ep));
    STEP(attn_fwd(at(SL_Q), ld(SL_Q), kv, ldkv, col(kv, s16, d), ldkv, at(SL_C), ld(SL_C), at(SL_LSE2), B, H, T, S, dh, 0, scale, p,
                  seed(S_CA_P), stream));
    ep = ep0();
    ep.bias = P->bc_out; ep.drop_p = p; ep.drop_seed = seed(S_CA_OUT); ep.addend = at(SL_X1); ep.ld_addend = d;
    STEP(G.run(0, 1, M, d, d, Mat{at(SL_C), ld(SL_C)}, W(P->wc_out, P->wc_out16, P->ld_wc_out16, d), at(SL_S2), d, false, ep));
    STEP(ln(SL_S2, P->ln2_w, P->ln2_b, SL_X2, SL_X2_16, SL_MEAN2, SL_RSTD2));
    // feed-forward block
    ep = ep0();
    ep.bias = P->b1; ep.act = CAPDEC_ACT_GELU; ep.pre = at(SL_U); ep.drop_p = p; ep.drop_seed = seed(S_FFN_IN);
    STEP(G.run(0, 1, M, dff, d, act_in(at(SL_X2), at(SL_X2_16), ld(SL_X2_16)), W(P->w1, P->w1_16, P->ld_w1_16, d), at(SL_H), ld(SL_H),
               s16, ep));
    ep = ep0();
    ep.bias = P->b2; ep.drop_p = p; ep.drop_seed = seed(S_FFN_OUT); ep.addend = at(SL_X2); ep.ld_addend = d;
    STEP(G.run(0, 1, M, d, dff, Mat{at(SL_H), ld(SL_H)}, W(P->w2, P->w2_16, P->ld_w2_16, dff), at(SL_S3), d, false, ep));
    STEP(ln(SL_S3, P->ln3_w, P->ln3_b, SL_X3, SL_X3_16, SL_MEAN3, SL_RSTD3));
    return CAPDEC_OK;
}

extern "C" int capdec_train_layer_bwd(const capdec_train_layer* P, const capdec_train_geom* g, int layer, uint64_t dropout_key,
                                      const void* dy, const void* x, const void* x16, int ldx16, const void* mem,
                                      const void* mem16, int ldmem16, const void* kv, int ldkv, void* dkv, int lddkv,
                                      void* gemm_workspace, size_t gemm_workspace_bytes, void* lane_gemm_workspace,
                                      size_t lane_gemm_workspace_bytes, capdec_stream_t stream, capdec_stream_t lane_stream) {
    CAPDEC_REQUIRE(geom_ok(g), "capdec_train_layer_bwd: geometry not covered by the one-call layer plans");
    const bool s16 = g->dtype == CAPDEC_BF16S;
    CAPDEC_REQUIRE(layer_ok(P, s16, true), "capdec_train_layer_bwd: a parameter, gradient, bf16-copy or arena pointer is null");
    CAPDEC_REQUIRE(dy && x && mem && kv && dkv && (!s16 || (x16 && mem16)),
                   "capdec_train_layer_bwd: dy / x / mem / kv / dkv (and x16 / mem16 with CAPDEC_BF16S) are required");
    const Layout L = make_layout(g);
    CAPDEC_REQUIRE(P->arena_bytes >= L.total && ((uintptr_t)P->arena & 255u) == 0,
                   "capdec_train_layer_bwd: the arena needs %zu bytes, 256-byte aligned (capdec_train_layer_arena_bytes)", L.total);
    const bool own_lane = lane_stream != stream;
    CAPDEC_REQUIRE(!own_lane || (lane_gemm_workspace && lane_gemm_workspace != gemm_workspace),
                   "capdec_train_layer_bwd: a weight-gradient lane needs a GEMM workspace of its own");
    char* base = (char*)P->arena;
    auto at = [&](int slot) -> void* { return base + L.s[slot].offset; };
    auto ld = [&](int slot) { return L.s[slot].ld; };
    const int B = g->B, T = g->T, S = g->S, H = g->H, d = g->d, dff = g->dff, M = B * T, Ms = B * S, dh = d / H;
    const float p = g->drop_p, scale = 1.0f / sqrtf((float)dh);
    const bool tr = p > 0.f;
    hipStream_t st = (hipStream_t)stream, lane = (hipStream_t)lane_stream;
    const Gemm G{g->dtype, gemm_workspace, gemm_workspace_bytes, stream};
    const Gemm GL{g->dtype, own_lane ? lane_gemm_workspace : gemm_workspace, own_lane ? lane_gemm_workspace_bytes : gemm_workspace_bytes,
                  lane_stream};
    auto seed = [&](int site) { return site_seed(dropout_key, layer, site); };
    auto W = [&](const void* w32, const void* w16, int ld16, int ld32) { return s16 ? Mat{w16, ld16} : Mat{w32, ld32}; };
    auto attn_bwd = s16 ? capdec_attn_bwd_bf16 : capdec_attn_bwd;

    // the weight gradients of the layer, queued in program order and handed to the lane together
    struct WG {
        Mat dyv, xv;
        int N, K, rows;  // dw [N, K] = dy[rows, N]^T x[rows, K]
        void* dw;
        int lddw;
        void* db;
    };
    WG queue[8];
    int nq = 0;
    // (LayerNorm parameter gradients ride on the next hand-over: a hand-over costs ~25 us of host time on this stack)
    struct LNFin {
        const void* part;
        void *dgamma, *dbeta;
    };
    LNFin fin[3];
    int nfin = 0;
    auto flush = [&](bool stage) -> int {
        if (nq == 0 && nfin == 0) return CAPDEC_OK;
        const bool grouping = g->dtype == CAPDEC_F32 && g->group_wgrads != 0;
        if (stage && grouping) return CAPDEC_OK;  // (they wait for the grouped launch at the layer's end)
        STEP(hand_over(st, lane));
        for (int i = 0; i < nfin; ++i) STEP(capdec_layernorm_bwd_dparams(fin[i].part, fin[i].dgamma, fin[i].dbeta, M, d, lane_stream));
        nfin = 0;
        bool in_group[8] = {false};
        int ng = 0;
        if (grouping)
            for (int i = 0; i < nq; ++i)
                if (groupable(queue[i].dyv.p, queue[i].dyv.ld) && groupable(queue[i].xv.p, queue[i].xv.ld) &&
                    groupable(queue[i].dw, queue[i].lddw)) {
                    in_group[i] = true;
                    ++ng;
                }
        if (ng < 2) {
            for (int i = 0; i < nq; ++i) in_group[i] = false;
            ng = 0;
        }
        if (ng) {
            // longest reductions first (stable): their tiles start first, the short ones fill the tail of the launch
            int order[8], n = 0;
            for (int i = 0; i < nq; ++i)
                if (in_group[i]) order[n++] = i;
            for (int i = 1; i < n; ++i)
                for (int j = i; j > 0 && queue[order[j]].rows > queue[order[j - 1]].rows; --j) {
                    const int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t;
                }
            capdec_gemm_desc descs[CAPDEC_GEMM_GROUP_MAX];
            capdec_epilogue eps[CAPDEC_GEMM_GROUP_MAX];
            for (int i = 0; i < n; ++i) {
                const WG& w = queue[order[i]];
                eps[i] = ep0();
                eps[i].colsum = w.db;
                descs[i] = capdec_gemm_desc{1, 0, w.N, w.K, w.rows, w.dyv.p, w.dyv.ld, w.xv.p, w.xv.ld, w.dw, w.lddw, &eps[i]};
            }
            STEP(capdec_gemm_grouped(descs, n, g->group_bm, g->group_bn, lane_stream));
        }
        for (int i = 0; i < nq; ++i) {
            if (in_group[i]) continue;
            const WG& w = queue[i];
            capdec_epilogue ep = ep0();
            ep.colsum = w.db;
            STEP(GL.run(1, 0, w.N, w.K, w.rows, w.dyv, w.xv, w.dw, w.lddw, false, ep));
        }
        nq = 0;
        return CAPDEC_OK;
    };
    // hand-over points inside the layer (measured at C2: two of them beat both "each product at once" and "all at the
    // layer's end" by 1 - 2 % of the step -- where the lane's products start decides which critical-path kernels they share
    // the matrix pipes with)
    const int flush_mask = g->wgrad_flush_mask ? g->wgrad_flush_mask : 0x12;
    int nwg = 0;
    auto wgrad = [&](Mat dyv, Mat xv, int N, int K, int rows, void* dw, void* db) -> int {
        queue[nq++] = WG{dyv, xv, N, K, rows, dw, K, db};
        return (flush_mask >> nwg++) & 1 ? flush(true) : CAPDEC_OK;
    };
    // LayerNorm backward: dx (+ the dropout-masked copy the branch products read) on the critical path, the parameter
    // gradients (weight gradients: nothing later in backward reads them) from private partials on the lane
    auto ln_bwd = [&](const void* dyv, int s_x, const void* gamma, const void* beta, int s_mean, int s_rstd, int s_dx, int s_dxm,
                      uint64_t seed2, int s_part, void* dgamma, void* dbeta) -> int {
        if (s16)
            STEP(capdec_layernorm_bwd_dx16(dyv, at(s_x), gamma, beta, at(s_mean), at(s_rstd), at(s_dx), at(s_dxm), ld(s_dxm), at(s_part),
                                           L.s[s_part].bytes, M, d, CAPDEC_ACT_NONE, 0.f, 0, p, seed2, stream));
        else
            STEP(capdec_layernorm_bwd_dx(dyv, at(s_x), gamma, beta, at(s_mean), at(s_rstd), at(s_dx), tr ? at(s_dxm) : nullptr,
                                         at(s_part), L.s[s_part].bytes, M, d, CAPDEC_ACT_NONE, 0.f, 0, p, seed2, stream));
        fin[nfin++] = LNFin{at(s_part), dgamma, dbeta};
        return CAPDEC_OK;
    };
    auto masked = [&](int s_dx, int s_dxm) { return (s16 || tr) ? Mat{at(s_dxm), ld(s_dxm)} : Mat{at(s_dx), d}; };
    auto act16 = [&](int s32, int s16slot) { return s16 ? Mat{at(s16slot), ld(s16slot)} : Mat{at(s32), d}; };

    // ---- feed-forward block
    STEP(ln_bwd(dy, SL_S3, P->ln3_w, P->ln3_b, SL_MEAN3, SL_RSTD3, SL_G3, SL_G3M, seed(S_FFN_OUT), SL_PART3, P->g_ln3_w, P->g_ln3_b));
    const Mat g3m = masked(SL_G3, SL_G3M);
    STEP(wgrad(g3m, Mat{at(SL_H), ld(SL_H)}, d, dff, M, P->g_w2, P->g_b2));
    capdec_epilogue ep = ep0();
    ep.aux = at(SL_U); ep.act = CAPDEC_ACT_GELU_BWD; ep.drop_p = p; ep.drop_seed = seed(S_FFN_IN);
    STEP(G.run(0, 0, M, dff, d, g3m, W(P->w2, P->w2_16, P->ld_w2_16, dff), at(SL_DU), ld(SL_DU), s16, ep));
    STEP(wgrad(Mat{at(SL_DU), ld(SL_DU)}, act16(SL_X2, SL_X2_16), dff, d, M, P->g_w1, P->g_b1));
    ep = ep0();
    ep.addend = at(SL_G3); ep.ld_addend = d;
    STEP(G.run(0, 0, M, d, dff, Mat{at(SL_DU), ld(SL_DU)}, W(P->w1, P->w1_16, P->ld_w1_16, d), at(SL_DX2), d, false, ep));
    // ---- cross-attention block
    STEP(ln_bwd(at(SL_DX2), SL_S2, P->ln2_w, P->ln2_b, SL_MEAN2, SL_RSTD2, SL_G2, SL_G2M, seed(S_CA_OUT), SL_PART2, P->g_ln2_w,
                P->g_ln2_b));
    const Mat g2m = masked(SL_G2, SL_G2M);
    STEP(wgrad(g2m, Mat{at(SL_C), ld(SL_C)}, d, d, M, P->g_wc_out, P->g_bc_out));
    STEP(G.run(0, 0, M, d, d, g2m, W(P->wc_out, P->wc_out16, P->ld_wc_out16, d), at(SL_DC), ld(SL_DC), s16, ep0()));
    STEP(attn_bwd(at(SL_Q), ld(SL_Q), kv, ldkv, col(kv, s16, d), ldkv, at(SL_C), ld(SL_C), at(SL_DC), ld(SL_DC), at(SL_LSE2),
                  at(SL_DELTA), at(SL_DQ), ld(SL_DQ), dkv, lddkv, col(dkv, s16, d), lddkv, B, H, T, S, dh, 0, scale, p, seed(S_CA_P),
                  stream));
    STEP(wgrad(Mat{at(SL_DQ), ld(SL_DQ)}, act16(SL_X1, SL_X1_16), d, d, M, P->g_wc_in, P->g_bc_in));
    STEP(wgrad(Mat{dkv, lddkv}, s16 ? Mat{mem16, ldmem16} : Mat{mem, d}, 2 * d, d, Ms, (float*)P->g_wc_in + (size_t)d * d,
               (float*)P->g_bc_in + d));
    ep = ep0();
    ep.addend = at(SL_G2); ep.ld_addend = d;
    STEP(G.run(0, 0, M, d, d, Mat{at(SL_DQ), ld(SL_DQ)}, W(P->wc_in, P->wcq16, P->ld_wcq16, d), at(SL_DX1), d, false, ep));
    // ---- self-attention block
    STEP(ln_bwd(at(SL_DX1), SL_S1, P->ln1_w, P->ln1_b, SL_MEAN1, SL_RSTD1, SL_G1, SL_G1M, seed(S_SA_OUT), SL_PART1, P->g_ln1_w,
                P->g_ln1_b));
    const Mat g1m = masked(SL_G1, SL_G1M);
    STEP(wgrad(g1m, Mat{at(SL_A), ld(SL_A)}, d, d, M, P->g_w_out, P->g_b_out));
    STEP(G.run(0, 0, M, d, d, g1m, W(P->w_out, P->w_out16, P->ld_w_out16, d), at(SL_DA), ld(SL_DA), s16, ep0()));
    void* qkv = at(SL_QKV);
    void* dqkv = at(SL_DQKV);
    STEP(attn_bwd(qkv, ld(SL_QKV), col(qkv, s16, d), ld(SL_QKV), col(qkv, s16, 2 * d), ld(SL_QKV), at(SL_A), ld(SL_A), at(SL_DA),
                  ld(SL_DA), at(SL_LSE1), at(SL_DELTA), dqkv, ld(SL_DQKV), col(dqkv, s16, d), ld(SL_DQKV), col(dqkv, s16, 2 * d),
                  ld(SL_DQKV), B, H, T, T, dh, g->causal, scale, p, seed(S_SA_P), stream));
    STEP(wgrad(Mat{dqkv, ld(SL_DQKV)}, s16 ? Mat{x16, ldx16} : Mat{x, d}, 3 * d, d, M, P->g_w_in, P->g_b_in));
    STEP(flush(false));
    ep = ep0();
    ep.addend = at(SL_G1); ep.ld_addend = d;
    STEP(G.run(0, 0, M, d, 3 * d, Mat{dqkv, ld(SL_DQKV)}, W(P->w_in, P->w_in16, P->ld_w_in16, d), at(SL_DX), d, false, ep));
    return CAPDEC_OK;
}
#undef STEP
