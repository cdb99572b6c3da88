// KV-cached decode step kernels: single-query attention over the cache (self, through a beam
// ancestry table; cross, over the per-image memory K/V shared by all beams), greedy argmax and the
// beam-search top-k / reorder kernel.  See include/capdec.h (capdec_decode_attn, capdec_greedy_select,
// capdec_beam_select).
//
// Beam reorder without moving the cache: K/V of position j always stay in the cache slot of the beam
// row that computed them; anc[r][j] names that slot for the hypothesis now living in row r.  The
// top-k kernel rewrites only the small int tables (ancestry, token history), never the
// [rows, Tmax, 2d] cache -- the reorder costs O(W*T) ints per image instead of O(L*W*T*d) floats.
#include "common.h"

namespace capdec {

struct DecodeAttnParams {
    const float* q; int ldq;      // [R, >= H*dh]
    const float* k; const float* v; int ldkv;  // key row stride (elements)
    float* out; int ldo;
    const int* anc;               // [R, Tmax] or nullptr (cross attention)
    int R, H, Tmax, t, W, S;
    float scale;
};

// Online-softmax state of one (row, head) and its update by one pass of 64/LPK keys.  Every
// operation is spelled with an explicit rounding intrinsic, so the compiler cannot contract it
// differently in the two kernels below: a hypothesis gets the same bits whatever the beam width.
struct RowAttn {
    float m, l;
    float4 acc;
};
template <int LPK>
__device__ __forceinline__ void row_attn_update(RowAttn& st, const float4& q4, const float4& k4, const float4& v4, bool valid) {
    float s = __fadd_rn(__fmaf_rn(q4.x, k4.x, __fmul_rn(q4.y, k4.y)), __fmaf_rn(q4.z, k4.z, __fmul_rn(q4.w, k4.w)));
#pragma unroll
    for (int o = 1; o < LPK; o <<= 1) s = __fadd_rn(s, __shfl_xor(s, o, 64));
    s = valid ? s : -INFINITY;
    float mi = s;
#pragma unroll
    for (int o = LPK; o < 64; o <<= 1) mi = fmaxf(mi, __shfl_xor(mi, o, 64));
    const float mn = fmaxf(st.m, mi);  // finite: key 0 is always valid in the first pass
    const float alpha = __expf(__fsub_rn(st.m, mn));
    const float pj = valid ? __expf(__fsub_rn(s, mn)) : 0.f;
    float ps = pj;
#pragma unroll
    for (int o = LPK; o < 64; o <<= 1) ps = __fadd_rn(ps, __shfl_xor(ps, o, 64));
    st.l = __fmaf_rn(st.l, alpha, ps);
    st.acc.x = __fmaf_rn(st.acc.x, alpha, __fmul_rn(pj, v4.x));
    st.acc.y = __fmaf_rn(st.acc.y, alpha, __fmul_rn(pj, v4.y));
    st.acc.z = __fmaf_rn(st.acc.z, alpha, __fmul_rn(pj, v4.z));
    st.acc.w = __fmaf_rn(st.acc.w, alpha, __fmul_rn(pj, v4.w));
    st.m = mn;
}
template <int LPK>
__device__ __forceinline__ float4 row_attn_finish(RowAttn& st) {
#pragma unroll
    for (int o = LPK; o < 64; o <<= 1) {
        st.acc.x = __fadd_rn(st.acc.x, __shfl_xor(st.acc.x, o, 64));
        st.acc.y = __fadd_rn(st.acc.y, __shfl_xor(st.acc.y, o, 64));
        st.acc.z = __fadd_rn(st.acc.z, __shfl_xor(st.acc.z, o, 64));
        st.acc.w = __fadd_rn(st.acc.w, __shfl_xor(st.acc.w, o, 64));
    }
    const float inv = __fdiv_rn(1.f, st.l);
    return make_float4(__fmul_rn(st.acc.x, inv), __fmul_rn(st.acc.y, inv), __fmul_rn(st.acc.z, inv), __fmul_rn(st.acc.w, inv));
}
__device__ __forceinline__ float4 scaled_q(const float* ptr, float scale) {
    float4 q4 = *reinterpret_cast<const float4*>(ptr);
    return make_float4(__fmul_rn(q4.x, scale), __fmul_rn(q4.y, scale), __fmul_rn(q4.z, scale), __fmul_rn(q4.w, scale));
}

// one wave per (row, head); LPK = DH/4 lanes share one key (float4 each), 64/LPK keys per pass.
template <int DH>
__global__ __launch_bounds__(256) void decode_attn_kernel(DecodeAttnParams p) {
    constexpr int LPK = DH / 4, KPI = 64 / LPK;
    const int lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= p.R * p.H) return;
    const int r = item / p.H, h = item % p.H;
    const int g = lane / LPK, e4 = lane % LPK;
    const int nkeys = p.anc ? p.t + 1 : p.S;
    const float4 q4 = scaled_q(p.q + (size_t)r * p.ldq + h * DH + e4 * 4, p.scale);
    RowAttn st{-INFINITY, 0.f, make_float4(0.f, 0.f, 0.f, 0.f)};
    // UNR passes per trip: their ancestry look-ups, then their K / V loads, are all issued before the first update, so
    // a trip costs one dependent round trip instead of UNR (the loop is latency-bound: one wave walks ~t keys); the
    // updates run in key order as before -- the same arithmetic, bit for bit.
    constexpr int UNR = 4;
    for (int j0 = 0; j0 < nkeys; j0 += KPI * UNR) {
        bool valid[UNR];
        size_t row[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int j = j0 + u * KPI + g;
            valid[u] = j < nkeys;
            row[u] = 0;
            if (valid[u]) {
                if (p.anc) {
                    const int slot = j == p.t ? r : p.anc[(size_t)r * p.Tmax + j];
                    row[u] = (size_t)slot * p.Tmax + j;
                } else {
                    row[u] = (size_t)(r / p.W) * p.S + j;
                }
            }
        }
        float4 k4[UNR], v4[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            k4[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            v4[u] = k4[u];
            if (valid[u]) {
                k4[u] = *reinterpret_cast<const float4*>(p.k + row[u] * p.ldkv + h * DH + e4 * 4);
                v4[u] = *reinterpret_cast<const float4*>(p.v + row[u] * p.ldkv + h * DH + e4 * 4);
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (j0 + u * KPI < nkeys) row_attn_update<LPK>(st, q4, k4[u], v4[u], valid[u]);
        }
    }
    const float4 o4 = row_attn_finish<LPK>(st);
    if (g == 0) *reinterpret_cast<float4*>(p.out + (size_t)r * p.ldo + h * DH + e4 * 4) = o4;
}

// Self attention over the cache for SMALL batches: one workgroup (four waves) per (row, head), the key passes dealt
// round-robin to the waves (each keeps its own online-softmax state) and merged through LDS in wave order.  With a
// few hundred (row, head) items the one-wave kernel above leaves most of the chip idle and walks ~t keys serially
// (20-25 us per call at t ~ 100, a quarter of a small-batch decode step); this one makes a call four times shorter
// in passes.  Deterministic; not bit-identical to the one-wave kernel (the merge reassociates the softmax sums),
// which keeps serving large batches and the cross-attention (where the beams' bit-identity is tested).
template <int DH>
__global__ __launch_bounds__(256) void decode_self_attn_split_kernel(DecodeAttnParams p) {
    constexpr int LPK = DH / 4, KPI = 64 / LPK, UNR = 4;
    __shared__ float sm_m[4], sm_l[4];
    __shared__ __attribute__((aligned(16))) float sm_acc[4][DH];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int item = blockIdx.x;
    const int r = item / p.H, h = item % p.H;
    const int g = lane / LPK, e4 = lane % LPK;
    const int nkeys = p.t + 1;
    const int npass = (nkeys + KPI - 1) / KPI;
    const float4 q4 = scaled_q(p.q + (size_t)r * p.ldq + h * DH + e4 * 4, p.scale);
    RowAttn st{-INFINITY, 0.f, make_float4(0.f, 0.f, 0.f, 0.f)};
    for (int p0 = wave; p0 < npass; p0 += 4 * UNR) {  // this wave's passes: wave, wave + 4, wave + 8, ...
        bool valid[UNR];
        size_t row[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int j = (p0 + 4 * u) * KPI + g;
            valid[u] = j < nkeys;
            row[u] = 0;
            if (valid[u]) {
                const int slot = j == p.t ? r : p.anc[(size_t)r * p.Tmax + j];
                row[u] = (size_t)slot * p.Tmax + j;
            }
        }
        float4 k4[UNR], v4[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            k4[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            v4[u] = k4[u];
            if (valid[u]) {
                k4[u] = *reinterpret_cast<const float4*>(p.k + row[u] * p.ldkv + h * DH + e4 * 4);
                v4[u] = *reinterpret_cast<const float4*>(p.v + row[u] * p.ldkv + h * DH + e4 * 4);
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (p0 + 4 * u < npass) row_attn_update<LPK>(st, q4, k4[u], v4[u], valid[u]);
        }
    }
    // fold the key groups of this wave (as row_attn_finish does, without the division), publish the wave's state
#pragma unroll
    for (int o = LPK; o < 64; o <<= 1) {
        st.acc.x = __fadd_rn(st.acc.x, __shfl_xor(st.acc.x, o, 64));
        st.acc.y = __fadd_rn(st.acc.y, __shfl_xor(st.acc.y, o, 64));
        st.acc.z = __fadd_rn(st.acc.z, __shfl_xor(st.acc.z, o, 64));
        st.acc.w = __fadd_rn(st.acc.w, __shfl_xor(st.acc.w, o, 64));
    }
    if (g == 0) *reinterpret_cast<float4*>(&sm_acc[wave][e4 * 4]) = st.acc;
    if (lane == 0) { sm_m[wave] = st.m; sm_l[wave] = st.l; }
    __syncthreads();
    if (wave == 0 && g == 0) {
        const float M = fmaxf(fmaxf(sm_m[0], sm_m[1]), fmaxf(sm_m[2], sm_m[3]));  // finite: wave 0 always holds key 0
        float L = 0.f;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float f = __expf(__fsub_rn(sm_m[w], M));  // a wave without keys: exp(-inf) = 0
            const float4 x = *reinterpret_cast<const float4*>(&sm_acc[w][e4 * 4]);
            L = __fmaf_rn(sm_l[w], f, L);
            a.x = __fmaf_rn(x.x, f, a.x); a.y = __fmaf_rn(x.y, f, a.y); a.z = __fmaf_rn(x.z, f, a.z); a.w = __fmaf_rn(x.w, f, a.w);
        }
        const float inv = __fdiv_rn(1.f, L);
        *reinterpret_cast<float4*>(p.out + (size_t)r * p.ldo + h * DH + e4 * 4) =
            make_float4(__fmul_rn(a.x, inv), __fmul_rn(a.y, inv), __fmul_rn(a.z, inv), __fmul_rn(a.w, inv));
    }
}

// Cross attention for beams: the NB hypotheses of one image attend the SAME per-image K/V, so one
// wave serves all NB rows of an (image, head) and reads each key / value once (the per-row kernel
// above reads them NB times; L2 absorbs that, but the CU-side delivery rate is what bounds it).
template <int DH, int NB>
__global__ __launch_bounds__(256) void decode_cross_attn_kernel(DecodeAttnParams p) {
    constexpr int LPK = DH / 4, KPI = 64 / LPK;
    const int lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= (p.R / NB) * p.H) return;
    const int n = item / p.H, h = item % p.H;
    const int g = lane / LPK, e4 = lane % LPK;
    float4 q4[NB];
    RowAttn st[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        q4[b] = scaled_q(p.q + (size_t)(n * NB + b) * p.ldq + h * DH + e4 * 4, p.scale);
        st[b] = RowAttn{-INFINITY, 0.f, make_float4(0.f, 0.f, 0.f, 0.f)};
    }
    const float* kb = p.k + (size_t)n * p.S * p.ldkv + h * DH + e4 * 4;
    const float* vb = p.v + (size_t)n * p.S * p.ldkv + h * DH + e4 * 4;
    for (int j0 = 0; j0 < p.S; j0 += KPI) {
        const int j = j0 + g;
        const bool valid = j < p.S;
        float4 k4 = make_float4(0.f, 0.f, 0.f, 0.f), v4 = k4;
        if (valid) {
            k4 = *reinterpret_cast<const float4*>(kb + (size_t)j * p.ldkv);
            v4 = *reinterpret_cast<const float4*>(vb + (size_t)j * p.ldkv);
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) row_attn_update<LPK>(st[b], q4[b], k4, v4, valid);
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const float4 o4 = row_attn_finish<LPK>(st[b]);
        if (g == 0) *reinterpret_cast<float4*>(p.out + (size_t)(n * NB + b) * p.ldo + h * DH + e4 * 4) = o4;
    }
}

// ---- block-wide "best candidate" reduction: higher score wins, ties -> lower index -------------
struct Cand {
    float s;
    int i;
};
__device__ __forceinline__ bool better(const Cand& a, const Cand& b) { return a.s > b.s || (a.s == b.s && a.i < b.i); }
__device__ __forceinline__ Cand wave_best(Cand c) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        Cand d;
        d.s = __shfl_xor(c.s, o, 64);
        d.i = __shfl_xor(c.i, o, 64);
        if (better(d, c)) c = d;
    }
    return c;
}
__device__ __forceinline__ Cand block_best(Cand c, Cand* sh) {  // 256 threads
    c = wave_best(c);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = c;
    __syncthreads();
    Cand b = sh[0];
#pragma unroll
    for (int w = 1; w < 4; ++w)
        if (better(sh[w], b)) b = sh[w];
    return b;
}

constexpr int TOK_PAD = 0, TOK_END = 2;  // ref dataset.py:112-116

// one block per row: token = argmax(logits) (ties -> lowest id); a finished row emits PAD.
__global__ __launch_bounds__(256) void greedy_select_kernel(const float* __restrict__ logits, int ldl, int* __restrict__ done,
                                                             int* __restrict__ seq, int64_t* __restrict__ next_tok,
                                                             float* __restrict__ gap, int V, int Tmax, int t) {
    __shared__ Cand sh[4];
    const int r = blockIdx.x;
    const float* z = logits + (size_t)r * ldl;
    // one pass: per-thread best and second best, then two block-wide tournaments (the second one gives
    // the runner-up for the recorded top-1/top-2 margin)
    Cand c{-INFINITY, 0x7fffffff}, c2{-INFINITY, 0x7fffffff};
    for (int v = threadIdx.x; v < V; v += 256) {
        const Cand d{z[v], v};
        if (better(d, c)) {
            c2 = c;
            c = d;
        } else if (better(d, c2)) {
            c2 = d;
        }
    }
    const Cand b = block_best(c, sh);
    __syncthreads();
    const Cand b2 = block_best(c.i == b.i ? c2 : c, sh);
    if (threadIdx.x == 0) {
        const int was_done = done[r];
        const int tok = was_done ? TOK_PAD : b.i;
        seq[(size_t)r * Tmax + t] = tok;
        next_tok[r] = tok;
        done[r] = was_done | (tok == TOK_END);
        if (gap && !was_done) gap[r] = fminf(gap[r], b.s - b2.s);
    }
}

// one block per image: top-W of {score[w] + log_softmax(logits[w])[v]} over W*V candidates.
// STAGE: each beam's logits row is read from memory ONCE (float4) into an LDS row buffer, and the
// sum-exp and candidate passes run on that copy (the unstaged kernel reads the row three times);
// both variants assign elements to threads identically, so their results are bit-identical.
constexpr int MAXW = 8;
constexpr int DECODE_SPLIT_MAX_ITEMS = 2048;  // (row, head) items up to which the split-key self-attention kernel is used
// LW: length of the per-thread candidate list (>= W + 1); the insertion cost is what bounds this kernel
template <bool STAGE, int LW>
__global__ __launch_bounds__(256) void beam_select_kernel(const float* __restrict__ logits, int ldl,
                                                           const float* __restrict__ score_in, float* __restrict__ score_out,
                                                           const int* __restrict__ fin_in, int* __restrict__ fin_out,
                                                           const int* __restrict__ anc_in, int* __restrict__ anc_out,
                                                           const int* __restrict__ seq_in, int* __restrict__ seq_out,
                                                           int64_t* __restrict__ next_tok, float* __restrict__ gap, int W,
                                                           int V, int Tmax, int t) {
    extern __shared__ __attribute__((aligned(16))) float rowbuf[];  // STAGE: V floats
    __shared__ Cand sh[4];
    __shared__ float red[4];
    __shared__ int row_fin[MAXW];
    __shared__ Cand picked[MAXW + 1];
    const int n = blockIdx.x, tid = threadIdx.x;
    const float NEG = -1.0e30f;
    // per-thread top-LW over its strided share of the W*V candidates; the list length is a
    // compile-time constant so it stays in registers
    Cand loc[LW];
#pragma unroll
    for (int k = 0; k < LW; ++k) loc[k] = Cand{-INFINITY, 0x7fffffff};
    auto offer = [&](float sc, int f) {
        const Cand c{sc, f};
        if (better(c, loc[LW - 1])) {
            loc[LW - 1] = c;
#pragma unroll
            for (int k = LW - 1; k > 0; --k)
                if (better(loc[k], loc[k - 1])) {
                    const Cand tmp = loc[k]; loc[k] = loc[k - 1]; loc[k - 1] = tmp;
                }
        }
    };
    for (int w = 0; w < W; ++w) {
        const int row = n * W + w;
        const float rscore = score_in[row];
        const int rfin = fin_in[row];  // block-uniform
        if (tid == 0) row_fin[w] = rfin;
        if (rfin) {  // a finished beam only continues with PAD, at its own score
            for (int v = tid; v < V; v += 256) offer(rscore + (v == TOK_PAD ? 0.f : NEG), w * V + v);
            continue;
        }
        // log-sum-exp in torch's log_softmax order: max, sum exp(z - max), log
        const float* z = logits + (size_t)row * ldl;
        float mx = -INFINITY;
        if (STAGE) {
            __syncthreads();  // the previous row's readers are done with rowbuf
            const float4* z4 = reinterpret_cast<const float4*>(z);
            float4* b4 = reinterpret_cast<float4*>(rowbuf);
            for (int i = tid; i < (V >> 2); i += 256) {
                const float4 x = z4[i];
                b4[i] = x;
                mx = fmaxf(fmaxf(mx, fmaxf(x.x, x.y)), fmaxf(x.z, x.w));
            }
            z = rowbuf;
        } else {
            for (int v = tid; v < V; v += 256) mx = fmaxf(mx, z[v]);
        }
        mx = wave_max(mx);
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = mx;
        __syncthreads();
        mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        float se = 0.f;
        for (int v = tid; v < V; v += 256) se += __expf(z[v] - mx);
        se = wave_sum(se);
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = se;
        __syncthreads();
        const float logsum = __logf((red[0] + red[1]) + (red[2] + red[3]));
        for (int v = tid; v < V; v += 256) offer(rscore + ((z[v] - mx) - logsum), w * V + v);
    }
    __syncthreads();
    // 3. W+1 rounds of block-wide best: winner pops the head of its local list
    for (int k = 0; k <= W; ++k) {
        const Cand b = block_best(loc[0], sh);
        if (loc[0].i == b.i && loc[0].s == b.s) {
#pragma unroll
            for (int q = 0; q < LW - 1; ++q) loc[q] = loc[q + 1];
            loc[LW - 1] = Cand{-INFINITY, 0x7fffffff};
        }
        if (tid == 0) picked[k] = b;
        __syncthreads();
    }
    // 4. write the new beam state; reorder = rewrite the int tables only
    for (int k = 0; k < W; ++k) {
        const Cand b = picked[k];
        const int pw = b.i / V, tok = b.i - pw * V;
        const int prow = n * W + pw, row = n * W + k;
        for (int j = tid; j < t; j += 256) {
            seq_out[(size_t)row * Tmax + j] = seq_in[(size_t)prow * Tmax + j];
            anc_out[(size_t)row * Tmax + j] = anc_in[(size_t)prow * Tmax + j];
        }
        if (tid == 0) {
            seq_out[(size_t)row * Tmax + t] = tok;
            anc_out[(size_t)row * Tmax + t] = prow;
            score_out[row] = b.s;
            fin_out[row] = row_fin[pw] | (tok == TOK_END);
            next_tok[row] = tok;
        }
    }
    if (tid == 0 && gap && picked[W].s > NEG * 0.5f) gap[n] = fminf(gap[n], picked[W - 1].s - picked[W].s);
}

// ---- the reference's sampling step (ref train.py:335-383) on device --------------------------------
// One block per row.  step i: forced END for the last two steps; otherwise logits / temperature, the
// period (id 4) / END (id 2) biases driven by the per-row sentence counters, top-5 (ties -> lowest id),
// and one draw from the top-5 weights by inverse CDF on a counter-based uniform (the reference's
// torch.multinomial: same distribution, different stream).  Rows keep sampling after END exactly as the
// reference does; the loop-level stop (all rows END in one step, or all rows at max_sentences,
// train.py:383) is recorded in per-step counters that later steps read, so nothing syncs with the host.
struct SampleParams {
    const float* logits; int ldl;
    int* words_since_period; int* sentences; int* seq; int64_t* next_tok;
    int* cnt_end; int* cnt_max;  // [max_len] each, zeroed by the caller before step 0
    int R, V, Tmax, i, max_len, max_sentences, min_words, late;
    float temperature;
    uint32_t s1, s2;
};

constexpr int SAMPLE_K = 5;

__global__ __launch_bounds__(256) void sample_select_kernel(SampleParams p) {
    __shared__ Cand sh[4];
    __shared__ int stop_sh;
    const int r = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) stop_sh = 0;
    __syncthreads();
    for (int j = tid; j < p.i; j += 256)
        if (p.cnt_end[j] == p.R || p.cnt_max[j] == p.R) stop_sh = 1;  // benign race: every writer stores 1
    __syncthreads();
    if (stop_sh) return;  // the reference's loop has already ended (train.py:383-384)

    const int wsp = p.words_since_period[r], sg = p.sentences[r];
    int tok;
    if (p.i >= p.max_len - 3) {  // train.py:335-336
        tok = TOK_END;
    } else {
        const float* z = p.logits + (size_t)r * p.ldl;
        auto biased = [&](int v) -> float {  // train.py:340-360, same order of additions
            float l = z[v] / p.temperature;
            if (v == 4) {
                if (wsp < p.min_words) l = -INFINITY;
                else l += 1.0f;
                if (p.late) l += 2.0f;
            } else if (v == TOK_END) {
                if (sg >= p.max_sentences - 1 && wsp >= p.min_words) l += 2.0f;
                if (p.late && wsp >= p.min_words) l += 3.0f;
            }
            return l;
        };
        // one pass: every thread keeps the best SAMPLE_K of its share (score desc, id asc -- a strict total
        // order, so the result does not depend on which thread sees which element), then SAMPLE_K block-wide
        // tournaments pop the winners.  -inf candidates stay eligible (ties -> lowest id).
        Cand loc[SAMPLE_K];
#pragma unroll
        for (int k = 0; k < SAMPLE_K; ++k) loc[k] = Cand{-INFINITY, 0x7fffffff};
        auto offer = [&](int v) {
            const Cand c{biased(v), v};
            if (better(c, loc[SAMPLE_K - 1])) {
                loc[SAMPLE_K - 1] = c;
#pragma unroll
                for (int k = SAMPLE_K - 1; k > 0; --k)
                    if (better(loc[k], loc[k - 1])) {
                        const Cand tmp = loc[k]; loc[k] = loc[k - 1]; loc[k - 1] = tmp;
                    }
            }
        };
        for (int v = tid; v < p.V; v += 256) offer(v);
        Cand top[SAMPLE_K];
#pragma unroll
        for (int k = 0; k < SAMPLE_K; ++k) {
            const Cand b = block_best(loc[0], sh);
            if (loc[0].i == b.i) {
#pragma unroll
                for (int q = 0; q < SAMPLE_K - 1; ++q) loc[q] = loc[q + 1];
                loc[SAMPLE_K - 1] = Cand{-INFINITY, 0x7fffffff};
            }
            top[k] = b;
            __syncthreads();
        }
        // weights exp(l - l_top) stand in for the top-5 probabilities (the softmax normaliser cancels)
        float w[SAMPLE_K], total = 0.f;
#pragma unroll
        for (int k = 0; k < SAMPLE_K; ++k) {
            w[k] = (top[k].i < p.V) ? expf(top[k].s - top[0].s) : 0.f;
            total = k == 0 ? w[0] : total + w[k];
        }
        const uint32_t h = mix32(mix32((uint32_t)r + p.s1) ^ p.s2 ^ ((uint32_t)p.i * 0x9E3779B9u));
        const float target = ((float)(h >> 8) * (1.0f / 16777216.0f)) * total;
        int pick = SAMPLE_K - 1;
#pragma unroll
        for (int k = SAMPLE_K - 1; k >= 0; --k) {  // first k (ascending) with cum_k > target
            float ck = 0.f;
            for (int q = 0; q <= k; ++q) ck += w[q];
            if (ck > target) pick = k;
        }
        tok = top[pick].i < p.V ? top[pick].i : top[0].i;
    }
    if (tid == 0) {
        p.seq[(size_t)r * p.Tmax + p.i + 1] = tok;
        p.next_tok[r] = tok;
        int nsg = sg;
        if (tok == 4) {  // train.py:374-380
            nsg = sg + 1;
            p.sentences[r] = nsg;
            p.words_since_period[r] = 0;
        } else if (tok > 4) {
            p.words_since_period[r] = wsp + 1;
        }
        if (tok == TOK_END) atomicAdd(&p.cnt_end[p.i], 1);
        if (nsg >= p.max_sentences) atomicAdd(&p.cnt_max[p.i], 1);
    }
}

}  // namespace capdec


using namespace capdec;

extern "C" int capdec_decode_attn(const void* q, int ldq, const void* k, const void* v, int ldkv, void* out, int ldo,
                                  const int* anc, int R, int H, int dh, int Tmax, int t, int W, int S, float scale,
                                  capdec_stream_t stream) {
    CAPDEC_REQUIRE(q && k && v && out, "capdec_decode_attn: null pointer");
    CAPDEC_REQUIRE(R > 0 && H > 0 && (dh == 32 || dh == 64 || dh == 128), "capdec_decode_attn: bad shape R=%d H=%d dh=%d", R, H, dh);
    CAPDEC_REQUIRE(anc ? (Tmax > 0 && t >= 0 && t < Tmax) : (S > 0 && W > 0), "capdec_decode_attn: bad cache geometry");
    CAPDEC_REQUIRE(aligned16(q) && aligned16(k) && aligned16(v) && aligned16(out) && ldq % 4 == 0 && ldkv % 4 == 0 &&
                       ldo % 4 == 0, "capdec_decode_attn: operands must be 16-byte aligned with strides %% 4 == 0");
    DecodeAttnParams p;
    p.q = (const float*)q; p.ldq = ldq; p.k = (const float*)k; p.v = (const float*)v; p.ldkv = ldkv;
    p.out = (float*)out; p.ldo = ldo; p.anc = anc; p.R = R; p.H = H; p.Tmax = Tmax; p.t = t; p.W = W; p.S = S;
    p.scale = scale;
    const dim3 grid((R * H + 3) / 4);
    hipStream_t st = (hipStream_t)stream;
    if (!anc && W > 1 && W <= 8 && R % W == 0) {  // beams of one image share its K/V: one wave per (image, head)
        const dim3 gx(((R / W) * H + 3) / 4);
#define CAPDEC_X(D, NB) hipLaunchKernelGGL((decode_cross_attn_kernel<D, NB>), gx, dim3(256), 0, st, p)
#define CAPDEC_XW(D) \
    switch (W) { case 2: CAPDEC_X(D, 2); break; case 3: CAPDEC_X(D, 3); break; case 4: CAPDEC_X(D, 4); break; \
                 case 5: CAPDEC_X(D, 5); break; case 6: CAPDEC_X(D, 6); break; case 7: CAPDEC_X(D, 7); break; \
                 default: CAPDEC_X(D, 8); break; }
        if (dh == 32) { CAPDEC_XW(32) } else if (dh == 64) { CAPDEC_XW(64) } else { CAPDEC_XW(128) }
#undef CAPDEC_XW
#undef CAPDEC_X
        return check_launch("capdec_decode_attn");
    }
    if (anc && R * H <= DECODE_SPLIT_MAX_ITEMS && t >= 16) {  // small batch, long prefix: four waves per (row, head)
        const dim3 gs(R * H);
        if (dh == 32) hipLaunchKernelGGL(decode_self_attn_split_kernel<32>, gs, dim3(256), 0, st, p);
        else if (dh == 64) hipLaunchKernelGGL(decode_self_attn_split_kernel<64>, gs, dim3(256), 0, st, p);
        else hipLaunchKernelGGL(decode_self_attn_split_kernel<128>, gs, dim3(256), 0, st, p);
        return check_launch("capdec_decode_attn");
    }
    if (dh == 32) hipLaunchKernelGGL(decode_attn_kernel<32>, grid, dim3(256), 0, st, p);
    else if (dh == 64) hipLaunchKernelGGL(decode_attn_kernel<64>, grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL(decode_attn_kernel<128>, grid, dim3(256), 0, st, p);
    return check_launch("capdec_decode_attn");
}

extern "C" int capdec_greedy_select(const void* logits, int ldl, int* done, int* seq, int64_t* next_tok, void* gap, int R,
                                    int V, int Tmax, int t, capdec_stream_t stream) {
    CAPDEC_REQUIRE(logits && done && seq && next_tok && R > 0 && V > 0 && ldl >= V && t >= 0 && t < Tmax,
                   "capdec_greedy_select: bad arguments");
    hipLaunchKernelGGL(greedy_select_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, (const float*)logits, ldl, done,
                       seq, next_tok, (float*)gap, V, Tmax, t);
    return check_launch("capdec_greedy_select");
}

extern "C" int capdec_beam_select(const void* logits, int ldl, const void* score_in, void* score_out, const int* fin_in,
                                  int* fin_out, const int* anc_in, int* anc_out, const int* seq_in, int* seq_out,
                                  int64_t* next_tok, void* gap, int N, int W, int V, int Tmax, int t,
                                  capdec_stream_t stream) {
    CAPDEC_REQUIRE(logits && score_in && score_out && fin_in && fin_out && anc_in && anc_out && seq_in && seq_out &&
                       next_tok, "capdec_beam_select: null pointer");
    CAPDEC_REQUIRE(N > 0 && W >= 1 && W <= MAXW && V >= W + 1 && ldl >= V && t >= 0 && t < Tmax,
                   "capdec_beam_select: bad shape N=%d W=%d (max %d) V=%d t=%d Tmax=%d", N, W, MAXW, V, t, Tmax);
    CAPDEC_REQUIRE((uint64_t)W * V < (1ull << 31), "capdec_beam_select: W*V too large");
    // rows that can be staged in LDS (float4-readable, at most 60 KB: two or more workgroups per CU and no opt-in
    // for large dynamic LDS needed) take the one-read kernel.  Measured for V = 30 000 (tools/beam_select_time.py,
    // 2048 images x 5 beams): staging the 120 KB row (large-LDS opt-in, ONE workgroup per CU) takes 2.31 ms against
    // 1.09 ms for the re-reading kernel, whose three passes run at 3.4 TB/s of logits reads: the lost occupancy costs more
    // than the two saved reads.  Round 3 tried the two remaining ideas on that shape (measured, not kept): the row held
    // in REGISTERS (120 per thread, read once, same element order: bit-identical) took 1.58 ms -- 266 registers, one
    // wave per SIMD, spills --, and pruning the insertion network with a block-wide threshold (the (W+1)-th best
    // candidate so far, raised after every row; nothing below it can win; bit-identical) changed nothing for the
    // re-reading kernel (1.25 vs 1.10 ms): the insertions are not what it waits for.  Such rows stay on the re-reading path
    const size_t lds = (size_t)V * sizeof(float);
    const bool stage = V % 4 == 0 && ldl % 4 == 0 && aligned16(logits) && lds <= 60 * 1024;
#define CAPDEC_BEAM(ST, LW) \
    hipLaunchKernelGGL((beam_select_kernel<ST, LW>), dim3(N), dim3(256), ST ? lds : 0, (hipStream_t)stream, \
                       (const float*)logits, ldl, (const float*)score_in, (float*)score_out, fin_in, fin_out, anc_in, \
                       anc_out, seq_in, seq_out, next_tok, (float*)gap, W, V, Tmax, t)
    if (stage) {
        if (W <= 1) CAPDEC_BEAM(true, 2);
        else if (W <= 3) CAPDEC_BEAM(true, 4);
        else if (W <= 5) CAPDEC_BEAM(true, 6);
        else CAPDEC_BEAM(true, MAXW + 1);
    } else {
        if (W <= 3) CAPDEC_BEAM(false, 4);
        else CAPDEC_BEAM(false, MAXW + 1);
    }
#undef CAPDEC_BEAM
    return check_launch("capdec_beam_select");
}

extern "C" int capdec_sample_select(const void* logits, int ldl, int* words_since_period, int* sentences, int* seq,
                                    int64_t* next_tok, int* cnt_end, int* cnt_max, int R, int V, int Tmax, int i,
                                    int max_len, int max_sentences, int min_words, float temperature, uint64_t seed,
                                    capdec_stream_t stream) {
    CAPDEC_REQUIRE(logits && words_since_period && sentences && seq && next_tok && cnt_end && cnt_max,
                   "capdec_sample_select: null argument");
    CAPDEC_REQUIRE(R > 0 && V > 4 && ldl >= V && i >= 0 && i + 1 < Tmax && max_len <= Tmax && i < max_len - 1 &&
                       temperature > 0.f,
                   "capdec_sample_select: bad shape (R=%d V=%d i=%d Tmax=%d max_len=%d)", R, V, i, Tmax, max_len);
    SampleParams p;
    p.logits = (const float*)logits; p.ldl = ldl;
    p.words_since_period = words_since_period; p.sentences = sentences; p.seq = seq; p.next_tok = next_tok;
    p.cnt_end = cnt_end; p.cnt_max = cnt_max;
    p.R = R; p.V = V; p.Tmax = Tmax; p.i = i; p.max_len = max_len; p.max_sentences = max_sentences; p.min_words = min_words;
    p.late = ((double)i >= (double)max_len * 0.9) ? 1 : 0;  // train.py:357, Python float comparison
    p.temperature = temperature;
    p.s1 = (uint32_t)seed; p.s2 = (uint32_t)(seed >> 32);
    hipLaunchKernelGGL(sample_select_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("capdec_sample_select");
}
