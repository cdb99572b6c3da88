"""Oracle #1: the reference's decoder half, rebuilt from the same stock ``torch.nn`` classes.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Runs on CPU in fp32.

Every block cites the reference line it restates (paths relative to /root/reference):

* PositionalEncoding            models.py:6-19
* feature_projection            models.py:37-44
* embedding / decoder / head    models.py:47-73
* xavier init                   models.py:78-82
* causal mask                   models.py:84-88
* training forward              models.py:98-124 (ViT at :98 replaced by given features)
* label-smoothed loss           train.py:41-50
* shift                         train.py:29-30
* clip                          train.py:56
* sampling decode loop          train.py:289-412 (prefix recompute at :312-332)

Greedy and beam decoding do not exist in the reference; the definitions
here are the build's own and are the oracle for them ("parity unpinned" by the reference).
"""
import math

import torch
import torch.nn as nn

PAD, START, END, UNK, PERIOD = 0, 1, 2, 3, 4  # dataset.py:112-116


class OraclePositionalEncoding(nn.Module):
    """models.py:6-19."""

    def __init__(self, d_model, dropout=0.1, max_len=5000):
        super().__init__()
        self.dropout = nn.Dropout(p=dropout)
        position = torch.arange(max_len).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
        pe = torch.zeros(max_len, 1, d_model)
        pe[:, 0, 0::2] = torch.sin(position * div_term)
        pe[:, 0, 1::2] = torch.cos(position * div_term)
        self.register_buffer("pe", pe)

    def forward(self, x):  # x: [T, B, d] (seq-first, models.py:112-114)
        return self.dropout(x + self.pe[: x.size(0)])


class OracleCaptionDecoder(nn.Module):
    """The reference ``ImageCaptioningModel`` minus the ViT (models.py:21-130).

    ``feature_dim`` generalises the literal 768 of models.py:38 so the synthetic grid configs of
    BASELINE.json (2048-d 7x7, 512-d 14x14) run through the same operator.  ``features`` may be
    ``[B, Din]`` (reference: memory length 1, models.py:104) or ``[B, S, Din]``.
    """

    def __init__(self, vocab_size, d_model=512, nhead=8, num_decoder_layers=6, feature_dim=768):
        super().__init__()
        self.feature_projection = nn.Sequential(  # models.py:37-44
            nn.Linear(feature_dim, d_model * 2),
            nn.LayerNorm(d_model * 2),
            nn.GELU(),
            nn.Dropout(0.1),
            nn.Linear(d_model * 2, d_model),
            nn.LayerNorm(d_model),
        )
        self.embedding = nn.Embedding(vocab_size, d_model)  # models.py:47
        self.pos_encoder = OraclePositionalEncoding(d_model)  # models.py:50
        decoder_layer = nn.TransformerDecoderLayer(  # models.py:53-60
            d_model=d_model, nhead=nhead, dim_feedforward=d_model * 4, dropout=0.1,
            activation="gelu", batch_first=True)
        self.transformer_decoder = nn.TransformerDecoder(  # models.py:61-65
            decoder_layer, num_layers=num_decoder_layers, norm=nn.LayerNorm(d_model))
        self.output_layer = nn.Sequential(  # models.py:68-73
            nn.Linear(d_model, d_model * 2), nn.GELU(), nn.Dropout(0.1),
            nn.Linear(d_model * 2, vocab_size))
        for p in self.parameters():  # models.py:78-82
            if p.dim() > 1 and p.requires_grad:
                nn.init.xavier_uniform_(p)

    @staticmethod
    def generate_square_subsequent_mask(sz):  # models.py:84-88
        mask = (torch.triu(torch.ones(sz, sz)) == 1).transpose(0, 1)
        return mask.float().masked_fill(mask == 0, float("-inf")).masked_fill(mask == 1, 0.0)

    def project(self, features):
        f = self.feature_projection(features)  # models.py:101
        return f.unsqueeze(1) if f.dim() == 2 else f  # models.py:104

    def decode(self, memory, captions):
        tgt = self.embedding(captions)  # models.py:109
        tgt = self.pos_encoder(tgt.transpose(0, 1)).transpose(0, 1)  # models.py:112-114
        mask = self.generate_square_subsequent_mask(captions.size(1))  # train.py:33
        out = self.transformer_decoder(tgt, memory, tgt_mask=mask)  # models.py:117-121
        return self.output_layer(out)  # models.py:124

    def forward(self, features, captions):
        return self.decode(self.project(features), captions)


def label_smoothed_loss(logits, targets, smooth_factor=0.1):
    """train.py:41-50, verbatim arithmetic: dense smoothed one-hot, log_softmax, mean over ALL B*T."""
    n_classes = logits.size(-1)
    one_hot = torch.zeros_like(logits).scatter(2, targets.unsqueeze(-1), 1)
    one_hot = one_hot * (1 - smooth_factor) + (smooth_factor / n_classes)
    log_prb = torch.log_softmax(logits, dim=-1)
    return -(one_hot * log_prb).sum(dim=-1).mean()


def train_step(model, optimizer, features, captions, grad_clip=1.0):
    """One iteration of train.py:24-61 (ViT replaced by given features).  Returns (loss, total_norm, logits)."""
    tgt_input, tgt_output = captions[:, :-1], captions[:, 1:]  # train.py:29-30
    optimizer.zero_grad()
    logits = model(features, tgt_input)
    loss = label_smoothed_loss(logits, tgt_output)
    loss.backward()
    total_norm = torch.nn.utils.clip_grad_norm_(model.parameters(), grad_clip)  # train.py:56
    optimizer.step()
    return loss.detach(), total_norm.detach(), logits.detach()


@torch.no_grad()
def step_logprobs(model, memory, prefix):
    """Log-softmax of the last position after re-running the whole prefix, as train.py:312-340 does."""
    return torch.log_softmax(model.decode(memory, prefix)[:, -1, :].double(), dim=-1)


@torch.no_grad()
def greedy_decode(model, features, max_len=20):
    """Build-defined greedy decode (no reference counterpart).

    START first; each step takes argmax of the last-position logits (ties -> lowest id); a row that
    has emitted END keeps emitting PAD.  Returns ids [B, max_len] (without START) and the minimum
    top-1/top-2 logit gap seen on live rows.
    """
    model.eval()
    memory = model.project(features)
    B = features.size(0)
    prefix = torch.full((B, 1), START, dtype=torch.long)
    done = torch.zeros(B, dtype=torch.bool)
    out = torch.zeros(B, max_len, dtype=torch.long)
    min_gap = float("inf")
    for t in range(max_len):
        logits = model.decode(memory, prefix)[:, -1, :]
        top2 = logits.topk(2, dim=-1).values
        live = ~done
        if live.any():
            min_gap = min(min_gap, float((top2[live, 0] - top2[live, 1]).min()))
        nxt = logits.argmax(dim=-1)
        nxt = torch.where(done, torch.full_like(nxt, PAD), nxt)
        out[:, t] = nxt
        done |= nxt == END
        prefix = torch.cat([prefix, nxt.unsqueeze(1)], dim=1)
    return out, min_gap


@torch.no_grad()
def beam_decode(model, features, beam=3, max_len=20):
    """Build-defined beam search (no reference counterpart).

    Scores are sums of log-softmax (float64 accumulate of fp32 logits is NOT used: fp32 sums, as
    the device kernel does).  A beam that has emitted END is frozen: it proposes exactly one
    continuation (PAD, score unchanged).  Candidates are ranked by (score desc, flat index asc)
    where flat = beam*V + token.  No length normalisation.  Returns ids [B, beam, max_len],
    scores [B, beam], and the minimum gap between the last kept and first rejected candidate.
    """
    model.eval()
    memory = model.project(features)
    B, S, d = memory.shape
    V = model.embedding.num_embeddings
    NEG = -1.0e30
    seqs = torch.full((B, beam, 1), START, dtype=torch.long)
    scores = torch.full((B, beam), NEG, dtype=torch.float32)
    scores[:, 0] = 0.0
    finished = torch.zeros(B, beam, dtype=torch.bool)
    min_gap = float("inf")
    mem_rep = memory.unsqueeze(1).expand(B, beam, S, d).reshape(B * beam, S, d)
    for t in range(max_len):
        logits = model.decode(mem_rep, seqs.reshape(B * beam, -1))[:, -1, :]
        logp = torch.log_softmax(logits, dim=-1).reshape(B, beam, V)
        cand = scores.unsqueeze(-1) + logp
        frozen = torch.full((V,), NEG)
        frozen[PAD] = 0.0
        cand = torch.where(finished.unsqueeze(-1), scores.unsqueeze(-1) + frozen, cand)
        flat = cand.reshape(B, beam * V)
        # stable sort -> ties resolved to the lowest flat index
        order = torch.sort(flat, dim=-1, descending=True, stable=True)
        top_s, top_i = order.values[:, : beam + 1], order.indices[:, : beam + 1]
        gap = top_s[:, beam - 1] - top_s[:, beam]
        valid = top_s[:, beam] > NEG / 2
        if valid.any():
            min_gap = min(min_gap, float(gap[valid].min()))
        top_s, top_i = top_s[:, :beam], top_i[:, :beam]
        parent, tok = top_i // V, top_i % V
        seqs = torch.cat([torch.gather(seqs, 1, parent.unsqueeze(-1).expand(-1, -1, seqs.size(2))),
                          tok.unsqueeze(-1)], dim=2)
        finished = torch.gather(finished, 1, parent) | (tok == END)
        scores = top_s
    return seqs[:, :, 1:], scores, min_gap


def synthetic_captions(B, T, V, gen, min_len=8):
    """SURVEY.md section 8d synthetic ids: START, uniform [5,V) words, END at a length ~U[min_len,T], PAD."""
    ids = torch.randint(5, V, (B, T + 1), generator=gen)
    ids[:, 0] = START
    lens = torch.randint(min(min_len, T), T + 1, (B,), generator=gen)
    pos = torch.arange(T + 1).unsqueeze(0)
    ids = torch.where(pos == lens.unsqueeze(1), torch.full_like(ids, END), ids)
    ids = torch.where(pos > lens.unsqueeze(1), torch.full_like(ids, PAD), ids)
    return ids


# ---------------------------------------------------------------------------------------------
# Reference sampler: train.py:289-412 restated with the ONE
# non-reproducible call, torch.multinomial(top_probs) at train.py:367, replaced by inverse-CDF
# sampling on caller-supplied uniforms u[i, b] (same distribution: index j with probability
# top_probs[j] / sum(top_probs)).  Everything else follows the reference line by line, including
# its quirks: rows keep sampling after they emit END, the loop stops only when ALL rows emit END in
# the same step or ALL rows reached max_sentences, the last two steps force END.
def _mix32(x):
    x &= 0xFFFFFFFF
    x ^= x >> 16
    x = (x * 0x7FEB352D) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x846CA68B) & 0xFFFFFFFF
    x ^= x >> 16
    return x


def sampler_uniforms(seed, steps, rows):
    """The device sampler's counter-based uniforms (csrc/decode.hip sample_select_kernel):
    u[i, b] = (mix32(mix32(b + lo32(seed)) ^ hi32(seed) ^ (i * 0x9E3779B9)) >> 8) / 2^24."""
    lo, hi = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    u = torch.empty(steps, rows, dtype=torch.float32)
    for i in range(steps):
        for b in range(rows):
            h = _mix32(_mix32((b + lo) & 0xFFFFFFFF) ^ hi ^ ((i * 0x9E3779B9) & 0xFFFFFFFF))
            u[i, b] = (h >> 8) / 16777216.0
    return u


@torch.no_grad()
def reference_sampler(model, features, uniforms, max_len=200, max_sentences=10, min_words_per_sentence=5):
    """train.py:289-412 on projected features.  Returns (ids [B, L] int64 0-padded, min_margin) where
    min_margin is the smallest relative distance of a draw from a CDF boundary (a draw that close to a
    boundary may legitimately flip under 1-ulp differences of exp)."""
    model.eval()
    memory = model.project(features)  # train.py:294-300
    B = features.size(0)
    generated = torch.full((B, 1), 1, dtype=torch.long)  # train.py:304
    words_since_period = torch.zeros(B, dtype=torch.long)  # train.py:309
    sentences_generated = torch.zeros(B, dtype=torch.long)  # train.py:310
    min_margin = float("inf")
    for i in range(max_len - 1):  # train.py:312
        output = model.decode(memory, generated)  # train.py:314-332
        if i >= max_len - 3:  # train.py:335-336
            next_token = torch.full((B, 1), 2, dtype=torch.long)
        else:
            temperature = 0.7
            logits = output[:, -1:] / temperature  # train.py:340
            for b in range(B):  # train.py:343-360
                if words_since_period[b] < min_words_per_sentence:
                    logits[b, :, 4] = float("-inf")
                elif words_since_period[b] >= min_words_per_sentence:
                    logits[b, :, 4] += 1.0
                if sentences_generated[b] >= max_sentences - 1 and words_since_period[b] >= min_words_per_sentence:
                    logits[b, :, 2] += 2.0
                if i >= max_len * 0.9:
                    logits[b, :, 4] += 2.0
                    if words_since_period[b] >= min_words_per_sentence:
                        logits[b, :, 2] += 3.0
            # train.py:363-364: topk of softmax == the 5 largest logits (ties -> lowest index, stable sort);
            # the normaliser cancels in the multinomial, so weights exp(l - max) stand in for top_probs
            z = logits[:, 0, :]
            order = torch.sort(z, dim=-1, descending=True, stable=True).indices[:, :5]
            top = z.gather(1, order)
            w = torch.exp(top - top[:, :1])
            next_token = torch.empty(B, 1, dtype=torch.long)
            for b in range(B):  # train.py:367-368 with inverse-CDF in place of multinomial
                total = w[b, 0]
                for j in range(1, 5):
                    total = total + w[b, j]
                target = uniforms[i, b] * total
                cum = torch.zeros((), dtype=torch.float32)
                pick = 4
                for j in range(5):
                    cum = cum + w[b, j]
                    if j < 4:
                        min_margin = min(min_margin, abs(float(cum - target)) / float(total))
                    if cum > target:
                        pick = j
                        break
                next_token[b, 0] = order[b, pick]
        generated = torch.cat([generated, next_token], dim=1)  # train.py:371
        for b in range(B):  # train.py:374-380
            token_id = next_token[b].item()
            if token_id == 4:
                sentences_generated[b] += 1
                words_since_period[b] = 0
            elif token_id not in [0, 1, 2, 3, 4]:
                words_since_period[b] += 1
        if (next_token == 2).all() or (sentences_generated >= max_sentences).all():  # train.py:383
            break
    final = []
    for b in range(B):  # train.py:387-396
        seq = generated[b]
        if seq[-1] != 2:
            if words_since_period[b] > 0:
                seq = torch.cat([seq, torch.tensor([4])])
            seq = torch.cat([seq, torch.tensor([2])])
        final.append(seq)
    L = max(s.size(0) for s in final)  # train.py:399-410
    out = torch.zeros(B, L, dtype=torch.long)
    for b, s in enumerate(final):
        out[b, :s.size(0)] = s
    return out, min_margin
