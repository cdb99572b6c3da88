"""Corpus BLEU-1 / BLEU-4 with nltk's ``SmoothingFunction().method1``.

Host-side text metric (the acceptance gate "BLEU-4 within 0.1" needs it; nltk is not installed
here).  Restates, from the published definitions, what the reference calls at train.py:203, 239-244:

* ``corpus_bleu(list_of_references, hypotheses, weights, smoothing_function)``: modified n-gram
  precisions with numerators / denominators summed over the corpus (Papineni et al. 2002), brevity
  penalty from the closest reference length, geometric mean under ``weights``;
* ``method1`` (Chen & Cherry 2014): a zero numerator is replaced by epsilon = 0.1;
* nltk returns 0 when there is no unigram overlap at all.

Parity against nltk itself is UNPINNED (it cannot be imported in this environment and the
reference records no BLEU value); the tests check the definition on hand-computed cases.
The reference's token filtering (train.py:221-232: drop ids 0-3, keep '.', empty -> ['<unk>']) is
``caption_tokens``.
"""
import math
from collections import Counter
from fractions import Fraction


def _ngrams(seq, n):
    return Counter(tuple(seq[i:i + n]) for i in range(len(seq) - n + 1))


def _closest_ref_length(refs, hyp_len):
    return min((len(r) for r in refs), key=lambda rl: (abs(rl - hyp_len), rl))


def corpus_bleu(list_of_references, hypotheses, weights=(0.25, 0.25, 0.25, 0.25), epsilon=0.1):
    assert len(list_of_references) == len(hypotheses)
    num, den = Counter(), Counter()
    hyp_len = ref_len = 0
    for refs, hyp in zip(list_of_references, hypotheses):
        for n in range(1, len(weights) + 1):
            counts = _ngrams(hyp, n)
            max_ref = Counter()
            for r in refs:
                for g, c in _ngrams(r, n).items():
                    max_ref[g] = max(max_ref[g], c)
            num[n] += sum(min(c, max_ref[g]) for g, c in counts.items())
            den[n] += max(1, sum(counts.values()))
        hyp_len += len(hyp)
        ref_len += _closest_ref_length(refs, len(hyp))
    if num[1] == 0:
        return 0.0
    p = []
    for n in range(1, len(weights) + 1):
        p.append(Fraction(num[n], den[n]) if num[n] != 0 else Fraction(epsilon).limit_denominator() / den[n])
    bp = 1.0 if hyp_len > ref_len else (math.exp(1 - ref_len / hyp_len) if hyp_len > 0 else 0.0)
    s = sum(w * math.log(float(pi)) for w, pi in zip(weights, p) if w != 0)
    return bp * math.exp(s)


def caption_tokens(ids, idx2word):
    """ref train.py:221-232: words of ids not in {0,1,2,3} ('.' = 4 is kept); empty -> ['<unk>']."""
    toks = [idx2word[int(i)] for i in ids if int(i) not in (0, 1, 2, 3)]
    return toks if toks else ["<unk>"]


def bleu_1_4(reference_ids, hypothesis_ids, idx2word):
    refs = [[caption_tokens(r, idx2word)] for r in reference_ids]
    hyps = [caption_tokens(h, idx2word) for h in hypothesis_ids]
    return (corpus_bleu(refs, hyps, (1.0, 0, 0, 0)), corpus_bleu(refs, hyps, (0.25, 0.25, 0.25, 0.25)))
