"""BLEU restatement (imagecaption_amd/bleu.py) on hand-computed cases (nltk parity is unpinned: not installable)."""
import math

from imagecaption_amd.bleu import bleu_1_4, caption_tokens, corpus_bleu


def test_perfect_and_disjoint():
    ref = "the upper clothing has long sleeves .".split()
    assert abs(corpus_bleu([[ref]], [ref]) - 1.0) < 1e-12
    assert corpus_bleu([[ref]], ["x y z w".split()]) == 0.0


def test_hand_computed_with_method1_and_brevity_penalty():
    ref = "a b c d e f".split()
    hyp = "a b c x".split()
    # p1 = 3/4, p2 = 2/3, p3 = 1/2, p4 = 0 -> eps/1 = 0.1 ; BP = exp(1 - 6/4)
    want = math.exp(1 - 6 / 4) * math.exp(0.25 * (math.log(3 / 4) + math.log(2 / 3) + math.log(1 / 2) + math.log(0.1)))
    assert abs(corpus_bleu([[ref]], [hyp]) - want) < 1e-12
    assert abs(corpus_bleu([[ref]], [hyp], (1.0, 0, 0, 0)) - math.exp(1 - 6 / 4) * 0.75) < 1e-12


def test_corpus_level_pooling_and_clipping():
    refs = [["a a b".split()], ["c d".split()]]
    hyps = ["a a a".split(), "c d".split()]
    # unigrams: clipped 2 + 2 of 3 + 2 ; bigrams: (a a) 1 of 2, (c d) 1 of 1 -> 2/3 ; hyp_len = ref_len = 5
    want = math.exp(0.5 * (math.log(4 / 5) + math.log(2 / 3)))
    assert abs(corpus_bleu(refs, hyps, (0.5, 0.5)) - want) < 1e-12


def test_reference_token_filter():
    idx2word = {0: "<PAD>", 1: "<START>", 2: "<END>", 3: "<UNK>", 4: ".", 5: "shirt", 6: "long"}
    assert caption_tokens([1, 6, 5, 4, 2, 0, 0], idx2word) == ["long", "shirt", "."]
    assert caption_tokens([1, 2, 0], idx2word) == ["<unk>"]
    b1, b4 = bleu_1_4([[1, 6, 5, 4, 2]], [[6, 5, 4, 2, 0]], idx2word)
    assert abs(b1 - 1.0) < 1e-12 and b4 > 0
