"""Product-side tokenizer / feature cache (imagecaption_amd/data.py) against the reference's vocabulary
fixture, the committed tokenised captions and the oracle restatement of ref dataset.py."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from imagecaption_amd.data import CachedFeatureCaptionDataset, CaptionTokenizer, FeatureCache, build_feature_cache
from oracle import tokenizer as oracle_tok


def _tok():
    with open(os.path.join(GOLDEN, "vocab.json")) as f:
        return CaptionTokenizer(json.load(f))


def test_vocabulary_rebuilt_from_decoded_captions_is_the_reference_fixture():
    tok = _tok()
    d = np.load(os.path.join(GOLDEN, "captions_train.npz"))
    ends = np.cumsum(d["lengths"].astype(np.int64))
    caps = [tok.decode(d["tokens"][e - n:e]) for e, n in list(zip(ends, d["lengths"]))[:3000]]
    rebuilt = CaptionTokenizer.from_captions(caps)
    # every word of the 3000 captions keeps its relative (sorted) order; specials are ids 0-4
    assert [rebuilt.idx2word[i] for i in range(5)] == ["<PAD>", "<START>", "<END>", "<UNK>", "."]
    words = [rebuilt.idx2word[i] for i in range(5, len(rebuilt))]
    assert words == sorted(words) and set(words) <= set(tok.vocab)


def test_encode_round_trips_the_committed_token_streams():
    tok = _tok()
    d = np.load(os.path.join(GOLDEN, "captions_test.npz"))
    ends = np.cumsum(d["lengths"].astype(np.int64))
    for e, n in list(zip(ends, d["lengths"]))[:500]:
        ids = d["tokens"][e - n:e].astype(np.int64).tolist()
        again = tok.encode(tok.decode(ids), 300)
        assert again[:n] == ids and again[n:] == [0] * (300 - n)


@pytest.mark.parametrize("text", ["The Upper-clothing has LONG sleeves, cotton fabric.", "a.b..c", "", "   ",
                                  "Zzz unknown words! 123 under_score", "x " * 400, "no period at the end"])
@pytest.mark.parametrize("max_length", [8, 300])
def test_matches_oracle_restatement(text, max_length):
    tok = _tok()
    assert tok.words_of(text) == oracle_tok.tokenize(text)
    assert tok.encode(text, max_length) == oracle_tok.encode(text, tok.vocab, max_length)
    assert CaptionTokenizer.from_captions([text, "b a ."]).vocab == oracle_tok.build_vocab([text, "b a ."])


def test_bad_vocabularies_are_rejected(tmp_path):
    with pytest.raises(ValueError):
        CaptionTokenizer(["<PAD>", "<START>", "<END>", ".", "<UNK>"])
    with pytest.raises(ValueError):
        CaptionTokenizer(["<PAD>", "<START>", "<END>", "<UNK>", ".", "a", "a"])
    tok = _tok()
    tok.save(str(tmp_path / "v.json"))
    assert CaptionTokenizer.load(str(tmp_path / "v.json")).vocab == tok.vocab


@pytest.mark.parametrize("shape,dtype", [((768,), "float32"), ((5, 24), "float16")])
def test_feature_cache_round_trip_and_dataset_contract(tmp_path, shape, dtype):
    names = [f"img_{i:03d}.jpg" for i in range(11)]
    g = torch.Generator().manual_seed(0)
    images = [torch.randn((n,) + shape, generator=g) for n in (4, 4, 3)]
    cache = build_feature_cache(str(tmp_path / "feat"), names, images, lambda x: x * 2, shape, dtype, source="test")
    cache = FeatureCache.open(str(tmp_path / "feat"))
    full = torch.cat(images) * 2
    tol = 0 if dtype == "float32" else 2e-3
    assert (cache.rows([10, 0, 3, 3]) - full[[10, 0, 3, 3]]).abs().max() <= tol * full.abs().max()
    assert (cache.get("img_007.jpg") - full[7]).abs().max() <= tol * full.abs().max()
    tok = _tok()
    ann = {n: "the sleeves are long . cotton fabric" for n in names[:6]}
    ds = CachedFeatureCaptionDataset(ann, cache, tok, max_length=300)
    f, ids = ds[2]  # ref dataset.py:185-235: (image-slot tensor, LongTensor[max_length])
    assert f.shape == shape and ids.shape == (300,) and ids.dtype == torch.long and ids[0] == 1
    fb, ib = ds.batch([5, 1])
    assert fb.shape == (2,) + shape and ib.shape == (2, 300) and torch.equal(ib[0], ids)
    with pytest.raises(KeyError):
        CachedFeatureCaptionDataset({"nope.jpg": "x"}, cache, tok)
    with pytest.raises(ValueError):
        cache2 = FeatureCache.open(str(tmp_path / "feat"), writable=True)
        cache2.put(0, torch.zeros((2, 3)))
