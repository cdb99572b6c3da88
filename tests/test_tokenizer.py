"""Tokenizer / vocabulary restatement (oracle/tokenizer.py) against the reference's own fixture: the
108-word vocabulary of data/cache/*_vocab_full_v1.pkl, committed as tests/golden/vocab.json."""
import json
import os
import pickle

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import tokenizer


def _vocab():
    with open(os.path.join(GOLDEN, "vocab.json")) as f:
        words = json.load(f)
    return {w: i for i, w in enumerate(words)}


def test_vocab_fixture_shape():
    v = _vocab()
    assert len(v) == 108
    assert [v[t] for t in tokenizer.SPECIALS] == [0, 1, 2, 3, 4]  # ref dataset.py:164-168
    words = list(v)[5:]
    assert words == sorted(words)  # ref dataset.py:156


def test_clean_and_encode():
    v = _vocab()
    assert tokenizer.clean_text("The Upper-clothing has LONG sleeves, cotton fabric.") == \
        "the upper clothing has long sleeves cotton fabric ."
    ids = tokenizer.encode("The sleeves. Zzzunknown!", v, 12)
    assert ids[0] == 1 and ids[-1] == 0 and len(ids) == 12
    assert ids[1:6] == [v["the"], v["sleeves"], 4, 3, 2]
    long = tokenizer.encode("the " * 50, v, 10)  # truncation keeps END last (ref dataset.py:223-224)
    assert len(long) == 10 and long[-1] == 2 and long[0] == 1


@pytest.mark.parametrize("name,count,tokens", [("train", 10155, 599160), ("test", 2538, 150476)])
def test_caption_fixture(name, count, tokens):
    d = np.load(os.path.join(GOLDEN, f"captions_{name}.npz"))
    assert len(d["lengths"]) == count and len(d["tokens"]) == tokens == int(d["lengths"].sum())
    assert d["tokens"].max() < 108
    ends = np.cumsum(d["lengths"].astype(np.int64))
    starts = ends - d["lengths"]
    assert (d["tokens"][starts] == 1).all() and (d["tokens"][ends - 1] == 2).all()
    assert d["lengths"].max() <= 300


def test_unpickler_executes_nothing(tmp_path):
    good = tmp_path / "v.pkl"
    with open(good, "wb") as f:
        pickle.dump({"vocab": {"<PAD>": 0, "a": 5}, "version": "v1", "timestamp": None}, f)
    assert tokenizer.read_vocab_pickle_without_executing(good)["vocab"] == {"<PAD>": 0, "a": 5}
    bad = tmp_path / "b.pkl"
    with open(bad, "wb") as f:
        pickle.dump({"vocab": os.getcwd}, f)  # GLOBAL / STACK_GLOBAL opcode
    with pytest.raises(ValueError):
        tokenizer.read_vocab_pickle_without_executing(bad)


@pytest.mark.skipif(not os.path.exists("/root/reference/data/train_captions.json"), reason="build container only")
def test_rebuild_from_reference_json_matches_fixture():
    with open("/root/reference/data/train_captions.json") as f:
        caps = json.load(f)
    v = tokenizer.build_vocab(caps.values())
    assert v == _vocab()
    ref = tokenizer.read_vocab_pickle_without_executing("/root/reference/data/cache/train_captions_vocab_full_v1.pkl")
    assert list(ref["vocab"].items()) == list(v.items())
