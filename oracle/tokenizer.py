"""Oracle for the caption tokenizer / vocabulary.  TEST INFRASTRUCTURE ONLY.

Restates /root/reference/dataset.py:
* ``clean_text``   dataset.py:118-128
* ``build_vocab``  dataset.py:112-116 (specials) + 130-168 (sorted word set)
* ``encode``       dataset.py:213-228 (START + tokens + END, pad / truncate to max_length)

The reference tokenises with nltk ``word_tokenize`` (dataset.py:150, 217), which is not installed
here.  After ``clean_text`` the text holds only ``[\\w\\s.]`` with every '.' already space-separated,
so a whitespace split yields the same tokens; this is pinned by the reference's own fixture
``data/cache/*_vocab_full_v1.pkl`` (all 108 (word, id) pairs reproduced -- tests/test_tokenizer.py
checks the committed copy of that vocabulary list).
"""
import pickletools
import re

SPECIALS = ["<PAD>", "<START>", "<END>", "<UNK>", "."]


def clean_text(text):
    text = str(text).lower()
    text = re.sub(r"[^\w\s\.]", " ", text)
    text = re.sub(r"\.", " . ", text)
    return re.sub(r"\s+", " ", text).strip()


def tokenize(text):
    return clean_text(text).split()


def build_vocab(captions):
    vocab = {tok: i for i, tok in enumerate(SPECIALS)}
    words = set()
    for c in captions:
        words.update(tokenize(c))
    for w in sorted(words):
        if w not in vocab:
            vocab[w] = len(vocab)
    return vocab


def encode(caption, vocab, max_length):
    ids = [vocab["<START>"]] + [vocab.get(t, vocab["<UNK>"]) for t in tokenize(caption)] + [vocab["<END>"]]
    if len(ids) > max_length:
        ids = ids[: max_length - 1] + [vocab["<END>"]]
    else:
        ids = ids + [vocab["<PAD>"]] * (max_length - len(ids))
    return ids


def read_vocab_pickle_without_executing(path):
    """Rebuild ``{'vocab': {...}, ...}`` from a reference vocab pickle by walking its opcodes.

    Nothing from the file is executed: only the literal-building opcodes listed below are accepted
    (the reference's four pickles use exactly these), anything else raises.
    """
    stack, marks, memo = [], [], []
    with open(path, "rb") as f:
        data = f.read()
    for op, arg, _pos in pickletools.genops(data):
        n = op.name
        if n in ("PROTO", "FRAME", "STOP"):
            continue
        if n == "EMPTY_DICT":
            stack.append({})
        elif n == "MEMOIZE":
            memo.append(stack[-1])
        elif n == "MARK":
            marks.append(len(stack))
        elif n in ("SHORT_BINUNICODE", "BINUNICODE", "BININT", "BININT1", "BININT2"):
            stack.append(arg)
        elif n == "NONE":
            stack.append(None)
        elif n == "BINGET":
            stack.append(memo[arg])
        elif n == "SETITEMS":
            m = marks.pop()
            items = stack[m:]
            del stack[m:]
            for k, v in zip(items[0::2], items[1::2]):
                stack[-1][k] = v
        elif n == "SETITEM":
            v = stack.pop()
            k = stack.pop()
            stack[-1][k] = v
        else:
            raise ValueError(f"refusing pickle opcode {n} in {path}")
    assert len(stack) == 1
    return stack[0]
