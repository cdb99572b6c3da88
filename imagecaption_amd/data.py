"""Host-side data formats either side of the hot path.

* ``CaptionTokenizer`` -- the reference dataset's text pipeline (ref dataset.py:112-168 vocabulary,
  213-228 encoding): lower-case, punctuation other than '.' to spaces, '.' isolated as its own token,
  ``<START> tokens <END>``, unknown words -> ``<UNK>``, pad / truncate to ``max_length`` keeping END
  last.  The reference tokenises with nltk ``word_tokenize``; after its ``_clean_text`` only
  ``[\\w\\s.]`` remains with every '.' space-separated, where a whitespace split is the same thing
  (pinned by the reference's vocabulary fixture, tests/golden/vocab.json).
* ``FeatureCache`` -- a memory-mappable ``[N, D]`` / ``[N, S, D]`` feature file (NumPy ``.npy`` +
  a JSON side-car with the image names) so the decoder path trains from pre-extracted ViT features
  in the ``images`` slot, the way the reference's ``use_cache`` keeps decoded images (dataset.py:170-199).
* ``CachedFeatureCaptionDataset`` -- ``__getitem__ -> (features, caption_ids[max_length])``, the
  contract of ref dataset.py:185-235 with features in place of the image tensor.
"""
import json
import re

import numpy as np
import torch

PAD, START, END, UNK, PERIOD = 0, 1, 2, 3, 4
SPECIAL_TOKENS = ("<PAD>", "<START>", "<END>", "<UNK>", ".")  # ref dataset.py:112-116
_NOT_WORD = re.compile(r"[^\w\s\.]")
_DOT = re.compile(r"\.")


class CaptionTokenizer:
    def __init__(self, words):
        """``words``: the vocabulary in id order (ids 0-4 must be the reference's special tokens)."""
        words = list(words)
        if tuple(words[:5]) != SPECIAL_TOKENS:
            raise ValueError(f"vocabulary must start with {SPECIAL_TOKENS} (ref dataset.py:164-168), got {words[:5]}")
        if len(set(words)) != len(words):
            raise ValueError("duplicate words in the vocabulary")
        self.idx2word = {i: w for i, w in enumerate(words)}
        self.vocab = {w: i for i, w in enumerate(words)}

    def __len__(self):
        return len(self.vocab)

    # ---- text ----------------------------------------------------------------------------------
    @staticmethod
    def words_of(text):
        text = _NOT_WORD.sub(" ", str(text).lower())
        return _DOT.sub(" . ", text).split()

    @classmethod
    def from_captions(cls, captions):
        """Specials first, then every other word in sorted order (ref dataset.py:130-160)."""
        seen = set()
        for c in captions:
            seen.update(cls.words_of(c))
        return cls(list(SPECIAL_TOKENS) + sorted(seen - set(SPECIAL_TOKENS)))

    @classmethod
    def load(cls, path):
        with open(path) as f:
            data = json.load(f)
        return cls(data["words"] if isinstance(data, dict) else data)

    def save(self, path):
        with open(path, "w") as f:
            json.dump({"version": 1, "words": [self.idx2word[i] for i in range(len(self))]}, f)

    # ---- ids -----------------------------------------------------------------------------------
    def encode(self, caption, max_length=300):
        ids = [START] + [self.vocab.get(w, UNK) for w in self.words_of(caption)] + [END]
        if len(ids) > max_length:  # ref dataset.py:223-224: truncate, END stays last
            ids[max_length - 1:] = [END]
        return ids + [PAD] * (max_length - len(ids))

    def encode_batch(self, captions, max_length=300):
        return torch.tensor([self.encode(c, max_length) for c in captions], dtype=torch.long)

    def decode(self, ids, keep_period=True):
        """Words of the ids that are not PAD/START/END/UNK ('.' kept unless asked otherwise); stops at END."""
        out = []
        for i in ids:
            i = int(i)
            if i == END:
                break
            if i in (PAD, START, UNK) or (i == PERIOD and not keep_period):
                continue
            out.append(self.idx2word[i])
        return " ".join(out)


class FeatureCache:
    """``<path>.npy`` (features, C-order, fp32 or fp16) + ``<path>.json`` ({version, names, shape, dtype, source})."""

    VERSION = 1

    def __init__(self, path, array, names, meta):
        self.path, self.array, self.names, self.meta = path, array, list(names), meta
        self.index = {n: i for i, n in enumerate(self.names)}
        if len(self.index) != len(self.names):
            raise ValueError("duplicate image names in the feature cache")

    @classmethod
    def create(cls, path, names, feature_shape, dtype="float32", source=""):
        names = list(names)
        shape = (len(names),) + tuple(int(s) for s in feature_shape)
        if dtype not in ("float32", "float16"):
            raise ValueError("feature cache dtype must be float32 or float16")
        arr = np.lib.format.open_memmap(path + ".npy", mode="w+", dtype=np.dtype(dtype), shape=shape)
        meta = {"version": cls.VERSION, "names": names, "shape": list(shape), "dtype": dtype, "source": source}
        with open(path + ".json", "w") as f:
            json.dump(meta, f)
        return cls(path, arr, names, meta)

    @classmethod
    def open(cls, path, writable=False):
        with open(path + ".json") as f:
            meta = json.load(f)
        if meta.get("version") != cls.VERSION:
            raise ValueError(f"{path}.json: feature cache version {meta.get('version')}, expected {cls.VERSION}")
        arr = np.load(path + ".npy", mmap_mode="r+" if writable else "r", allow_pickle=False)
        if list(arr.shape) != meta["shape"] or str(arr.dtype) != meta["dtype"]:
            raise ValueError(f"{path}: array {arr.shape} {arr.dtype} does not match its side-car {meta['shape']} {meta['dtype']}")
        return cls(path, arr, meta["names"], meta)

    def __len__(self):
        return len(self.names)

    def put(self, start, feats):
        """Write a batch of features at rows [start, start + len(feats))."""
        feats = feats.detach().float().cpu().numpy() if torch.is_tensor(feats) else np.asarray(feats)
        if feats.shape[1:] != self.array.shape[1:]:
            raise ValueError(f"feature shape {feats.shape[1:]} does not match the cache's {self.array.shape[1:]}")
        self.array[start:start + len(feats)] = feats.astype(self.array.dtype)

    def flush(self):
        if hasattr(self.array, "flush"):
            self.array.flush()

    def get(self, key):
        i = self.index[key] if isinstance(key, str) else int(key)
        return torch.from_numpy(np.array(self.array[i], dtype=np.float32))

    def rows(self, idx):
        """fp32 tensor of rows ``idx`` (any integer sequence): one gather from the mapped file."""
        idx = np.asarray(idx, dtype=np.int64)
        order = np.argsort(idx, kind="stable")
        out = np.empty((len(idx),) + self.array.shape[1:], dtype=np.float32)
        out[order] = self.array[idx[order]]
        return torch.from_numpy(out)


def build_feature_cache(path, names, batches, extractor, feature_shape, dtype="float32", source=""):
    """Run ``extractor`` (e.g. the caller's ViT with its head removed, ref models.py:26-29) over
    ``batches`` (an iterable of image tensors, in ``names`` order) and write the cache."""
    cache = FeatureCache.create(path, names, feature_shape, dtype, source)
    at = 0
    with torch.no_grad():
        for images in batches:
            feats = extractor(images)
            cache.put(at, feats)
            at += len(feats)
    if at != len(cache):
        raise ValueError(f"{at} features written for {len(cache)} names")
    cache.flush()
    return cache


class CachedFeatureCaptionDataset(torch.utils.data.Dataset):
    """(features, caption_ids) per annotation, ref dataset.py:185-235 with features in the image slot.

    ``annotations``: ``{image_name: caption}`` (the reference's JSON files, dataset.py:60-62)."""

    def __init__(self, annotations, cache, tokenizer, max_length=300):
        self.items = list(annotations.items())
        self.cache, self.tokenizer, self.max_length = cache, tokenizer, max_length
        missing = [n for n, _ in self.items if n not in cache.index]
        if missing:
            raise KeyError(f"{len(missing)} annotated images have no cached features, e.g. {missing[:3]}")

    def __len__(self):
        return len(self.items)

    def __getitem__(self, idx):
        name, caption = self.items[idx]
        return self.cache.get(name), torch.tensor(self.tokenizer.encode(caption, self.max_length), dtype=torch.long)

    def batch(self, indices):
        """One collated batch without a DataLoader: (features [B, ...] fp32, ids [B, max_length] int64)."""
        names = [self.items[i][0] for i in indices]
        feats = self.cache.rows([self.cache.index[n] for n in names])
        ids = self.tokenizer.encode_batch([self.items[i][1] for i in indices], self.max_length)
        return feats, ids


class DevicePrefetcher:
    """Iterate a loader's ``(features, caption_ids)`` batches as device tensors, with the host-to-device copy of
    the following batches in flight on a copy stream while the current one is trained on.

    The reference moves each batch synchronously at the top of its loop (``images.to(device)``, ref
    train.py:26-27); at the C2 shape that is a 103 MB copy per step, 1-2 ms of PCIe time that this hides.
    Batches go through pinned host memory (torch's caching host allocator recycles the buffers once the copy's
    event has passed); the consumer's stream waits on the copy's event, and the device tensors are marked as used
    by the consumer's stream so the caching allocator does not hand them out again early.
    """

    def __init__(self, loader, device, depth=2):
        self.loader, self.device, self.depth = loader, torch.device(device), max(1, int(depth))
        if self.device.type != "cuda":
            raise ValueError("DevicePrefetcher copies to a GPU; iterate the loader itself on CPU")

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        import collections
        copy_stream = torch.cuda.Stream(self.device)
        it = iter(self.loader)
        queue = collections.deque()

        def issue():
            try:
                batch = next(it)
            except StopIteration:
                return
            with torch.cuda.stream(copy_stream):
                dev = tuple((t if t.is_pinned() else t.pin_memory()).to(self.device, non_blocking=True)
                            if torch.is_tensor(t) else t for t in batch)
                ev = torch.cuda.Event()
                ev.record(copy_stream)
            queue.append((dev, ev))

        for _ in range(self.depth):
            issue()
        while queue:
            dev, ev = queue.popleft()
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)
            for t in dev:
                if torch.is_tensor(t):
                    t.record_stream(cur)
            issue()
            yield dev
