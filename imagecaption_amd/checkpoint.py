"""Checkpoint interchange with the reference.

The reference writes ``checkpoints/best_model.pth`` (ref train.py:485-495) as a dict with keys
``epoch, model_state_dict, optimizer_state_dict, scheduler_state_dict, loss, metrics, vocab_size,
vocab_idx2word, initial_lr`` and reads it back at ref main.py:107-149 (resume) and test.py:13-21
(inference).  Its ``model_state_dict`` holds the ViT under ``vit.*`` in front of the decoder-path
keys, and its AdamW was built over ALL ``model.parameters()`` (ref main.py:95), so optimizer state
indices count the ViT's parameters first.

This build keeps no ViT (``model.vit`` is whatever feature extractor the caller plugs in), so:

* ``load_reference_checkpoint`` takes the decoder-path keys (strictly: every one must be present
  and fit), hands the ``vit.*`` tensors back untouched, and re-indexes the optimizer state onto the
  decoder-only parameter list (for a ``torch.optim.AdamW`` or a ``CaptionTrainer``);
* ``save_reference_checkpoint`` writes the same dict; given the ViT's state (``vit_state``) the file
  loads in the reference with its strict ``load_state_dict`` and its optimizer resumes, without it
  the file is decoder-only (``strict=False`` on the reference side).

Files are read with ``torch.load(..., weights_only=True)`` only: nothing in a checkpoint is executed.
The reference's ``metrics`` dict holds ``np.mean(...)`` results (ref train.py:251-264 -> numpy.float64), which
pickle through ``numpy._core.multiarray.scalar`` + a dtype class; those data constructors (and nothing else)
are allow-listed for the load.  A file the safe loader still refuses is reported, not force-loaded.

One such refusal is expected and is the right failure: torch keys its allow-list by a constructor's own
``__module__.__qualname__``, so under numpy >= 2 the list holds ``numpy._core.multiarray.scalar``; a checkpoint
whose metrics were pickled under numpy 1.x names ``numpy.core.multiarray.scalar`` and is REFUSED here (loudly, with
the loader's message) -- re-save it with plain floats, or load it under the numpy it was written with.  The old
spelling is deliberately not aliased into the list.  After a load every numpy scalar found in the file must be of a
numeric kind (float / int / unsigned / bool); anything else (object, structured, string dtypes, which ``numpy.dtype``
in the allow-list could in principle describe) is rejected.
"""
import numpy as np
import torch

VIT_PREFIX = "vit."
KEYS = ("epoch", "model_state_dict", "optimizer_state_dict", "scheduler_state_dict", "loss", "metrics",
        "vocab_size", "vocab_idx2word", "initial_lr")


def _numpy_scalar_globals():
    """What a pickled numpy scalar needs: the scalar re-constructor, ``numpy.dtype`` and the dtype classes of the
    numeric types a metrics dict can hold.  They rebuild a number from raw bytes and execute no code."""
    try:
        scalar = np._core.multiarray.scalar   # numpy >= 2
    except AttributeError:                  # numpy 1.x
        scalar = np.core.multiarray.scalar
    kinds = ("float64", "float32", "float16", "int64", "int32", "int16", "int8", "uint8", "bool")
    return [scalar, np.dtype] + sorted({type(np.dtype(k)) for k in kinds}, key=lambda c: c.__name__)


def _require_numeric_scalars(obj, path, where="checkpoint"):
    """Every numpy scalar / dtype reachable in a loaded checkpoint must be of a numeric kind."""
    if isinstance(obj, dict):
        for k, v in obj.items():
            _require_numeric_scalars(v, path, f"{where}[{k!r}]")
    elif isinstance(obj, (list, tuple)):
        for i, v in enumerate(obj):
            _require_numeric_scalars(v, path, f"{where}[{i}]")
    elif isinstance(obj, (np.generic, np.ndarray, np.dtype)):
        dt = obj if isinstance(obj, np.dtype) else obj.dtype
        if dt.kind not in "fiub":
            raise RuntimeError(f"{path}: {where} is a numpy value of dtype {dt} (kind {dt.kind!r}); only numeric numpy "
                               "scalars are accepted in a checkpoint")


def split_state_dict(sd):
    """(decoder-path entries, vit entries with the prefix kept)."""
    dec = {k: v for k, v in sd.items() if not k.startswith(VIT_PREFIX)}
    vit = {k: v for k, v in sd.items() if k.startswith(VIT_PREFIX)}
    return dec, vit


def _shift_optimizer_state(osd, offset, n_keep=None):
    """Re-index a torch optimizer state_dict: new index = old index - offset; entries outside
    [0, n_keep) are dropped (offset > 0 strips the ViT's leading parameters, < 0 makes room for them)."""
    state = {}
    for k, v in osd["state"].items():
        j = int(k) - offset
        if j >= 0 and (n_keep is None or j < n_keep):
            state[j] = v
    groups = []
    for g in osd["param_groups"]:
        g = dict(g)
        g["params"] = [int(i) - offset for i in g["params"] if int(i) - offset >= 0 and (n_keep is None or int(i) - offset < n_keep)]
        groups.append(g)
    return {"state": state, "param_groups": groups}


def load_reference_checkpoint(path, model, optimizer=None, scheduler=None, map_location="cpu"):
    """Load a reference ``best_model.pth`` into the drop-in ``model`` (+ optimizer / scheduler).

    ``optimizer`` may be a ``torch.optim.Optimizer`` over ``model.parameters()`` or a
    ``CaptionTrainer``.  Returns the checkpoint dict with ``model_state_dict`` replaced by the
    ``vit.*`` remainder (for the caller's feature extractor) under ``vit_state_dict``."""
    try:
        with torch.serialization.safe_globals(_numpy_scalar_globals()):
            ckpt = torch.load(path, map_location=map_location, weights_only=True)
    except Exception as e:  # the safe loader refused: say so, never fall back to unpickling
        raise RuntimeError(f"{path}: torch.load(weights_only=True) refused this file ({e}); it is not loaded") from e
    if "model_state_dict" not in ckpt:
        raise RuntimeError(f"{path}: no 'model_state_dict' (keys: {sorted(ckpt)})")
    _require_numeric_scalars(ckpt, path)
    dec, vit = split_state_dict(ckpt["model_state_dict"])
    own = model.state_dict()
    own_dec = {k for k in own if not k.startswith(VIT_PREFIX)}
    missing = sorted(own_dec - set(dec) - {"pos_encoder.pe"})
    unexpected = sorted(set(dec) - own_dec)
    if missing or unexpected:
        raise RuntimeError(f"{path}: decoder-path keys differ (missing {missing[:5]}, unexpected {unexpected[:5]})")
    with torch.no_grad():
        for k, v in dec.items():
            if own[k].shape != v.shape:
                raise RuntimeError(f"{path}: {k} has shape {tuple(v.shape)}, the model expects {tuple(own[k].shape)}"
                                   " (vocab_size / d_model of the checkpoint differ)")
            own[k].copy_(v)  # state_dict() tensors alias the parameters (and the trainer's flat block)
    n_vit = len(vit)  # torchvision's ViT registers parameters only, so entries == parameters
    n_dec = sum(1 for _ in model.parameters())
    if optimizer is not None and ckpt.get("optimizer_state_dict") is not None:
        osd = _shift_optimizer_state(ckpt["optimizer_state_dict"], n_vit, n_dec)
        if hasattr(optimizer, "load_adamw_state_dict"):
            optimizer.load_adamw_state_dict(osd)
        else:
            optimizer.load_state_dict(osd)
    if scheduler is not None and ckpt.get("scheduler_state_dict") is not None:
        scheduler.load_state_dict(ckpt["scheduler_state_dict"])
    out = {k: v for k, v in ckpt.items() if k != "model_state_dict"}
    out["vit_state_dict"] = vit
    return out


def save_reference_checkpoint(path, model, optimizer=None, scheduler=None, epoch=0, loss=0.0, metrics=None,
                              vocab_idx2word=None, initial_lr=None, vit_state=None):
    """Write ``path`` in the reference's format (ref train.py:485-495).  ``vit_state``: the ``vit.*``
    entries to put in front (as returned by ``load_reference_checkpoint``, or ``{'vit.' + k: v}`` of a
    torchvision ViT's state_dict)."""
    vit_state = dict(vit_state or {})
    bad = [k for k in vit_state if not k.startswith(VIT_PREFIX)]
    if bad:
        raise ValueError(f"vit_state keys must start with '{VIT_PREFIX}': {bad[:3]}")
    dec, _ = split_state_dict(model.state_dict())
    sd = {k: v.detach().cpu().clone() for k, v in vit_state.items()}
    sd.update({k: v.detach().cpu().clone() for k, v in dec.items()})
    osd = None
    if optimizer is not None:
        raw = optimizer.adamw_state_dict() if hasattr(optimizer, "adamw_state_dict") else optimizer.state_dict()
        osd = _shift_optimizer_state(raw, -len(vit_state))
        if vit_state:  # the reference's optimizer lists the (stateless, frozen) ViT parameters first
            for g in osd["param_groups"]:
                g["params"] = list(range(len(vit_state))) + g["params"]
        osd["state"] = {k: {n: (t.detach().cpu().clone() if torch.is_tensor(t) else t) for n, t in st.items()}
                        for k, st in osd["state"].items()}
    vocab_size = dec["embedding.weight"].shape[0]
    torch.save({
        "epoch": int(epoch),
        "model_state_dict": sd,
        "optimizer_state_dict": osd,
        "scheduler_state_dict": scheduler.state_dict() if scheduler is not None else None,
        "loss": float(loss),
        "metrics": {k: float(v) for k, v in (metrics or {}).items()},
        "vocab_size": int(vocab_size),
        "vocab_idx2word": {int(k): str(v) for k, v in (vocab_idx2word or {}).items()},
        "initial_lr": float(initial_lr) if initial_lr is not None else None,
    }, path)
    return path
