"""Generate the committed golden fixtures under tests/golden/.  TEST INFRASTRUCTURE ONLY.

Run in the build container (needs /root/reference for the caption JSON + vocab pickles):

    python -m oracle.gen_golden

Outputs (all small, data only):
* vocab.json                  the 108-word vocabulary, id order, rebuilt by oracle/tokenizer.py and
                              checked here against the reference fixture data/cache/*_vocab_full_v1.pkl
* captions_{train,test}.npz   tokenised captions (START..END) as flat uint8 + uint16 lengths
* tiny_weights.npz            tiny-config initial state_dict (seed 42; shared by both S)
* tiny_S{1,5}.npz             tiny config: inputs, activations, logits, loss, every gradient, clip
                              norm; S=1 also holds the parameters after one AdamW step
* c1_shape.npz                C1 shape (d=512, V=108, T=299, B=2): logits, loss, grad norm
* c2_shape.npz                bench workload (C2, B=256): 64 sampled logit rows, loss, global and
                              per-parameter gradient norms (weights are rebuilt from seed 42)
* c3_shape.npz                configs[2] at full size (C3, B=256, d 1024, V 30 000, S 196): the same summary
                              (`python -m oracle.gen_golden c3`; not in the default set: minutes of CPU time)
* decode_tiny.npz             greedy / beam-3 / beam-5 ids + scores + logit-gap margins (tiny config)
"""
import json
import os
import sys

import numpy as np
import torch

from . import plain_oracle, tokenizer
from .torch_oracle import (OracleCaptionDecoder, beam_decode, greedy_decode, label_smoothed_loss,
                           synthetic_captions, train_step)

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

TINY = dict(vocab_size=50, d_model=64, nhead=2, num_decoder_layers=2, feature_dim=24)  # dh = 32


def _np(sd):
    return {k: v.detach().cpu().numpy() for k, v in sd.items()}


def gen_vocab_and_captions():
    with open(f"{REF}/data/train_captions.json") as f:
        train = json.load(f)
    with open(f"{REF}/data/test_captions.json") as f:
        test = json.load(f)
    vocab = tokenizer.build_vocab(train.values())
    vocab_test = tokenizer.build_vocab(test.values())
    for name, v in (("train", vocab), ("test", vocab_test)):
        ref = tokenizer.read_vocab_pickle_without_executing(f"{REF}/data/cache/{name}_captions_vocab_full_v1.pkl")
        assert ref["version"] == "v1"
        assert list(ref["vocab"].items()) == list(v.items()), f"{name} vocab differs from reference fixture"
    assert vocab == vocab_test
    words = [w for w, _ in sorted(vocab.items(), key=lambda kv: kv[1])]
    with open(f"{OUT}/vocab.json", "w") as f:
        json.dump(words, f)
    for name, caps in (("train", train), ("test", test)):
        toks, lens = [], []
        for c in caps.values():
            ids = tokenizer.encode(c, vocab, 300)  # main.py:36 max_len
            n = ids.index(vocab["<END>"]) + 1
            toks.extend(ids[:n])
            lens.append(n)
        np.savez_compressed(f"{OUT}/captions_{name}.npz", tokens=np.asarray(toks, dtype=np.uint8),
                            lengths=np.asarray(lens, dtype=np.uint16))
        print(name, len(lens), "captions", len(toks), "tokens; len min/med/max",
              min(lens), int(np.median(lens)), max(lens))


def gen_tiny(S):
    torch.manual_seed(42)
    model = OracleCaptionDecoder(**TINY)
    gen = torch.Generator().manual_seed(1234)
    B, T = 3, 12
    feat_shape = (B, TINY["feature_dim"]) if S == 1 else (B, S, TINY["feature_dim"])
    features = torch.randn(feat_shape, generator=gen)
    captions = synthetic_captions(B, T, TINY["vocab_size"], gen, min_len=4)
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    out = {"features": features.numpy(), "captions": captions.numpy()}
    if S == 1:  # same seed -> same initial weights for every S; stored once
        np.savez_compressed(f"{OUT}/tiny_weights.npz",
                            **{k: v for k, v in _np(sd0).items() if k != "pos_encoder.pe"})
    # eval-mode activations from the plain restatement, cross-checked against stock modules
    model.eval()
    collect = {}
    with torch.no_grad():
        logits_plain = plain_oracle.forward(sd0, features, captions[:, :-1], TINY["nhead"],
                                            TINY["num_decoder_layers"], collect)
        logits = model(features, captions[:, :-1])
        memory = model.project(features)
    err = (logits - logits_plain).abs().max().item()
    assert err < 2e-5, err
    out["memory"] = memory.numpy()
    for k, v in collect.items():
        out["act/" + k] = v.numpy()
    out["logits"] = logits.numpy()
    # one training step with dropout off
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=0.01)  # main.py:95
    loss, norm, _ = train_step(model, opt, features, captions)
    out["loss"] = loss.numpy()
    out["grad_norm"] = norm.numpy()
    lp = plain_oracle.label_smoothed_loss(logits_plain, captions[:, 1:])
    assert abs(lp.item() - loss.item()) < 1e-5
    # NB: .grad now holds the CLIPPED grads (train.py:56 clips in place); save unclipped = grad / coef
    coef = min(1.0, 1.0 / (norm.item() + 1e-6))
    for n, p in model.named_parameters():
        out["g/" + n] = (p.grad / coef).numpy()
    if S == 1:  # parameters after the AdamW step (main.py:95 hyper-parameters)
        for k, v in _np(model.state_dict()).items():
            if k != "pos_encoder.pe":
                out["w1/" + k] = v
    np.savez_compressed(f"{OUT}/tiny_S{S}.npz", **out)
    print(f"tiny S={S}: loss {loss.item():.6f} norm {norm.item():.6f} plain-vs-stock logits err {err:.2e}")


def gen_c1_shape():
    torch.manual_seed(42)
    model = OracleCaptionDecoder(vocab_size=108, d_model=512, nhead=8, num_decoder_layers=6, feature_dim=768)
    model.eval()
    gen = torch.Generator().manual_seed(1234)
    B, T = 2, 299
    features = torch.randn(B, 768, generator=gen)
    caps = np.load(f"{OUT}/captions_train.npz")
    ids = torch.zeros(B, T + 1, dtype=torch.long)
    off = 0
    for b in range(B):
        n = int(caps["lengths"][b])
        ids[b, :n] = torch.from_numpy(caps["tokens"][off:off + n].astype(np.int64))
        off += n
    logits = model(features, ids[:, :-1])
    loss = label_smoothed_loss(logits, ids[:, 1:])
    loss.backward()
    gn = torch.sqrt(sum((p.grad ** 2).sum() for p in model.parameters()))
    np.savez_compressed(f"{OUT}/c1_shape.npz", features=features.numpy(), captions=ids.numpy(),
                        logits=logits.detach().numpy(), loss=loss.detach().numpy(), grad_norm=gn.numpy())
    print(f"c1 shape: loss {loss.item():.6f} grad-norm {gn.item():.6f}")


def c2_batch(B=256, S=49, din=2048, T=20, V=10000, seed=1234):
    """The bench workload's batch: features ~N(0,1), synthetic ids."""
    gen = torch.Generator().manual_seed(seed)
    features = torch.randn(B, S, din, generator=gen)
    return features, synthetic_captions(B, T, V, gen)


def gen_c2_shape():
    """Full bench size (B=256): too big to commit whole, so 64 sampled logit rows, the loss, the global
    and per-parameter gradient norms.  Weights are not stored: seed 42 + OracleCaptionDecoder init is
    deterministic on CPU, the test rebuilds them the same way."""
    torch.manual_seed(42)
    model = OracleCaptionDecoder(vocab_size=10000, d_model=512, nhead=8, num_decoder_layers=6, feature_dim=2048)
    model.eval()
    features, ids = c2_batch()
    logits = model(features, ids[:, :-1])
    loss = label_smoothed_loss(logits, ids[:, 1:])
    loss.backward()
    names = [n for n, _ in model.named_parameters()]
    norms = np.array([float(p.grad.double().pow(2).sum().sqrt()) for _, p in model.named_parameters()])
    gn = float(np.sqrt((norms ** 2).sum()))
    rows = torch.randperm(logits.shape[0] * logits.shape[1], generator=torch.Generator().manual_seed(7))[:64]
    flat = logits.detach().reshape(-1, logits.shape[-1])
    np.savez_compressed(f"{OUT}/c2_shape.npz", rows=rows.numpy(), logit_rows=flat[rows].numpy(),
                        loss=loss.detach().numpy(), grad_norm=np.float64(gn), param_names=np.array(names),
                        param_grad_norms=norms, weight_checksum=np.float64(sum(float(p.detach().double().sum()) for p in model.parameters())))
    print(f"c2 shape: loss {loss.item():.6f} grad-norm {gn:.6f}")


def c3_batch(B=256, S=196, din=512, T=20, V=30000, seed=1234):
    """BASELINE.json configs[2] (C3): 14 x 14 x 512 grid features, hidden 1024, vocabulary 30 000."""
    gen = torch.Generator().manual_seed(seed)
    features = torch.randn(B, S, din, generator=gen)
    return features, synthetic_captions(B, T, V, gen)


def gen_c3_shape():
    """configs[2] at FULL size (B = 256, d 1024, dh 128, V 30 000, S 196), summarised like c2_shape.npz: 64 sampled
    logit rows, loss, global and per-parameter gradient norms, weight checksum (weights rebuilt from seed 42)."""
    torch.manual_seed(42)
    model = OracleCaptionDecoder(vocab_size=30000, d_model=1024, nhead=8, num_decoder_layers=6, feature_dim=512)
    model.eval()
    features, ids = c3_batch()
    logits = model(features, ids[:, :-1])
    loss = label_smoothed_loss(logits, ids[:, 1:])
    loss.backward()
    names = [n for n, _ in model.named_parameters()]
    norms = np.array([float(p.grad.double().pow(2).sum().sqrt()) for _, p in model.named_parameters()])
    gn = float(np.sqrt((norms ** 2).sum()))
    rows = torch.randperm(logits.shape[0] * logits.shape[1], generator=torch.Generator().manual_seed(7))[:64]
    flat = logits.detach().reshape(-1, logits.shape[-1])
    np.savez_compressed(f"{OUT}/c3_shape.npz", rows=rows.numpy(), logit_rows=flat[rows].numpy(),
                        loss=loss.detach().numpy(), grad_norm=np.float64(gn), param_names=np.array(names),
                        param_grad_norms=norms, weight_checksum=np.float64(sum(float(p.detach().double().sum()) for p in model.parameters())))
    print(f"c3 shape: loss {loss.item():.6f} grad-norm {gn:.6f}")


def gen_decode():
    out = {}
    for S in (1, 5):
        torch.manual_seed(42)
        model = OracleCaptionDecoder(**TINY)
        # sharpen the head so decode is not a near-uniform tie-fest on random weights
        with torch.no_grad():
            model.output_layer[3].weight.mul_(8.0)
        gen = torch.Generator().manual_seed(4321)
        B = 8
        feat_shape = (B, TINY["feature_dim"]) if S == 1 else (B, S, TINY["feature_dim"])
        features = torch.randn(feat_shape, generator=gen)
        out[f"S{S}/features"] = features.numpy()
        ids, gap = greedy_decode(model, features, max_len=20)
        out[f"S{S}/greedy_ids"] = ids.numpy()
        out[f"S{S}/greedy_min_gap"] = np.float32(gap)
        for W in (3, 5):
            ids, sc, gap = beam_decode(model, features, beam=W, max_len=20)
            out[f"S{S}/beam{W}_ids"] = ids.numpy()
            out[f"S{S}/beam{W}_scores"] = sc.numpy()
            out[f"S{S}/beam{W}_min_gap"] = np.float32(gap)
            print(f"decode S={S} beam {W}: min gap {gap:.3e}")
    np.savez_compressed(f"{OUT}/decode_tiny.npz", **out)


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["vocab", "tiny", "c1", "c2", "decode"]
    if "vocab" in which:
        gen_vocab_and_captions()
    if "tiny" in which:
        gen_tiny(1)
        gen_tiny(5)
    if "c1" in which:
        gen_c1_shape()
    if "c2" in which:
        gen_c2_shape()
    if "c3" in which:  # minutes of CPU time and ~20 GB of host memory: not part of the default set
        gen_c3_shape()
    if "decode" in which:
        gen_decode()


if __name__ == "__main__":
    main()
