"""CPU suite: the oracle against the committed golden vectors (oracle #1 stock torch.nn, oracle #2 plain
restatement), so a drifting torch build or an edited oracle is caught before it is trusted as the checker."""
import os

import numpy as np
import torch

from conftest import GOLDEN
from oracle import plain_oracle
from oracle.torch_oracle import OracleCaptionDecoder, beam_decode, greedy_decode, label_smoothed_loss

TINY = dict(vocab_size=50, d_model=64, nhead=2, num_decoder_layers=2, feature_dim=24)


def _sd():
    w = np.load(os.path.join(GOLDEN, "tiny_weights.npz"))
    return {k: torch.from_numpy(w[k]) for k in w.files}


def test_seeded_init_reproduces_golden_weights():
    torch.manual_seed(42)
    model = OracleCaptionDecoder(**TINY)
    sd = _sd()
    for k, v in model.state_dict().items():
        if k != "pos_encoder.pe":
            assert torch.equal(v, sd[k]), k


def test_stock_and_plain_oracle_match_golden():
    for S in (1, 5):
        g = np.load(os.path.join(GOLDEN, f"tiny_S{S}.npz"))
        sd = _sd()
        feats, caps = torch.from_numpy(g["features"]), torch.from_numpy(g["captions"])
        collect = {}
        logits = plain_oracle.forward(sd, feats, caps[:, :-1], 2, 2, collect)
        assert np.abs(logits.numpy() - g["logits"]).max() < 2e-5
        for k, v in collect.items():
            assert np.abs(v.numpy() - g["act/" + k]).max() < 2e-5, k
        loss = plain_oracle.label_smoothed_loss(logits, caps[:, 1:])
        assert abs(loss.item() - float(g["loss"])) < 1e-5
        model = OracleCaptionDecoder(**TINY)
        model.load_state_dict(sd, strict=False)
        model.eval()
        out = model(feats, caps[:, :-1])
        assert np.abs(out.detach().numpy() - g["logits"]).max() < 1e-5
        loss = label_smoothed_loss(out, caps[:, 1:])
        loss.backward()
        for n, p in model.named_parameters():
            ref = g["g/" + n]
            assert np.abs(p.grad.numpy() - ref).max() <= 1e-5 * max(np.abs(ref).max(), 1e-3), n


def test_closed_form_loss_equals_reference_arithmetic():
    g = torch.Generator().manual_seed(0)
    z = torch.randn(4, 7, 31, generator=g) * 3
    y = torch.randint(0, 31, (4, 7), generator=g)
    a = label_smoothed_loss(z, y)  # ref train.py:41-50 verbatim
    b = plain_oracle.label_smoothed_loss(z, y)
    c = torch.nn.functional.cross_entropy(z.reshape(-1, 31), y.reshape(-1), label_smoothing=0.1)
    assert abs(a.item() - b.item()) < 1e-5 and abs(a.item() - c.item()) < 1e-5


def test_plain_adamw_matches_golden_step():
    g = np.load(os.path.join(GOLDEN, "tiny_S1.npz"))
    sd = _sd()
    coef = min(1.0, 1.0 / (float(g["grad_norm"]) + 1e-6))  # ref train.py:56
    for k in ("output_layer.3.weight", "transformer_decoder.layers.1.linear1.weight", "embedding.weight"):
        grad = torch.from_numpy(g["g/" + k]) * coef
        p1, _, _ = plain_oracle.adamw_step(sd[k], grad, torch.zeros_like(grad), torch.zeros_like(grad), 1)
        sig = np.abs(g["g/" + k]) > 1e-4 * np.abs(g["g/" + k]).max()
        assert np.abs(p1.numpy() - g["w1/" + k])[sig].max() < 2e-6, k


def test_decode_oracle_reproduces_golden_ids():
    g = np.load(os.path.join(GOLDEN, "decode_tiny.npz"))
    for S in (1, 5):
        torch.manual_seed(42)
        model = OracleCaptionDecoder(**TINY)
        with torch.no_grad():
            model.output_layer[3].weight.mul_(8.0)
        feats = torch.from_numpy(g[f"S{S}/features"])
        ids, gap = greedy_decode(model, feats, max_len=20)
        assert np.array_equal(ids.numpy(), g[f"S{S}/greedy_ids"])
        ids, sc, gap = beam_decode(model, feats, beam=3, max_len=20)
        assert np.array_equal(ids.numpy(), g[f"S{S}/beam3_ids"])
        assert np.abs(sc.numpy() - g[f"S{S}/beam3_scores"]).max() < 1e-4
        # beam 1 == greedy until END; best beam is never worse than greedy
        b1, s1, _ = beam_decode(model, feats, beam=1, max_len=20)
        gi, _ = greedy_decode(model, feats, max_len=20)
        assert np.array_equal(b1[:, 0].numpy(), gi.numpy())
        assert (sc[:, 0] >= s1[:, 0] - 1e-5).all()


def test_causal_mask_equals_torch_canonical():
    m = OracleCaptionDecoder.generate_square_subsequent_mask(9)
    assert torch.equal(m, torch.nn.Transformer.generate_square_subsequent_mask(9))
