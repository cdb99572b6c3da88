"""Checkpoint interchange and the reference's LR driver code ON THE DEVICE.

A reference-format ``best_model.pth`` (ref train.py:485-495) is written from the CPU oracle after two AdamW steps --
a fake ``vit.*`` in front, numpy-scalar metrics as ``evaluate_metrics`` returns them -- then loaded into the HIP
model + fused ``CaptionTrainer``: the logits must match the oracle's, and the NEXT fused step must equal the
oracle's next ``clip_grad_norm_`` + ``AdamW.step`` (moments and step counter really resumed)."""
import numpy as np
import pytest
import torch

from test_checkpoint import TINY, FakeRefModel

pytestmark = pytest.mark.gpu
DEV = "cuda"


def relerr(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def _oracle_step(ref, opt, x, caps):
    """One iteration of ref train.py:36-59 on the fake reference model (eval mode: dropout off)."""
    from oracle.torch_oracle import label_smoothed_loss
    opt.zero_grad()
    loss = label_smoothed_loss(ref(x, caps[:, :-1]), caps[:, 1:])
    loss.backward()
    torch.nn.utils.clip_grad_norm_(ref.parameters(), max_norm=1.0)
    opt.step()
    return float(loss)


def test_reference_checkpoint_resumes_on_device_and_next_step_matches_oracle(tmp_path):
    from imagecaption_amd import ImageCaptioningModel
    from imagecaption_amd.checkpoint import load_reference_checkpoint, save_reference_checkpoint
    from imagecaption_amd.trainer import CaptionTrainer
    torch.manual_seed(0)
    ref = FakeRefModel().eval()
    opt = torch.optim.AdamW(ref.parameters(), lr=1e-3, weight_decay=0.01)  # ref main.py:95
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, mode="max", factor=0.5, patience=2)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(6, 5, 8, generator=g)  # "images" of the fake ViT: [B, S, 8] -> features [B, S, 24]
    caps = torch.randint(5, 50, (6, 13), generator=g)
    caps[:, 0] = 1
    for _ in range(2):
        _oracle_step(ref, opt, x, caps)
    path = str(tmp_path / "best_model.pth")
    torch.save({"epoch": 7, "model_state_dict": ref.state_dict(), "optimizer_state_dict": opt.state_dict(),
                "scheduler_state_dict": sched.state_dict(), "loss": 3.5,
                "metrics": {"bleu1": 0.5, "bleu4": 0.25, "meteor": np.mean([0.5, 0.25]), "rouge_l": np.mean([0.5]),
                            "cider": np.mean([1.0]), "spice": np.mean([0.25])},
                "vocab_size": 50, "vocab_idx2word": {0: "<PAD>", 1: "<START>", 2: "<END>"}, "initial_lr": 1e-3}, path)

    model = ImageCaptioningModel(**TINY).to(DEV)
    tr = CaptionTrainer(model, dropout=0.0, lr=5e-4)
    sched2 = torch.optim.lr_scheduler.ReduceLROnPlateau(tr.optimizer, mode="max", factor=0.5, patience=2)
    info = load_reference_checkpoint(path, model, tr, sched2)
    assert info["epoch"] == 7 and float(info["metrics"]["meteor"]) == 0.375 and tr.step_count == 2 and tr.lr == 1e-3
    assert all(p.device.type == "cuda" for p in model.parameters())
    assert model.embedding.weight.data_ptr() == tr.flat_p.data_ptr()  # still views of the flat block

    with torch.no_grad():
        feats = ref.vit(x)  # the encoder stays on the caller's side; the decoder path gets its features
        want = ref(x, caps[:, :-1])
    model.eval()
    mask = model.generate_square_subsequent_mask(caps.shape[1] - 1)  # ref train.py:33
    got = model(feats.to(DEV), caps[:, :-1].to(DEV), mask)            # ref train.py:38
    assert relerr(got, want) < 5e-5
    # the NEXT step: moments, step counter (bias correction) and lr all come from the file
    loss_ref = _oracle_step(ref, opt, x, caps)
    loss = tr.step(feats.to(DEV), caps.to(DEV), train=False)
    assert abs(loss.item() - loss_ref) < 1e-5 * abs(loss_ref)
    rparams = {n: p for n, p in ref.named_parameters() if not n.startswith("vit.")}
    for n, p in model.named_parameters():
        # third Adam step: the update is lr * m_hat / (sqrt(v_hat) + eps), smooth in the gradient -- except where
        # the gradient is pure rounding noise on both sides (the key bias of attention: its exact gradient is 0),
        # there the update's sign is noise too, so those elements are only bounded by the step size
        diff = (p.detach().cpu() - rparams[n].detach()).abs()
        g = rparams[n].grad.abs()
        sig = g > 1e-4 * g.max()
        assert float(diff[sig].max()) < 2e-5, n
        assert float(diff.max()) <= 2.1e-3, n
    # and back: what the trainer saves, the reference's strict loads accept (ViT entries in front)
    out = str(tmp_path / "resaved.pth")
    save_reference_checkpoint(out, model, tr, sched2, epoch=8, loss=float(loss.item()), metrics={"bleu4": 0.3},
                              vocab_idx2word=info["vocab_idx2word"], initial_lr=1e-3, vit_state=info["vit_state_dict"])
    ck = torch.load(out, map_location="cpu", weights_only=True)
    ref2 = FakeRefModel()
    ref2.load_state_dict(ck["model_state_dict"])
    opt2 = torch.optim.AdamW(ref2.parameters(), lr=1e-3, weight_decay=0.01)
    opt2.load_state_dict(ck["optimizer_state_dict"])
    p_emb = dict(ref2.named_parameters())["embedding.weight"]
    assert float(opt2.state[p_emb]["step"]) == 3.0
    assert relerr(opt2.state[p_emb]["exp_avg"], opt.state[dict(ref.named_parameters())["embedding.weight"]]["exp_avg"]) < 1e-4


def test_reference_warmup_and_plateau_scheduler_drive_the_fused_trainer():
    """ref train.py:433-447 (warm-up writes param_groups[*]['lr']) and train.py:460 (ReduceLROnPlateau.step) run
    verbatim against ``trainer.optimizer``; the fused AdamW kernel must then step with exactly that lr."""
    from imagecaption_amd import ImageCaptioningModel
    from imagecaption_amd.trainer import CaptionTrainer
    torch.manual_seed(1)
    model = ImageCaptioningModel(**TINY).to(DEV)
    tr = CaptionTrainer(model, dropout=0.0, lr=1e-3)
    optimizer = tr.optimizer
    scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, mode="max", factor=0.5, patience=2)  # main.py:98-104
    g = torch.Generator().manual_seed(2)
    feats = torch.randn(4, 5, 24, generator=g).to(DEV)
    caps = torch.randint(5, 50, (4, 11), generator=g).to(DEV)

    def update_size():
        """|delta p| of one step on a parameter whose Adam update is ~lr in magnitude (first steps: m_hat/sqrt(v_hat) = +-1)."""
        before = tr.flat_p.clone()
        tr.step(feats, caps, train=False)
        d = (tr.flat_p - before * (1 - tr.lr * tr.wd)).abs()
        return float(d.max())

    # ---- the reference's lines, verbatim ----
    warmup_epochs = 2
    warmup_factor = 0.1
    initial_lr = optimizer.param_groups[0]['lr']
    seen = []
    for epoch in range(3):
        if epoch < warmup_epochs:
            factor = warmup_factor + (1 - warmup_factor) * (epoch / warmup_epochs)
            for param_group in optimizer.param_groups:
                param_group['lr'] = initial_lr * factor
        seen.append((optimizer.param_groups[0]['lr'], update_size()))
    assert [round(lr, 8) for lr, _ in seen] == [1e-4, 5.5e-4, 5.5e-4]  # 1.0x is never restored
    for lr, upd in seen:
        assert 0.5 * lr < upd <= 1.01 * lr, (lr, upd)  # the kernel really stepped with this lr
    # plateau: BLEU-4 does not improve for patience + 1 epochs -> lr halves (train.py:460)
    for bleu4 in (0.2, 0.2, 0.2, 0.2):
        scheduler.step(bleu4)
    current_lr = optimizer.param_groups[0]['lr']
    assert abs(current_lr - 2.75e-4) < 1e-12 and tr.lr == current_lr
    upd = update_size()
    assert 0.3 * current_lr < upd <= 1.01 * current_lr
    # zero_grad() of the facade must leave the gradient views in place (the stock one would set them to None)
    optimizer.zero_grad()
    assert model.embedding.weight.grad is not None and model.embedding.weight.grad.data_ptr() == tr.flat_g.data_ptr()
    # state_dict round trip in torch.optim.AdamW's format
    sd = optimizer.state_dict()
    assert sd["param_groups"][0]["lr"] == current_lr and len(sd["state"]) == len(list(model.parameters()))
    optimizer.load_state_dict(sd)
    assert tr.step_count == 4 and tr.lr == current_lr
