"""Checkpoint interchange with the reference format (ref train.py:485-495, main.py:107-149, test.py:13-21).

CPU only: a synthetic 'reference' model = fake ViT (parameters only, registered first) + the oracle's
decoder-path modules, trained one AdamW step, saved exactly as the reference saves."""

import numpy as np
import pytest
import torch

TINY = dict(vocab_size=50, d_model=64, nhead=2, num_decoder_layers=2, feature_dim=24)


class FakeRefModel(torch.nn.Module):
    """Registration order of ref models.py:21-73: vit first, then the decoder path."""

    def __init__(self):
        super().__init__()
        from oracle.torch_oracle import OracleCaptionDecoder
        self.vit = torch.nn.Sequential(torch.nn.Linear(8, 24), torch.nn.LayerNorm(24))
        for p in self.vit.parameters():
            p.requires_grad = False  # ref models.py:31-34
        dec = OracleCaptionDecoder(**TINY)
        for name in ("feature_projection", "embedding", "pos_encoder", "transformer_decoder", "output_layer"):
            setattr(self, name, getattr(dec, name))
        self._dec = [dec]  # not a submodule

    def forward(self, x, caps):
        return self._dec[0](self.vit(x), caps)


def _trained_reference(tmp_path):
    from oracle.torch_oracle import label_smoothed_loss
    torch.manual_seed(0)
    ref = FakeRefModel().eval()
    opt = torch.optim.AdamW(ref.parameters(), lr=1e-3, weight_decay=0.01)  # ref main.py:95: ALL parameters
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, mode="max", factor=0.5, patience=2)
    caps = torch.randint(5, 50, (3, 9))
    loss = label_smoothed_loss(ref(torch.randn(3, 8), caps[:, :-1]), caps[:, 1:])
    loss.backward()
    opt.step()
    path = str(tmp_path / "best_model.pth")
    torch.save({"epoch": 3, "model_state_dict": ref.state_dict(), "optimizer_state_dict": opt.state_dict(),
                "scheduler_state_dict": sched.state_dict(), "loss": 1.25,
                # exactly what evaluate_metrics returns (ref train.py:251-264, 280-287): corpus_bleu gives Python
                # floats, the other four are np.mean(...) results, i.e. numpy.float64 scalars in the pickle
                "metrics": {"bleu1": 0.5, "bleu4": 0.125, "meteor": np.mean([0.25, 0.5]), "rouge_l": np.mean([0.5]),
                            "cider": np.mean([1.0, 2.0]), "spice": np.mean([0.125, 0.25])},
                "vocab_size": 50, "vocab_idx2word": {0: "<PAD>", 1: "<START>"}, "initial_lr": 1e-3}, path)  # train.py:485-495
    return ref, opt, path


def test_load_reference_checkpoint_into_model_and_torch_adamw(tmp_path):
    from imagecaption_amd import ImageCaptioningModel
    from imagecaption_amd.checkpoint import load_reference_checkpoint
    ref, ropt, path = _trained_reference(tmp_path)
    model = ImageCaptioningModel(**TINY)
    opt = torch.optim.AdamW(model.parameters(), lr=5e-4, weight_decay=0.01)
    info = load_reference_checkpoint(path, model, opt)
    assert info["epoch"] == 3 and info["vocab_size"] == 50 and info["vocab_idx2word"][1] == "<START>"
    assert info["metrics"]["bleu4"] == 0.125 and float(info["metrics"]["meteor"]) == 0.375  # numpy scalars load
    assert sorted(info["vit_state_dict"]) == sorted(k for k in ref.state_dict() if k.startswith("vit."))
    rsd = ref.state_dict()
    for k, v in model.state_dict().items():
        assert torch.equal(v, rsd[k]), k
    rparams = [p for n, p in ref.named_parameters() if not n.startswith("vit.")]
    for p, rp in zip(model.parameters(), rparams):
        assert torch.equal(opt.state[p]["exp_avg"], ropt.state[rp]["exp_avg"])
        assert torch.equal(opt.state[p]["exp_avg_sq"], ropt.state[rp]["exp_avg_sq"])
    assert opt.param_groups[0]["lr"] == 1e-3  # the checkpoint's, not the constructor's


def test_trainer_round_trip_and_reference_can_load_what_we_save(tmp_path):
    from imagecaption_amd import ImageCaptioningModel
    from imagecaption_amd.checkpoint import load_reference_checkpoint, save_reference_checkpoint
    from imagecaption_amd.trainer import CaptionTrainer
    ref, ropt, path = _trained_reference(tmp_path)
    model = ImageCaptioningModel(**TINY)
    tr = CaptionTrainer(model, dropout=0.0)  # CPU construction: flat blocks only, no kernels run
    info = load_reference_checkpoint(path, model, tr)
    assert tr.step_count == 1 and tr.lr == 1e-3
    rparams = [p for n, p in ref.named_parameters() if not n.startswith("vit.")]
    off = {id(p): o for p, o in tr.flat.placed}
    for p, rp in zip(model.parameters(), rparams):
        o, n = off[id(p)], p.numel()
        assert torch.equal(tr.flat_m[o:o + n].view_as(p), ropt.state[rp]["exp_avg"])
        assert torch.equal(p.data, rp.data)  # parameters are still views of the flat block
        assert p.data.data_ptr() == tr.flat_p[o:o + n].data_ptr()
    # write it back with the ViT entries in front: the reference's strict loads must accept it
    out = str(tmp_path / "resaved.pth")
    save_reference_checkpoint(out, model, tr, None, epoch=4, loss=0.5, metrics={"bleu4": 0.2},
                              vocab_idx2word=info["vocab_idx2word"], initial_lr=1e-3, vit_state=info["vit_state_dict"])
    ck = torch.load(out, map_location="cpu", weights_only=True)
    assert list(ck) == ["epoch", "model_state_dict", "optimizer_state_dict", "scheduler_state_dict", "loss", "metrics",
                        "vocab_size", "vocab_idx2word", "initial_lr"]
    ref2 = FakeRefModel()
    ref2.load_state_dict(ck["model_state_dict"])  # strict (ref main.py:140, test.py:17)
    opt2 = torch.optim.AdamW(ref2.parameters(), lr=1e-3, weight_decay=0.01)
    opt2.load_state_dict(ck["optimizer_state_dict"])  # ref main.py:141
    for (n, p), (_, rp) in zip(ref2.named_parameters(), ref.named_parameters()):
        assert torch.equal(p, rp), n
        if not n.startswith("vit."):
            assert torch.equal(opt2.state[p]["exp_avg"], ropt.state[rp]["exp_avg"]), n
            assert float(opt2.state[p]["step"]) == 1.0
    assert all(p not in opt2.state for n, p in ref2.named_parameters() if n.startswith("vit."))


def test_mismatched_vocab_and_unsafe_files_are_reported(tmp_path):
    from imagecaption_amd import ImageCaptioningModel
    from imagecaption_amd.checkpoint import load_reference_checkpoint
    _, _, path = _trained_reference(tmp_path)
    other = ImageCaptioningModel(**dict(TINY, vocab_size=60))
    with pytest.raises(RuntimeError, match="vocab_size"):
        load_reference_checkpoint(path, other)

    class Evil:
        def __reduce__(self):
            return (print, ("executed",))
    bad = str(tmp_path / "bad.pth")
    torch.save({"model_state_dict": {}, "metrics": Evil()}, bad)
    with pytest.raises(RuntimeError, match="refused"):
        load_reference_checkpoint(bad, ImageCaptioningModel(**TINY))


def _respell_numpy_module(src, dst, old, new):
    """Rewrite a torch.save zip with the pickled module name ``old`` replaced by ``new`` (torch.save pickles with
    protocol 2, whose GLOBAL opcode is newline-terminated text: no length prefix to fix up)."""
    import zipfile
    with zipfile.ZipFile(src) as zin, zipfile.ZipFile(dst, "w", zipfile.ZIP_STORED) as zout:
        for item in zin.infolist():
            data = zin.read(item.filename)
            if item.filename.endswith("data.pkl"):
                assert old in data
                data = data.replace(old, new)
            zout.writestr(item.filename, data)


def test_both_numpy_scalar_spellings_and_non_numeric_numpy_values(tmp_path):
    """The running numpy's spelling of a pickled scalar loads; the OTHER major version's spelling is refused loudly
    (torch keys its allow-list by the constructor's own module name -- see the module docstring); a non-numeric
    numpy value in a file is rejected after the load."""
    from imagecaption_amd import ImageCaptioningModel
    from imagecaption_amd.checkpoint import load_reference_checkpoint
    _, _, path = _trained_reference(tmp_path)
    model = ImageCaptioningModel(**TINY)
    assert float(load_reference_checkpoint(path, model)["metrics"]["cider"]) == 1.5
    own = b"numpy._core.multiarray" if hasattr(np, "_core") else b"numpy.core.multiarray"
    other = b"numpy.core.multiarray" if own == b"numpy._core.multiarray" else b"numpy._core.multiarray"
    old = str(tmp_path / "other_numpy.pth")
    _respell_numpy_module(path, old, own, other)
    with pytest.raises(RuntimeError, match="refused"):
        load_reference_checkpoint(old, model)
    # a numpy string scalar needs no constructor outside the allow-list's reach on some numpy versions: whether the
    # loader refuses it or the post-load check does, it must not come back as a loaded checkpoint
    ck = torch.load(path, map_location="cpu", weights_only=False)  # a file this test wrote itself
    ck["metrics"]["note"] = np.str_("free text")
    odd = str(tmp_path / "odd.pth")
    torch.save(ck, odd)
    with pytest.raises(RuntimeError, match="refused|only numeric numpy"):
        load_reference_checkpoint(odd, model)
