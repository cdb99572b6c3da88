"""The reference's own training-loop body (ref train.py:24-61: dense one-hot loss, loss.item(), clip_grad_norm_,
torch.optim.AdamW) running UNCHANGED on the drop-in module, against the fused CaptionTrainer, C2 and C1 shapes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from imagecaption_amd import ImageCaptioningModel
from imagecaption_amd.trainer import CaptionTrainer

dev = torch.device("cuda")
for name in ("c2", "c1"):
    cfg = bench.CONFIGS[name]
    f, c = bench.synth_batch(cfg, cfg["batch"], dev, 0)
    torch.manual_seed(0)
    model = ImageCaptioningModel(cfg["vocab"], cfg["d"], cfg["heads"], cfg["layers"], feature_dim=cfg["din"]).to(dev).train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=0.01)

    def ref_step():
        tgt_input, tgt_output = c[:, :-1], c[:, 1:]
        tgt_mask = model.generate_square_subsequent_mask(tgt_input.size(1)).to(dev)
        opt.zero_grad()
        output = model(f, tgt_input, tgt_mask)
        one_hot = torch.zeros_like(output).scatter(2, tgt_output.unsqueeze(-1), 1)
        one_hot = one_hot * 0.9 + 0.1 / output.size(-1)
        loss = -(one_hot * torch.log_softmax(output, dim=-1)).sum(dim=-1).mean()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        return loss.item()

    for _ in range(3): ref_step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): ref_step()
    torch.cuda.synchronize(); t_ref = (time.perf_counter() - t0) / 10
    tr = CaptionTrainer(model, dropout=0.1, seed=1)
    for _ in range(3): tr.step(f, c)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): tr.step(f, c)
    torch.cuda.synchronize(); t_fused = (time.perf_counter() - t0) / 10
    B = cfg["batch"]
    print(f"{name}: reference loop body unchanged {t_ref*1e3:7.2f} ms/step {B/t_ref:8.0f} images/s | CaptionTrainer {t_fused*1e3:7.2f} ms/step {B/t_fused:8.0f} images/s", flush=True)
