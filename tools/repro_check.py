"""Two runs of N bf16-storage C2 training steps from one seed: report the first step whose loss differs (GPU only).
   python tools/repro_check.py [side=0|1] [one_call=0|1] [steps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from imagecaption_amd import ImageCaptioningModel, ops
from imagecaption_amd.trainer import CaptionTrainer
side = int(sys.argv[1]) if len(sys.argv) > 1 else 1
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
cfg = bench.CONFIGS["c2"]
dev = torch.device("cuda")
f, c = bench.synth_batch(cfg, cfg["batch"], dev, 0)
def run():
    torch.manual_seed(42)
    model = ImageCaptioningModel(cfg["vocab"], cfg["d"], cfg["heads"], cfg["layers"], feature_dim=cfg["din"]).to(dev).train()
    tr = CaptionTrainer(model, dropout=0.1, seed=42, dtype="bf16s", side_stream_wgrad=bool(side))
    if len(sys.argv) > 2 and sys.argv[2] == "0":
        tr.one_call_layers = False
    losses = [float(tr.step(f, c).item()) for _ in range(steps)]
    torch.cuda.synchronize()
    return tr.flat_p.clone(), tr.flat_g.clone() if hasattr(tr, "flat_g") else None, losses
p1, g1, l1 = run()
p2, g2, l2 = run()
print("side", side, "losses equal", l1 == l2, "params equal", bool(torch.equal(p1, p2)))
print(l1); print(l2)
