"""cProfile of the host side of one training step (where does the enqueue time go?).  usage: host_profile.py [dtype] [config]"""
import cProfile, pstats, sys, io
sys.path.insert(0, ".")
import torch
import bench
from imagecaption_amd import ImageCaptioningModel
from imagecaption_amd.trainer import CaptionTrainer

dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16s"
cfg = bench.CONFIGS[sys.argv[2] if len(sys.argv) > 2 else "c2"]
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = ImageCaptioningModel(cfg["vocab"], cfg["d"], cfg["heads"], cfg["layers"], feature_dim=cfg["din"]).to(dev)
tr = CaptionTrainer(model, dtype=dtype)
f, c = bench.synth_batch(cfg, cfg["batch"], dev, 0)
for _ in range(5):
    tr.step(f, c)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    tr.step(f, c)
pr.disable()
torch.cuda.synchronize()
out = io.StringIO()
st = pstats.Stats(pr, stream=out)
st.sort_stats("tottime").print_stats(35)
print(out.getvalue())
