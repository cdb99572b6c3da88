"""C2 training step fed from HOST batches: synchronous .to(device) per step (the reference's loop) vs DevicePrefetcher,
against device-resident inputs (what bench.py times)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagecaption_amd import ImageCaptioningModel
from imagecaption_amd.trainer import CaptionTrainer
from imagecaption_amd.data import DevicePrefetcher

torch.manual_seed(0)
dev = torch.device("cuda")
model = ImageCaptioningModel(10000, 512, 8, 6, feature_dim=2048).to(dev).train()
tr = CaptionTrainer(model, dropout=0.1, seed=1)
g = torch.Generator().manual_seed(1)
host = [(torch.randn(256, 49, 2048, generator=g), torch.randint(5, 10000, (256, 21), generator=g)) for _ in range(4)]
N = 24
batches = [host[i % 4] for i in range(N)]
res = [(f.to(dev), c.to(dev)) for f, c in host]

def run(name, it):
    for f, c in res[:2]: tr.step(f, c)
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
    for f, c in it:
        tr.step(f, c); n += 1
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{name:34s} {dt / n * 1e3:7.2f} ms/step  {256 * n / dt:8.0f} images/s", flush=True)

run("device-resident inputs", (res[i % 4] for i in range(N)))
run("host batches, synchronous .to()", ((f.to(dev), c.to(dev)) for f, c in batches))
pinned = [(f.pin_memory(), c.pin_memory()) for f, c in host]
run("pinned host, synchronous .to()", ((f.to(dev), c.to(dev)) for f, c in (pinned[i % 4] for i in range(N))))
run("host batches, DevicePrefetcher", DevicePrefetcher(batches, dev))
run("pinned host, DevicePrefetcher", DevicePrefetcher([pinned[i % 4] for i in range(N)], dev))
