"""Small-batch decode (the reference's generate_caption as test.py uses it): host-bound or GPU-bound?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagecaption_amd.models import ImageCaptioningModel
from imagecaption_amd.decode import CaptionDecoder
torch.manual_seed(0)
model = ImageCaptioningModel(108, 512, 8, 6).cuda().eval()
dec = CaptionDecoder(model)
sizes = [int(a) for a in sys.argv[1:]] or [1, 32, 256, 1024]
for n in sizes:
    feats = torch.randn(n, 768, device="cuda")
    dec.sample(feats, max_len=200, seed=1); torch.cuda.synchronize()
    t0 = time.perf_counter(); dec.sample(feats, max_len=200, seed=1); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"C1 sampler L=200, {n:5d} images: host enqueue {(t1-t0)*1e3:7.1f} ms, total {(t2-t0)*1e3:7.1f} ms, {n/(t2-t0):8.0f} captions/s, {(t2-t0)/200*1e3:6.2f} ms/step", flush=True)
