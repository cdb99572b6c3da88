"""Decode leg alone (beam-3 on N images, length 20) for rocprofv3 --kernel-trace: python3 tools/decode_prof.py [images] [mode]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagecaption_amd.models import ImageCaptioningModel
from imagecaption_amd.decode import CaptionDecoder

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
mode = sys.argv[2] if len(sys.argv) > 2 else "beam3"
torch.manual_seed(0)
model = ImageCaptioningModel(10000, 512, 8, 6, feature_dim=2048).cuda().eval()
feats = torch.randn(n, 49, 2048, device="cuda")
dec = CaptionDecoder(model)
def run():
    if mode == "beam3": return dec.beam(feats, 3, 20)
    if mode == "beam5": return dec.beam(feats, 5, 20)
    if mode == "greedy": return dec.greedy(feats, 20)
    return dec.sample(feats, max_len=20, seed=1)
run(); torch.cuda.synchronize()
t0 = time.perf_counter(); run(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"{mode} {n} images: {dt*1e3:.1f} ms, {n/dt:.0f} captions/s", flush=True)
