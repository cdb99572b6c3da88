"""imagecaption_amd -- MI355X-native caption-decoder hot path (drop-in for EricMoin/ImageCaption's models.py).

Host side is Python on PyTorch-ROCm (device memory, streams, torch.distributed); all arithmetic of
the path runs in hand-written gfx950 HIP kernels behind the C ABI of include/capdec.h.
"""
from .models import ImageCaptioningModel, PositionalEncoding  # noqa: F401

__all__ = ["ImageCaptioningModel", "PositionalEncoding"]
