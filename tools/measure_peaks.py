"""Sustained peaks of this box: fp32 / bf16 MFMA rate (register-resident loop on every SIMD) and streaming-copy
HBM bandwidth.  Prints one JSON line; bench.py states these next to the nominal roofline denominators."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagecaption_amd import _capi
from imagecaption_amd._capi import check, lib, ptr, stream


def measure():
    l = lib()
    out = torch.zeros(16, device="cuda")
    res = {}
    for name, dt, iters in (("mfma_f32_tflops", _capi.F32, 20000), ("mfma_bf16_tflops", _capi.BF16, 40000)):
        fl = C.c_double(0.0)
        best = 0.0
        for blocks in (1024, 2048):  # 4 / 8 waves per SIMD
            check(l.capdec_probe_mfma(dt, blocks, 200, ptr(out), C.byref(fl), stream()), "probe")
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            check(l.capdec_probe_mfma(dt, blocks, iters, ptr(out), C.byref(fl), stream()), "probe")
            e1.record()
            torch.cuda.synchronize()
            best = max(best, fl.value / (e0.elapsed_time(e1) * 1e-3) / 1e12)
        res[name] = best
    n = 2 << 30
    src = torch.empty(n, dtype=torch.uint8, device="cuda").fill_(1)
    dst = torch.empty_like(src)
    best = 0.0
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(l.capdec_probe_copy(ptr(src), ptr(dst), n, stream()), "copy")
        e1.record()
        torch.cuda.synchronize()
        best = max(best, 2.0 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    res["hbm_copy_gbs_read_plus_write"] = best
    return res


if __name__ == "__main__":
    print(json.dumps(measure()))
