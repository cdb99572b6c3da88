"""Child process of tests/test_model_gpu.py::test_rccl_bucket_collectives_on_one_rank...: a ONE-rank `nccl` (= RCCL)
process group on this GPU, CaptionTrainer steps with the bucket all-reduces really issued on the measured side stream
(CAPDEC_FORCE_COLLECTIVES=1), against the same steps with no collectives.  Prints one JSON line."""
import json
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import bench
    from imagecaption_amd import ImageCaptioningModel, ops
    from imagecaption_amd.trainer import CaptionTrainer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29531"))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    warm = torch.ones(8, device="cuda")
    dist.all_reduce(warm)  # RCCL creates its communicator and streams now, before any lane is measured
    torch.cuda.synchronize()
    cfg = dict(bench.CONFIGS["c2"], batch=64)
    dev = torch.device("cuda", 0)
    f, c = bench.synth_batch(cfg, cfg["batch"], dev, 0)
    calls = []
    real = dist.all_reduce

    def counting(t, *a, **k):
        calls.append((t.numel(), torch.cuda.current_stream().cuda_stream))
        return real(t, *a, **k)

    def run(force):
        if force:
            os.environ["CAPDEC_FORCE_COLLECTIVES"] = "1"
        else:
            os.environ.pop("CAPDEC_FORCE_COLLECTIVES", None)
        torch.manual_seed(42)
        model = ImageCaptioningModel(cfg["vocab"], cfg["d"], cfg["heads"], cfg["layers"], feature_dim=cfg["din"]).to(dev).train()
        tr = CaptionTrainer(model, dropout=0.1, seed=42)
        main_stream = torch.cuda.current_stream().cuda_stream
        del calls[:]
        dist.all_reduce = counting
        try:
            losses = [float(tr.step(f, c).item()) for _ in range(4)]
        finally:
            dist.all_reduce = real
        torch.cuda.synchronize()
        info = dict(active=tr.reducer.active, has_side=tr.reducer.side is not None, buckets=len(tr.bucket_ranges),
                    calls=len(calls), on_side_stream=bool(calls) and all(s != main_stream for _, s in calls),
                    elements=sum(n for n, _ in calls), total=tr.flat.total,
                    side_is_lane=tr.reducer.side is not None and tr.side is not None
                    and tr.reducer.side.cuda_stream != tr.side.stream.cuda_stream)
        return tr.flat_p.clone(), losses, info

    p1, l1, i1 = run(True)
    lanes = {str(k[2]): v for k, v in ops.LANE_INFO.items()}
    p0, l0, i0 = run(False)
    out = dict(bit_equal=bool(torch.equal(p0, p1)), losses_equal=l0 == l1, forced=i1, plain=i0, lanes=lanes,
               finite=all(x == x for x in l1), backend=dist.get_backend())
    dist.destroy_process_group()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
