#!/usr/bin/env python3
"""Headline benchmark: teacher-forced training throughput of the caption decoder (images/s).

    python bench.py --gpus N --steps K --warmup W

Workload at every N: BASELINE.json configs[1] ("C2") per GPU -- synthetic 7x7x2048 grid features
(S = 49 memory positions), hidden 512, 8 heads, 6 layers, vocab 10k, max len 20, batch 256 per GPU,
fp32, dropout 0.1 ACTIVE (the reference trains in model.train(), ref train.py:15), full step =
forward + label-smoothed loss + backward + gradient all-reduce (N > 1) + global-norm clip + AdamW.
Weak scaling: per-GPU batch is fixed; value = all ranks' images / max-over-ranks time.

One JSON line on rank 0 (contract in the task statement), plus:
  roofline     -- the dominant kernel (fp32 MFMA GEMM): algorithmic FLOPs of every GEMM launch of
                  one step / their summed launch durations, measured live with HIP events on the
                  launch stream in a separate instrumented step ("Measurement").
  cpu_baseline -- the oracle (stock torch.nn on the host cores, the reference's own CPU path) on
                  the same C2 step, bounded sample (N = 1, rank 0 only).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (vocab, d_model, heads, layers, feature_dim, S, T, batch)
    "c2": dict(vocab=10000, d=512, heads=8, layers=6, din=2048, S=49, T=20, batch=256),
    "c1": dict(vocab=108, d=512, heads=8, layers=6, din=768, S=1, T=299, batch=32),
    "c3": dict(vocab=30000, d=1024, heads=8, layers=6, din=512, S=196, T=20, batch=256),
    "tiny": dict(vocab=50, d=64, heads=2, layers=2, din=24, S=5, T=12, batch=8),
}
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 MFMA
PEAK_HBM_GBS = 8000.0


def synth_batch(cfg, batch, device, rank):
    g = torch.Generator().manual_seed(1234 + rank)
    S, din, T, V = cfg["S"], cfg["din"], cfg["T"], cfg["vocab"]
    feats = torch.randn((batch, din) if S == 1 else (batch, S, din), generator=g)
    ids = torch.randint(5, V, (batch, T + 1), generator=g)
    ids[:, 0] = 1
    lens = torch.randint(min(8, T), T + 1, (batch,), generator=g)
    pos = torch.arange(T + 1).unsqueeze(0)
    ids = torch.where(pos == lens.unsqueeze(1), torch.full_like(ids, 2), ids)
    ids = torch.where(pos > lens.unsqueeze(1), torch.zeros_like(ids), ids)
    return feats.to(device), ids.to(device)


def c1_batch(cfg, batch, device, rank, split="train"):
    """BASELINE.json configs[0] (C1): the reference's own captions -- the committed uint8 token fixture made from
    data/train_captions.json by oracle/gen_golden.py, every row padded to max_len 300 with <PAD> (ref main.py:36,
    dataset.py:213-228) -- and seeded synthetic [B, 768] features in the image slot (the ViT is out of scope)."""
    import numpy as np
    z = np.load(os.path.join(ROOT, "tests", "golden", f"captions_{split}.npz"))
    lens = z["lengths"].astype(np.int64)
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
    T1 = cfg["T"] + 1
    ids = torch.zeros((batch, T1), dtype=torch.int64)
    for i in range(batch):
        j = (rank * batch + i) % len(lens)
        n = min(int(lens[j]), T1)
        ids[i, :n] = torch.from_numpy(z["tokens"][starts[j]:starts[j] + n].astype(np.int64))
    g = torch.Generator().manual_seed(42 + rank)
    return torch.randn((batch, cfg["din"]), generator=g).to(device), ids.to(device)


def make_batch(name, cfg, batch, device, rank):
    return c1_batch(cfg, batch, device, rank) if name == "c1" else synth_batch(cfg, batch, device, rank)


def train_flops_per_image(cfg):
    """forward FLOPs (2*MAC) per image; training = 3x."""
    L, T, S, d, V, din = cfg["layers"], cfg["T"], cfg["S"], cfg["d"], cfg["vocab"], cfg["din"]
    fwd = L * (28 * T * d * d + 4 * S * d * d + 2 * d * T * (T + 1) + 4 * T * S * d) + (4 * T * d * d + 4 * T * d * V) \
        + S * (4 * din * d + 4 * d * d)
    return 3.0 * fwd


def decode_flops_per_image(cfg, W, T):
    """SURVEY.md section 8d, KV-cached decode with W hypotheses for T steps, FLOPs = 2*MAC.  Once per image: the
    memory projection and the L cross K/V projections.  Per hypothesis and step: per layer 28 d^2 of projections
    (self qkv 6, self out 2, cross q 2, cross out 2, FFN 16), 4 S d of cross attention and ~4 (T/2) d of self
    attention over the cache; then the head, 4 d^2 + 4 d V.  C2, W=5, T=20: 7.2 GFLOP per image."""
    L, S, d, V, din = cfg["layers"], cfg["S"], cfg["d"], cfg["vocab"], cfg["din"]
    once = S * (4 * din * d + 4 * d * d) + L * 4 * S * d * d
    per_tok = L * (2 * d * 3 * d + 2 * d * d + 2 * d * d + 2 * d * d + 16 * d * d + 4 * S * d + 4 * (T / 2.0) * d) \
        + (4 * d * d + 4 * d * V)
    return once + W * T * per_tok


def cpu_train(name, cfg, steps, warmup, threads):
    """`steps` timed training steps of the oracle (stock torch.nn, the reference's CPU path) on the host cores."""
    from oracle.torch_oracle import OracleCaptionDecoder, train_step
    torch.manual_seed(42)
    model = OracleCaptionDecoder(cfg["vocab"], cfg["d"], cfg["heads"], cfg["layers"], cfg["din"]).train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=0.01)
    feats, ids = make_batch(name, cfg, cfg["batch"], "cpu", 0)
    for _ in range(warmup):
        train_step(model, opt, feats, ids)
    t0 = time.perf_counter()
    for _ in range(steps):
        train_step(model, opt, feats, ids)
    dt = (time.perf_counter() - t0) / steps
    return model, feats, {"value": cfg["batch"] / dt, "unit": "images/s", "cores": threads, "kind": "port",
                          "sample": f"{steps} timed training steps (after {warmup} warm-up) of the same workload, batch "
                                    f"{cfg['batch']}, stock torch.nn fp32 on {threads} host threads, {dt:.2f} s/step"}


def cpu_baseline(name, cfg, extra=True, full=False):
    """The reference's CPU path (oracle/torch_oracle.py) on this box's host cores, BASELINE.md's plan on bounded
    samples: the bench workload itself (3 steps) is the `cpu_baseline` object; `also` holds C1 -- 10 training steps at
    batch 32 on the reference's own captions, then greedy decode of a test slice -- and, with --cpu-full, 3 C3 steps
    and the CPU beam-5 restatement on 256 images (minutes of CPU time: not part of the default run)."""
    from oracle.torch_oracle import beam_decode, greedy_decode
    threads = min(16, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    model, feats, out = cpu_train(name, cfg, 3, 1, threads)
    also = {}
    if full:
        n = 256
        f = feats[:n] if feats.shape[0] >= n else feats
        t0 = time.perf_counter()
        beam_decode(model, f, beam=5, max_len=20)
        dt = time.perf_counter() - t0
        also["beam5_captions_per_sec"] = {"value": f.shape[0] / dt, "unit": "captions/s", "cores": threads,
                                          "sample": f"CPU beam-5 restatement, length 20, {f.shape[0]} images of the bench "
                                                    f"workload's shape, {dt:.1f} s"}
        c3 = dict(CONFIGS["c3"])
        also["c3_train"] = cpu_train("c3", c3, 3, 0, threads)[2]
    if extra and name != "c1":
        c1 = dict(CONFIGS["c1"])
        m1, f1, r1 = cpu_train("c1", c1, 10, 2, threads)
        also["c1_train"] = r1
        n = 8
        t0 = time.perf_counter()
        greedy_decode(m1, f1[:n], max_len=20)
        dt = time.perf_counter() - t0
        also["c1_greedy_captions_per_sec"] = {"value": n / dt, "unit": "captions/s", "cores": threads,
                                              "sample": f"oracle greedy decode (full-prefix recompute, as the reference's "
                                                        f"loop does), {n} images, length 20, {dt:.2f} s"}
    if also:
        out["also"] = also
    return out


def decode_bench(model, cfg, dev, rank, world, dtype="fp32", images=4096, max_len=20):
    """Beam-3 / beam-5 / greedy / reference-sampler captions/s, KV-cached engine, images sharded per rank, no collective."""
    from imagecaption_amd.decode import CaptionDecoder
    dec = CaptionDecoder(model, dtype=dtype)
    feats, _ = synth_batch(cfg, images, dev, 100 + rank)
    res = {}
    for name, fn in (("beam3", lambda: dec.beam(feats, beam=3, max_len=max_len)),
                     ("beam5", lambda: dec.beam(feats, beam=5, max_len=max_len)),
                     ("greedy", lambda: dec.greedy(feats, max_len=max_len)),
                     # the reference's own decode (train.py:289-412: temperature 0.7, top-5 draw, sentence rules)
                     ("sampler", lambda: dec.sample(feats, max_len=max_len + 1, seed=1))):
        fn()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        res[f"{name}_captions_per_sec"] = images * world / dt
        W = {"beam3": 3, "beam5": 5}.get(name, 1)
        tf = decode_flops_per_image(cfg, W, max_len) * images / dt / 1e12  # per GPU (dt is the slowest rank's)
        peak = mfma_peak(dtype)
        res[f"{name}_roofline"] = {"bound": "mfma", "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak}
    res["images_per_gpu"], res["max_len"] = images, max_len
    model.train()
    return res


def mfma_peak(dtype):
    """Dense matrix peak in ALGORITHMIC (2*M*N*K) TFLOP/s: the split-fp32 mode issues six bf16 products per
    fp32 product, so its ceiling is the bf16 peak / 6."""
    return {"fp32": PEAK_F32_MFMA_TFLOPS, "bf16": PEAK_BF16_MFMA_TFLOPS, "bf16s": PEAK_BF16_MFMA_TFLOPS,
            "f32x3": PEAK_BF16_MFMA_TFLOPS / 6.0}[dtype]


def measured_mfma_peak(dtype):
    """Sustained matrix rate of THIS box from a register-resident MFMA loop (csrc/probe.hip), TFLOP/s."""
    import ctypes as C
    from imagecaption_amd import _capi
    from imagecaption_amd._capi import check, lib, ptr, stream
    out = torch.zeros(16, device="cuda")
    fl = C.c_double(0.0)
    dt = _capi.F32 if dtype == "fp32" else _capi.BF16
    check(lib().capdec_probe_mfma(dt, 2048, 200, ptr(out), C.byref(fl), stream()), "capdec_probe_mfma")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(lib().capdec_probe_mfma(dt, 2048, 10000, ptr(out), C.byref(fl), stream()), "capdec_probe_mfma")
    e1.record()
    torch.cuda.synchronize()
    return fl.value / (e0.elapsed_time(e1) * 1e-3) / 1e12


def vocab_traffic(config, dtype="fp32"):
    """HBM-side bytes per launch of the vocab GEMM from the committed PMC passes (profiles/): C2 fp32 and C3 bf16 storage."""
    if config == "c3" and dtype == "bf16s":
        path = os.path.join(ROOT, "profiles", "r03c_vocab_gemm_bf16s_traffic.json")
    elif config == "c2" and dtype == "fp32":
        path = os.path.join(ROOT, "profiles", "r02_vocab_gemm_traffic.json")
        if not os.path.exists(path):
            path = os.path.join(ROOT, "profiles", "r01_vocab_gemm_traffic.json")
    else:
        return None
    if not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f)["traffic_bytes"]


def profile_step(trainer, feats, ids, reps=3):
    """``reps`` instrumented steps: HIP-event duration of every GEMM / attention launch (on the launch
    stream).  Returned counts / work / seconds are per step (means over the reps)."""
    from imagecaption_amd import ops
    side, trainer.side = trainer.side, None  # serialise on one stream: per-launch durations without overlap
    try:
        trainer.step(feats, ids)  # the single-stream step allocates differently: one untimed pass first
        torch.cuda.synchronize()
        ops.PROFILE = []
        for _ in range(reps):
            trainer.step(feats, ids)
        torch.cuda.synchronize()
    finally:
        trainer.side = side
    rec, ops.PROFILE = ops.PROFILE, None
    agg = {}
    for kind, work, nbytes, e0, e1 in rec:
        cls = kind.split("_")[0] + "_" + kind.split("_")[1]
        a = agg.setdefault(cls, [0, 0.0, 0.0, 0.0])
        a[0] += 1
        a[1] += work
        a[2] += nbytes
        a[3] += e0.elapsed_time(e1) * 1e-3
    detail = {}
    for kind, work, nbytes, e0, e1 in rec:
        d = detail.setdefault(kind, [0, 0.0, 0.0, 0.0])
        d[0] += 1
        d[1] += work
        d[2] += nbytes
        d[3] += e0.elapsed_time(e1) * 1e-3
    for table in (agg, detail):
        for v in table.values():
            v[0] //= reps
            v[1], v[2], v[3] = v[1] / reps, v[2] / reps, v[3] / reps
    return agg, detail


def measured_hbm_copy_gbs(n=1 << 30):
    """Read + write rate of a streaming copy of ``n`` bytes on THIS box (csrc/probe.hip), GB/s: what an HBM-bound
    kernel can reach here, next to the nominal 8 TB/s the roofline fractions are quoted on.  1 GiB streams from HBM;
    a copy of a launch's own working set (tens of MiB) runs out of the 256 MiB Infinity Cache, as the attention
    launches of a step do."""
    from imagecaption_amd._capi import check, lib, ptr, stream
    n = max(1 << 20, int(n) // 256 * 256)
    src = torch.empty(n, dtype=torch.uint8, device="cuda")
    dst = torch.empty(n, dtype=torch.uint8, device="cuda")
    src.zero_()
    best = 0.0
    reps = 1 if n >= (1 << 28) else 8  # short copies: several launches between the events
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            check(lib().capdec_probe_copy(ptr(src), ptr(dst), n, stream()), "capdec_probe_copy")
        e1.record()
        torch.cuda.synchronize()
        best = max(best, 2.0 * n * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    return best


def profile_marker():
    """A one-workgroup copy: marks the start of the measured part of the run in a rocprofv3 kernel trace
    (tools/rocpd_stats.py --timed counts what follows it)."""
    from imagecaption_amd._capi import check, lib, ptr, stream
    buf = torch.zeros(64, dtype=torch.uint8, device="cuda")
    check(lib().capdec_probe_copy(ptr(buf), ptr(buf[32:]), 16, stream()), "capdec_probe_copy")


def timed_steps(trainer, feats, ids, steps, warmup, world, dev):
    """`warmup` untimed steps, then exactly `steps` steps between barrier + synchronize; MAX over ranks."""
    for _ in range(warmup):
        trainer.step(feats, ids)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = trainer.step(feats, ids)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, loss


def roofline_block(cfg, dtype, agg, detail, traffic=None, with_peaks=True):
    """`roofline` (the dominant launch: vocabulary-projection forward), `gemm_all` and `attention` from the
    per-launch HIP-event records of profile_step."""
    gemm_flops = sum(a[1] for k, a in agg.items() if k.startswith("gemm"))
    gemm_time = sum(a[3] for k, a in agg.items() if k.startswith("gemm"))
    gemm_n = sum(a[0] for k, a in agg.items() if k.startswith("gemm"))
    ach = gemm_flops / gemm_time / 1e12 if gemm_time > 0 else 0.0
    attn = {k: a for k, a in agg.items() if k.startswith("attn")}
    attn_b = sum(a[2] for a in attn.values())
    attn_t = sum(a[3] for a in attn.values())
    peak = mfma_peak(dtype)
    vk = f"gemm_nt_{cfg['batch'] * cfg['T']}x{cfg['vocab']}x{2 * cfg['d']}"
    vocab = detail.get(vk)
    kname = {"fp32": "f32", "f32x3": "f32x3", "bf16": "bf16", "bf16s": "bf16s"}[dtype]
    roof = None
    if vocab:
        roof = {"bound": "mfma", "achieved": vocab[1] / vocab[3] / 1e12, "peak": peak, "unit": "TFLOP/s",
                "frac": vocab[1] / vocab[3] / 1e12 / peak, "traffic": traffic,
                "kernel": f"gemm_{kname}_kernel, vocab projection forward {vk[8:]} with the cross-entropy partials in its "
                          f"epilogue (one launch per step)",
                "ms": vocab[3] * 1e3 / vocab[0]}
        if with_peaks:
            roof["peak_measured"] = measured_mfma_peak("fp32" if dtype == "fp32" else "bf16") / (6.0 if dtype == "f32x3" else 1.0)
    att = {"bound": "hbm", "achieved": attn_b / attn_t / 1e9 if attn_t else 0.0, "peak": PEAK_HBM_GBS, "unit": "GB/s",
           "frac": (attn_b / attn_t / 1e9 / PEAK_HBM_GBS) if attn_t else 0.0, "ms_per_step": attn_t * 1e3,
           "launches_per_step": sum(a[0] for a in attn.values())}
    if with_peaks:
        att["peak_measured"] = measured_hbm_copy_gbs()
        # the ceiling for THESE launches: a copy that moves as many bytes as the average attention launch (read + write
        # together), i.e. of half that size -- tens of MiB, served by the Infinity Cache between launches
        n_launch = sum(a[0] for a in attn.values())
        if n_launch:
            att["bytes_per_launch"] = attn_b / n_launch
            att["peak_measured_working_set"] = measured_hbm_copy_gbs(attn_b / n_launch / 2)
            att["frac_of_working_set_copy"] = att["achieved"] / att["peak_measured_working_set"]
    return {"roofline": roof,
            "gemm_all": {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                         "launches_per_step": gemm_n, "ms_per_step": gemm_time * 1e3},
            "attention": att}


def run_leg(name, dtype, steps, warmup, dev, rank, world, args, decode_images=0):
    """Another BASELINE configuration timed by the same harness (its own model, trainer, measured GEMM plans and
    roofline); never the headline `value`."""
    from imagecaption_amd import ImageCaptioningModel
    from imagecaption_amd.trainer import CaptionTrainer
    cfg = dict(CONFIGS[name])
    # every leg / mode starts from fresh allocator segments, as a run of its own would: where a buffer lands inside a
    # recycled block moves the bf16-storage products by up to 35 % (measured: the bf16s mode 6.2 ms alone, 8.4 ms as
    # the third mode of one process without this call)
    torch.cuda.empty_cache()
    torch.manual_seed(42)
    model = ImageCaptioningModel(cfg["vocab"], cfg["d"], cfg["heads"], cfg["layers"], feature_dim=cfg["din"]).to(dev)
    model.train()
    tr = CaptionTrainer(model, dropout=0.0 if args.no_dropout else 0.1, seed=42, dtype=dtype,
                        side_stream_wgrad=not args.no_side_streams)
    feats, ids = make_batch(name, cfg, cfg["batch"], dev, rank)
    if not args.no_tune:
        tr.tune(feats, ids)
    dt, loss = timed_steps(tr, feats, ids, steps, warmup, world, dev)
    agg, detail = profile_step(tr, feats, ids, reps=2)
    value = cfg["batch"] * world * steps / dt
    out = {"metric": "train_images_per_sec", "value": value, "unit": "images/s", "ms_per_step": dt / steps * 1e3,
           "steps": steps, "warmup": warmup, "dtype": {"fp32": "f32", "f32x3": "f32x3", "bf16": "bf16 products, fp32 storage",
                                              "bf16s": "bf16 (products read bf16 copies of weights / activations / gradients; fp32 "
                                                       "accumulate, norms, attention, loss, master weights and optimizer)"}[dtype],
           "loss": float(loss.item()), "params": tr.n_params,
           "config": {"workload": LEG_WORKLOADS[name], "global_batch": cfg["batch"] * world, "seq_len": cfg["T"],
                      "memory_len": cfg["S"]},
           "step_tflops_algorithmic": train_flops_per_image(cfg) * value / 1e12}
    out.update(roofline_block(cfg, dtype, agg, detail, traffic=vocab_traffic(name, dtype), with_peaks=False))
    dec = None
    if decode_images:
        # decoding at this leg's dimensions (fp32, as the headline's decode numbers): for C3 the 30 000-word vocabulary
        # projection and beam selection (BASELINE configs[4]-style decoding on the large-vocabulary shape)
        del tr
        torch.cuda.empty_cache()
        dec = decode_bench(model, cfg, dev, rank, world, dtype="fp32", images=decode_images, max_len=cfg["T"])
        dec["config"] = {"workload": LEG_WORKLOADS[name].split(":")[0] + " dimensions, KV-cached decode, fp32", "beam": [3, 5]}
        tr = None
    del tr, model
    torch.cuda.empty_cache()
    return (out, dec) if decode_images else out


LEG_WORKLOADS = {
    "c1": "BASELINE.json configs[0] (C1): the repo's own model shape (d 512, 6 layers, vocab 108, max_len 300, memory of ONE "
          "projected 768-d vector), batch 32, the reference's training captions (committed token fixture), synthetic features",
    "c3": "BASELINE.json configs[2] (C3): 14x14x512 grid features (196 memory positions), hidden 1024, vocab 30000, len 20, "
          "batch 256, bf16 matrix products",
}


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n, argv):
    """``python bench.py --gpus N`` without a launcher: start N rank processes (one per GPU) through
    ``torch.distributed.run`` as FRESH children -- this parent has made no GPU call and makes none -- relay
    their stderr, and print rank 0's single JSON line.  Returns the exit code."""
    import subprocess
    port = os.environ.get("MASTER_PORT") or str(_free_port())
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.pop("MASTER_PORT", None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this pool (RCCL needs it)
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and ln.rstrip().endswith("}")]
    if r.returncode != 0 or not lines:
        sys.stderr.write(r.stdout)
        print(f"bench.py: the {n}-rank run failed (exit code {r.returncode}, {len(lines)} result line(s))", file=sys.stderr)
        return r.returncode or 1
    print(lines[-1], flush=True)
    return 0


def launch_check(world, rank, saved_stdout):
    """--launch-check: every rank joins the process group, takes part in one all-reduce and rank 0 reports the
    group size -- the launcher / rendezvous path of ``--gpus N`` without any kernel (runs on CPU with gloo)."""
    use_gpu = torch.cuda.device_count() > 0 and os.environ.get("CAPDEC_DIST_BACKEND", "nccl") == "nccl"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if use_gpu:
            lr = int(os.environ.get("LOCAL_RANK", "0"))
            torch.cuda.set_device(lr)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", lr))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    ones = torch.ones(4, device="cuda" if (use_gpu and world > 1) else "cpu")
    if world > 1:
        dist.all_reduce(ones)
    ranks = dist.get_world_size() if world > 1 else 1
    ok = bool((ones == float(world)).all().item())
    if world > 1:
        dist.barrier()
        backend = dist.get_backend()
        dist.destroy_process_group()
    else:
        backend = None
    if rank == 0:
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps({"metric": "launch_check", "n_gpus": world, "rccl_ranks": ranks, "backend": backend,
                          "allreduce_ok": ok}), flush=True)
    return 0 if ok else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the config's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dropout", action="store_true", help="eval-mode step (parity runs only; not the metric)")
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "f32x3", "bf16", "bf16s"],
                    help="fp32 = reference precision (the headline); bf16 = bf16 MFMA products, fp32 elsewhere")
    ap.add_argument("--no-decode", action="store_true")
    ap.add_argument("--no-side-streams", action="store_true", help="single-stream step (diagnostic)")
    ap.add_argument("--decode-images", type=int, default=12500,
                    help="images per GPU for the decode leg (BASELINE configs[4]: 100k images over 8 GPUs)")
    ap.add_argument("--detail", action="store_true", help="print the per-shape kernel table to stderr")
    ap.add_argument("--modes", default="f32x3,bf16,bf16s", help="the optional arithmetic modes timed after the fp32 run")
    ap.add_argument("--no-modes", action="store_true", help="skip the f32x3 / bf16 / bf16s legs of the default fp32 run")
    ap.add_argument("--c3-decode-images", type=int, default=2048, help="images per GPU of the C3-dimension decode leg")
    ap.add_argument("--no-legs", action="store_true", help="skip the C1 (fp32) and C3 (bf16) legs of the default run")
    ap.add_argument("--cpu-full", action="store_true", help="CPU baseline also for C3 (3 steps) and beam-5 (256 images): minutes")
    ap.add_argument("--no-tune", action="store_true", help="keep the cost model's GEMM plans (skip the measured-plan set-up)")
    ap.add_argument("--launch-check", action="store_true",
                    help="only start the ranks, all-reduce once and report the group size (no kernels; works on CPU)")
    args = ap.parse_args()
    # `python bench.py --gpus N` (N > 1) outside a launcher: start the N ranks here.  Nothing above or in
    # launch_ranks touches the GPU, so the children are fresh processes of a parent without a HIP context.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    # The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a version banner when its
    # first communicator is built), so file descriptor 1 points at stderr for the whole run and is restored only
    # to print the result line.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    if os.environ.get("CAPDEC_HANG_DUMP"):  # debugging aid: dump every thread's stack after N seconds and exit
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["CAPDEC_HANG_DUMP"]), exit=True)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if args.launch_check:
        sys.exit(launch_check(world, rank, saved_stdout))
    # rehearsal hooks (tests only): CAPDEC_DIST_BACKEND=gloo + CAPDEC_SINGLE_DEVICE=1 run N ranks on one GPU
    backend = os.environ.get("CAPDEC_DIST_BACKEND", "nccl")
    if os.environ.get("CAPDEC_SINGLE_DEVICE") == "1":
        local_rank = 0
    forced = os.environ.get("CAPDEC_FORCE_COLLECTIVES") == "1"  # one-rank group: exercises the RCCL path on 1 GPU
    if forced and world == 1:
        os.environ.setdefault("MASTER_PORT", "29555")
    if world > 1 or forced:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the capdec ops have no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from imagecaption_amd import ImageCaptioningModel
    from imagecaption_amd.trainer import CaptionTrainer

    cfg = dict(CONFIGS[args.config])
    if args.batch:
        cfg["batch"] = args.batch
    torch.manual_seed(42)
    model = ImageCaptioningModel(cfg["vocab"], cfg["d"], cfg["heads"], cfg["layers"], feature_dim=cfg["din"]).to(dev)
    model.train()
    trainer = CaptionTrainer(model, dropout=0.0 if args.no_dropout else 0.1, seed=42, dtype=args.dtype,
                             side_stream_wgrad=not args.no_side_streams)
    feats, ids = make_batch(args.config, cfg, cfg["batch"], dev, rank)
    # one-time set-up, outside every timed region: measure the GEMM plans of this step's shapes on this device
    tuned = {} if args.no_tune else trainer.tune(feats, ids, verbose=args.detail)
    profile_marker()

    dt, loss = timed_steps(trainer, feats, ids, args.steps, args.warmup, world, dev)
    loss_val = float(loss.item())
    decode = None if args.no_decode else decode_bench(model, cfg, dev, rank, world, dtype=args.dtype, images=args.decode_images,
                                                      max_len=min(cfg["T"], 20))

    # the instrumented step contains the gradient all-reduce: every rank must run it
    agg, detail = profile_step(trainer, feats, ids)
    # the same step in the optional arithmetic modes (never the headline `value`; N = 1 only, like cpu_baseline)
    modes = {}
    if args.dtype == "fp32" and not args.no_modes and world == 1:
        for mode in [m for m in args.modes.split(",") if m]:
            torch.cuda.empty_cache()  # fresh segments, as a run of its own would have (see run_leg)
            torch.manual_seed(42)
            m2 = ImageCaptioningModel(cfg["vocab"], cfg["d"], cfg["heads"], cfg["layers"], feature_dim=cfg["din"]).to(dev)
            m2.train()
            t2 = CaptionTrainer(m2, dropout=0.0 if args.no_dropout else 0.1, seed=42, dtype=mode,
                                side_stream_wgrad=not args.no_side_streams)
            if not args.no_tune:
                t2.tune(feats, ids)
            dt2, loss2 = timed_steps(t2, feats, ids, args.steps, args.warmup, world, dev)
            modes[mode if mode not in modes else mode + "#2"] = {"value": cfg["batch"] * world * args.steps / dt2, "unit": "images/s",
                           "ms_per_step": dt2 / args.steps * 1e3, "loss": float(loss2.item()),
                           "arithmetic": {"f32x3": "fp32 operands split into 3 bf16 terms, six bf16 MFMA products, fp32-level accuracy",
                                          "bf16": "bf16 MFMA products, fp32 accumulate / storage / optimizer",
                                          "bf16s": "bf16 storage of every matrix-product and attention operand (the C3 "
                                                   "leg's mode), fp32 accumulate / norms / loss / master weights / optimizer"}[mode]}
            del t2, m2
    # the other BASELINE configurations, timed by the same harness in the same run (N = 1 only, like cpu_baseline)
    legs = {}
    if args.config == "c2" and args.dtype == "fp32" and not args.no_legs and world == 1:
        legs["c1_fp32"] = run_leg("c1", "fp32", max(5, args.steps // 2), 3, dev, rank, world, args)
        if args.no_decode:
            legs["c3_bf16"] = run_leg("c3", "bf16s", max(5, args.steps // 2), 3, dev, rank, world, args)
        else:
            legs["c3_bf16"], legs["c3_decode"] = run_leg("c3", "bf16s", max(5, args.steps // 2), 3, dev, rank, world, args,
                                                         decode_images=args.c3_decode_images)
    out = None
    if rank == 0:
        ms = dt / args.steps * 1e3
        value = cfg["batch"] * world * args.steps / dt
        out = {
            "metric": "train_images_per_sec", "value": value, "unit": "images/s", "n_gpus": world,
            # size of the process group the gradient all-reduce really ran on (RCCL when backend == "nccl")
            "rccl_ranks": dist.get_world_size() if dist.is_initialized() else 1,
            "dist_backend": dist.get_backend() if dist.is_initialized() else None,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": {"fp32": "f32", "f32x3": "f32 (3 x bf16 split operands, six bf16 MFMA products)", "bf16": "bf16", "bf16s": "bf16"}[args.dtype], "data": "synthetic",
            "config": {"workload": f"BASELINE.json configs[1] (C2): 7x7x2048 grid features, hidden {cfg['d']}, "
                                   f"{cfg['layers']} layers, vocab {cfg['vocab']}, len {cfg['T']}, batch "
                                   f"{cfg['batch']}/GPU, teacher-forced train step incl. dropout, clip, AdamW"
                       if args.config == "c2" else LEG_WORKLOADS.get(args.config, args.config),
                       "global_batch": cfg["batch"] * world, "seq_len": cfg["T"], "memory_len": cfg["S"],
                       "parallelism": f"dp{world}", "params": trainer.n_params},
            "loss": loss_val,
            "gemm_plans_tuned": sum(1 for k, v in tuned.items() if k != "ce_backward" and v[0] is not None),
            "ce_backward": trainer.ce_backward,
            "modes": modes,
            "legs": legs,
            "decode": decode,
            "step_tflops_algorithmic": train_flops_per_image(cfg) * value / 1e12,
        }
        out.update(roofline_block(cfg, args.dtype, agg, detail,
                                  traffic=vocab_traffic(args.config, args.dtype)))
        if args.detail:
            for k, d in sorted(detail.items(), key=lambda kv: -kv[1][3]):
                print(f"{k:42s} n={d[0]:3d} {d[3] * 1e3:8.3f} ms  {d[1] / d[3] / 1e12:7.2f} TF/s"
                      + (f"  {d[2] / d[3] / 1e9:8.1f} GB/s" if d[2] else ""), file=sys.stderr)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.config, cfg, extra=not args.no_legs, full=args.cpu_full)
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
    if world > 1 or forced:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
