"""Data-parallel plumbing: flat parameter / gradient blocks, backward-ordered buckets, and the
bucketed gradient all-reduce (RCCL over xGMI on the GPU box; gloo in the CPU tests).

Device-agnostic on purpose: nothing here computes, it only lays tensors out and issues
``torch.distributed`` collectives, so the N>1 logic is testable with world_size-2 gloo on CPU.

The reference has no distributed code at all; the one exchange step of the
path is the gradient all-reduce(sum), placed after backward and BEFORE the global-norm clip
(ref train.py:56) so every rank clips by the same global norm.
"""
import os

import torch
import torch.distributed as dist

ALIGN = 64  # elements (256 B): every parameter view stays 16-byte aligned for the float4 kernels


class FlatParams:
    """Lay the parameters of ``groups`` = [(bucket_name, [params...]), ...] out in one contiguous block.

    After construction every ``p.data`` is a view into ``flat_p`` and every ``p.grad`` a view into
    ``flat_g`` (padding between tensors stays zero forever), so ``state_dict()``, checkpoints and
    stock optimizers keep working on the same storage.  ``bucket_ranges[i] = (name, start, end)``.
    """

    def __init__(self, groups, device=None):
        total, self.bucket_ranges, self.placed = 0, [], []
        for name, plist in groups:
            start = total
            for p in plist:
                self.placed.append((p, total))
                total += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
            self.bucket_ranges.append((name, start, total))
        dev = device if device is not None else self.placed[0][0].device
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        for p, off in self.placed:
            view = self.flat_p[off:off + p.numel()].view_as(p)
            view.copy_(p.data)
            p.data = view
            p.grad = self.flat_g[off:off + p.numel()].view_as(p)
        self.n_params = sum(p.numel() for p, _ in self.placed)
        self.total = total


class BucketReducer:
    """all-reduce(sum) of gradient buckets, each issued as soon as the caller says it is complete.

    On CUDA/HIP tensors the collective runs on a side stream ordered after the work enqueued so far
    on the current stream, so it overlaps the rest of backward; ``finish()`` joins it back.
    """

    def __init__(self, flat_g, bucket_ranges, process_group=None, overlap=True):
        self.flat_g, self.ranges, self.pg = flat_g, bucket_ranges, process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        # CAPDEC_FORCE_COLLECTIVES=1: issue the bucket collectives even in a one-rank group (a sum over one
        # rank), so the RCCL + side-stream choreography can be exercised on a single GPU
        self.active = self.world > 1 or (os.environ.get("CAPDEC_FORCE_COLLECTIVES") == "1" and dist.is_available()
                                         and dist.is_initialized())
        # gloo on device tensors is a rehearsal vehicle only (tests, one-GPU dry runs of bench.py): its staged
        # device collectives deadlock on this stack while the GPU is busy (reproduced with plain torch,
        # tools/gloo_probe2.py), so that combination stages the buckets through host memory in finish().
        self.serial = self.active and flat_g.is_cuda and dist.get_backend(process_group) == "gloo"
        # the collectives' stream is picked by measurement like the weight-gradient lane (ops.lane_stream): it must
        # not share a hardware queue with the compute stream, or RCCL's kernels run in line with backward
        self.side = None
        if overlap and self.active and flat_g.is_cuda and not self.serial:
            from . import ops
            self.side = ops.lane_stream(flat_g.device, slot=1)
        self.works, self.deferred = [], []

    def reduce(self, idx, after=()):
        """``after``: extra streams whose enqueued work also writes this bucket (the trainer's
        weight-gradient lanes); the collective is ordered behind them without stalling the caller."""
        if not self.active:
            return
        if self.serial:
            self.deferred.append((idx, tuple(after)))
            return
        _, a, b = self.ranges[idx]
        buf = self.flat_g[a:b]
        if self.side is not None:
            ev = torch.cuda.Event()
            ev.record()
            self.side.wait_event(ev)
            for st in after:
                self.side.wait_stream(st)
            with torch.cuda.stream(self.side):
                # async_op=False: current ProcessGroupNCCL launches a synchronous collective on the CURRENT stream
                # (here: the measured side stream) instead of its own pool stream; the host does not block
                dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg, async_op=False)
        else:
            for st in after:
                torch.cuda.current_stream().wait_stream(st)
            self.works.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def finish(self):
        for idx, after in self.deferred:
            for st in after:
                torch.cuda.current_stream().wait_stream(st)
            _, a, b = self.ranges[idx]
            host = self.flat_g[a:b].cpu()  # staged by hand: gloo's own device path is what deadlocks
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.pg)
            self.flat_g[a:b].copy_(host)
        self.deferred = []
        for w in self.works:
            w.wait()
        self.works = []
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)


class EqualRowsGuard:
    """Data-parallel steps pre-scale the loss gradient by 1/(rows * world), which is exact only if every rank runs the
    same B*T.  ``step(rows)`` is called by EVERY rank on EVERY step and always issues the same tiny collective
    (MAX over [rows, -rows]), so the ranks' collective sequences can never diverge -- a check that only some ranks
    enter (say, only those whose shape changed) would pair its all-reduce with the other ranks' next gradient bucket.

    The verdict is read without stalling the pipeline: a rank whose own shape changed (and every rank on the first
    step) reads it at once; otherwise the two words travel to pinned host memory behind the collective and are
    looked at when the next step starts (``verify``), one step late but before any further work is enqueued.
    gloo groups (CPU tests, one-GPU rehearsals) exchange host tensors and check immediately."""

    def __init__(self, process_group=None, device=None):
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        dev = torch.device("cpu") if device is None else torch.device(device)
        if self.world > 1 and dist.get_backend(process_group) == "gloo":
            dev = torch.device("cpu")
        self.device, self.rows, self.pending = dev, None, None

    @staticmethod
    def _judge(lo_hi):
        hi, lo = int(lo_hi[0]), -int(lo_hi[1])
        if hi != lo:
            raise RuntimeError(f"CaptionTrainer: ranks run different batch shapes (B*T between {lo} and {hi}); "
                               f"data-parallel steps need equal per-rank batches")

    def verify(self):
        if self.pending is not None:
            (host, ev), self.pending = self.pending, None
            ev.synchronize()
            self._judge(host)

    def step(self, rows):
        if self.world == 1:
            return
        self.verify()
        t = torch.tensor([rows, -rows], dtype=torch.int64).to(self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.pg)
        changed, self.rows = self.rows != rows, rows
        if self.device.type == "cpu" or changed:
            self._judge(t)
            return
        host = torch.empty(2, dtype=torch.int64, pin_memory=True)
        host.copy_(t, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.pending = (host, ev)


def shard_indices(n, rank, world, epoch=0, seed=42, shuffle=True):
    """Seeded per-rank partition of range(n) replacing DataLoader(shuffle=True) (ref main.py:63-70).

    Every rank draws the same permutation and takes a strided slice; the tail is dropped so all
    ranks run the same number of equal-sized batches (mean-of-means stays exact).
    """
    g = torch.Generator().manual_seed(seed + epoch)
    perm = torch.randperm(n, generator=g) if shuffle else torch.arange(n)
    per = n // world
    return perm[: per * world].view(per, world)[:, rank].contiguous()


def shard_range(n, rank, world):
    """Contiguous image shard for inference (no collective): [start, end) of rank."""
    base, rem = divmod(n, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def decoder_param_groups(model):
    """Buckets in the order backward completes them last-to-first: embed+proj | layer0..L-1 | norm+head."""
    dec = model.transformer_decoder
    groups = [("embed+proj", [model.embedding.weight] + list(model.feature_projection.parameters()))]
    for i, layer in enumerate(dec.layers):
        groups.append((f"layer{i}", list(layer.parameters())))
    groups.append(("norm+head", list(dec.norm.parameters()) + list(model.output_layer.parameters())))
    return groups
