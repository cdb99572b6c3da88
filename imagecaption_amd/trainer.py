"""Teacher-forced training step of the caption decoder, MI355X-first.

One ``CaptionTrainer.step`` is the body of the reference's loop (ref train.py:24-61): shift,
forward, label-smoothed loss over ALL B*T positions, backward, global-norm clip at 1.0, AdamW --
but as a fixed launch plan over flat HBM-resident blocks instead of autograd over module tensors:

* parameters, gradients and both Adam moments each live in ONE contiguous fp32 block (288 GB of
  HBM makes fragmentation pointless); the module's ``nn.Parameter``s are views into it, so
  ``state_dict()`` / checkpoints / the reference's own optimizer objects keep working;
* weight-gradient GEMMs write straight into the flat gradient block (no ``.grad`` accumulation
  pass, no ``zero_grad``); clip + AdamW is one streaming kernel reading the norm from device memory
  (the reference's ``loss.item()`` / clip host syncs are gone: nothing here synchronises);
* weight-gradient GEMMs are enqueued on a second stream (``ops.SideStream``, a lane verified by
  measurement to run beside the caller's stream): they depend on nothing later in backward and fill
  the matrix pipes while the critical path runs its latency-bound kernels;
* data parallel: the block is cut into buckets in backward order (head, layer L-1 .. 0,
  projection + embedding); each bucket's RCCL all-reduce(sum) is issued on a side stream as soon as
  its last gradient kernel is enqueued, overlapping the rest of backward.  The loss gradient is
  pre-scaled by 1/(rows * world) so the summed gradient IS the global-batch mean gradient and every
  rank clips by the same global norm (placement: before ref train.py:56).
"""
import torch
import torch.distributed as dist

import os

from . import _capi, engine, ops, train_plan
from .engine import DropoutPlan
from .parallel import BucketReducer, EqualRowsGuard, FlatParams, decoder_param_groups


class FusedAdamW(torch.optim.Optimizer):
    """The optimizer object the reference's driver code expects (ref main.py:95-104), backed by the trainer's
    fused clip + AdamW kernel over the flat blocks.  It IS a ``torch.optim.Optimizer``, so
    ``ReduceLROnPlateau(optimizer, ...)`` accepts it (ref main.py:98-104) and the warm-up loop's
    ``for g in optimizer.param_groups: g['lr'] = ...`` (ref train.py:436-447) and ``scheduler.step(bleu4)``
    (train.py:460) steer the kernel: ``CaptionTrainer.optimizer_step`` reads lr / betas / eps / weight_decay from
    ``param_groups[0]`` on every step.  One group over ``model.parameters()`` (what the reference builds);
    ``state_dict`` / ``load_state_dict`` speak ``torch.optim.AdamW``'s format (checkpoints, ref train.py:485-495)."""

    def __init__(self, trainer, params, lr, betas, eps, weight_decay):
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self._trainer = trainer

    def step(self, closure=None):
        loss = closure() if closure is not None else None
        self._trainer.optimizer_step()
        return loss

    def zero_grad(self, set_to_none=False):
        """No-op: every gradient is fully overwritten by the step's kernels, and the ``.grad`` views into the flat
        block must stay in place (the stock ``set_to_none=True`` would drop them)."""

    def state_dict(self):
        return self._trainer.adamw_state_dict()

    def load_state_dict(self, sd):
        self._trainer.load_adamw_state_dict(sd)


class CaptionTrainer:
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, max_norm=1.0,
                 smooth_factor=0.1, dropout=0.1, seed=0, process_group=None, overlap=True, dtype="fp32",
                 side_stream_wgrad=True, ce_backward="loader"):
        self.model = model
        self.optimizer = FusedAdamW(self, list(model.parameters()), lr, betas, eps, weight_decay)
        self.max_norm, self.smooth, self.p, self.seed = max_norm, smooth_factor, dropout, seed
        self.pg = process_group
        # how the two backward products of the vocabulary projection get dlogits: "loader" = they read the LOGITS
        # through the gradient transform in their operand loaders (nothing is written back); "pass" = one in-place
        # streaming pass turns the block into dlogits first.  tune() measures both and keeps the faster.
        self.ce_backward = ce_backward
        # "fp32" (exact fp32 MFMA, the reference's precision), "f32x3" (split-bf16 products, fp32-level accuracy)
        # "bf16" (bf16 MFMA products on fp32 operands rounded in the loaders, fp32 everything else) or "bf16s" (bf16
        # STORAGE: the products read bf16 copies of weights / activations / gradients, made once per tensor and step;
        # fp32 accumulation, results, norms, attention, loss, master weights and optimizer)
        self.dtype = dtype
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        self.rank = dist.get_rank(process_group) if self.world > 1 else 0
        self.step_count = 0
        dec = model.transformer_decoder
        self.nhead, self.nl = dec.nhead, len(dec.layers)
        # ---- flat blocks, ordered so that backward finishes buckets back to front -----------------
        groups = decoder_param_groups(model)
        seen = {id(p) for _, plist in groups for p in plist}
        extra = [n for n, p in model.named_parameters() if id(p) not in seen]
        if extra:
            raise RuntimeError(f"CaptionTrainer: parameters outside the decoder path: {extra}")
        self.flat = FlatParams(groups)
        self.flat_p, self.flat_g = self.flat.flat_p, self.flat.flat_g
        self.bucket_ranges, self.n_params = self.flat.bucket_ranges, self.flat.n_params
        self.flat_m = torch.zeros_like(self.flat_p)
        self.flat_v = torch.zeros_like(self.flat_p)
        dev = self.flat_p.device
        self.gnorm_sq = torch.zeros(1, dtype=torch.float32, device=dev)
        self.reducer = BucketReducer(self.flat_g, self.bucket_ranges, process_group, overlap)
        self._logits = self._ce_part = self._embed_ws = None
        # bf16 storage keeps the logits block in bf16 too (what a bf16-autocast F.linear hands the reference's loss): the
        # vocabulary product writes half the bytes, the gradient pass rewrites the block in place; the LOSS is still
        # the fp32 accumulators' (softmax partials and target logits leave through the product's epilogue)
        self.logits_bf16 = dtype == "bf16s" and os.environ.get("CAPDEC_LOGITS_BF16", "1") == "1"
        # bf16 storage: the weight operands' copies, refreshed by one launch per step (ops.Bf16Copies) unless switched
        # off (then each weight is cast by the first product of the step that reads it: same values, ~40 launches)
        self._w16, self.one_launch_weight_casts = None, True
        self.side = ops.SideStream(dev) if (side_stream_wgrad and dev.type == "cuda") else None
        # each decoder layer as ONE C call per direction (train_plan.LayerPlans) wherever those plans cover the step;
        # False: the op-by-op launch plan of engine.layer_fwd / layer_bwd (same launches, same bits, more host time)
        self.one_call_layers = os.environ.get("CAPDEC_ONE_CALL_LAYERS", "1") == "1"
        self._plans, self._plan_cache = None, {}
        # data-parallel ranks must all run the same B*T (the loss gradient is pre-scaled by 1/(rows*world)): checked by
        # a collective that every rank issues on every step (parallel.EqualRowsGuard)
        self._rows_guard = EqualRowsGuard(process_group, dev)
        if self.world > 1:
            # every rank must start from the same parameters (the ranks only ever exchange gradients): rank 0's win
            dist.broadcast(self.flat_p, src=dist.get_global_rank(process_group, 0) if process_group is not None else 0,
                           group=process_group)
        self._refresh_views()

    # lr / betas / eps / weight_decay live in the optimizer facade's param_groups[0]: that is where the reference's
    # warm-up loop and ReduceLROnPlateau write them (ref train.py:436-447, 460)
    def _group_get(self, key):
        return self.optimizer.param_groups[0][key]

    def _group_set(self, key, value):
        for g in self.optimizer.param_groups:
            g[key] = value

    lr = property(lambda self: self._group_get("lr"), lambda self, v: self._group_set("lr", v))
    betas = property(lambda self: self._group_get("betas"), lambda self, v: self._group_set("betas", tuple(v)))
    eps = property(lambda self: self._group_get("eps"), lambda self, v: self._group_set("eps", v))
    wd = property(lambda self: self._group_get("weight_decay"), lambda self, v: self._group_set("weight_decay", v))

    def _refresh_views(self):
        m = self.model
        self.P_layers = [layer.param_dict() for layer in m.transformer_decoder.layers]
        self.G_layers = [{k: v.grad for k, v in P.items()} for P in self.P_layers]
        for G in self.G_layers:  # these views are written by this trainer's launches only (engine._zero_query_grads)
            G["_cross_q_grads_zero"] = False
        self._kv_stack = None
        self._plans, self._plan_cache = None, {}
        fp = m.feature_projection
        self.P_proj = dict(zip(engine.PROJ_KEYS, [fp[0].weight, fp[0].bias, fp[1].weight, fp[1].bias, fp[4].weight,
                                                  fp[4].bias, fp[5].weight, fp[5].bias]))
        self.G_proj = {k: v.grad for k, v in self.P_proj.items()}
        ol = m.output_layer
        self.P_head = dict(zip(engine.HEAD_KEYS, [ol[0].weight, ol[0].bias, ol[3].weight, ol[3].bias]))
        self.G_head = {k: v.grad for k, v in self.P_head.items()}

    def _layer_plans(self, B, T, S, mem, drop_p):
        """The one-call layer plans for this step's geometry, or None where they do not apply (then engine.layer_fwd /
        layer_bwd run op by op): switched off, a recording / per-launch-timing pass, a geometry they do not cover, or
        bf16 storage before the persistent weight copies exist (the first step of a run)."""
        if not self.one_call_layers or ops.TUNE_RECORD is not None or ops.PROFILE is not None or mem.device.type != "cuda":
            return None
        dtype = ops.COMPUTE_DTYPE
        P0 = self.P_layers[0]
        dff, d = P0["linear1.weight"].shape
        if not train_plan.covers(B, T, S, self.nhead, d, dff, dtype, self.nl, mem.shape[0]):
            return None
        copies = None
        if dtype == _capi.BF16S:
            if self._w16 is None or not self.one_launch_weight_casts:
                return None
            copies = self._w16.memo()
        key = (B, T, S, self.nhead, True, float(drop_p), dtype, bool(ops.GROUP_WGRADS and ops.SIDE is not None),
               tuple(ops.GROUP_TILE))
        if self._plans is None or self._plans.key != key:
            # a few geometries stay planned side by side (training and evaluation steps alternate: dropout on / off, the
            # last batch of an epoch is shorter); the oldest goes when a fifth one turns up
            self._plans = self._plan_cache.get(key)
            if self._plans is None:
                if len(self._plan_cache) >= 4:
                    self._plan_cache.pop(next(iter(self._plan_cache)))
                try:
                    self._plans = train_plan.LayerPlans(self.P_layers, self.G_layers, B, T, S, self.nhead, True, drop_p, dtype,
                                                        mem.device, copies)
                except KeyError:  # a weight operand without a persistent bf16 copy: stay with the op-by-op plan
                    return None
                self._plan_cache[key] = self._plans
        return self._plans

    # ---- one iteration of ref train.py:24-61 ---------------------------------------------------------
    def forward_backward(self, features, captions, train=True, keep_logits=False):
        prev, ops.SIDE = ops.SIDE, self.side
        # bf16 storage: one bf16 copy per tensor and step (weights, activations, gradients), shared by the forward,
        # input-gradient and weight-gradient products that read it; dropped when the step's kernels are enqueued
        memo, ops.BF16_MEMO = ops.BF16_MEMO, ({} if self.dtype == "bf16s" else None)
        try:
            if self._w16 is not None and self.dtype == "bf16s" and self.one_launch_weight_casts:
                self._w16.refresh()
                ops.BF16_MEMO.update(self._w16.memo())
            with ops.compute_dtype(self.dtype):
                out = self._forward_backward(features, captions, train, keep_logits)
            if self._w16 is None and self.dtype == "bf16s" and self.one_launch_weight_casts:
                # the first step cast each weight operand on first use; every later step refreshes those same copies
                # with one launch up front.  Weight operands are recognised by their storage: views of the flat block
                base = self.flat_p.untyped_storage().data_ptr()
                w = [(t, c) for t, c in ops.BF16_MEMO.values()
                     if t.dtype == torch.float32 and t.untyped_storage().data_ptr() == base]
                if w:
                    self._w16 = ops.Bf16Copies([t for t, _ in w], [c for _, c in w])
            return out
        finally:
            if self.side is not None:
                self.side.join()
            ops.SIDE = prev
            ops.BF16_MEMO = memo

    def _forward_backward(self, features, captions, train, keep_logits):
        m = self.model
        tgt_in, tgt_out = ops.shift_captions(captions)   # ref train.py:29-30
        B, T = tgt_in.shape
        self._rows_guard.step(B * T)
        # each rank draws its own dropout masks (rank folded into the key): the same masks on different data would
        # correlate the noise across the global batch
        dp = DropoutPlan(self.p if train else 0.0, self.seed ^ engine._splitmix64(self.rank) if self.rank else self.seed,
                         self.step_count)
        f2 = features.reshape(-1, features.shape[-1])
        S = 1 if features.dim() == 2 else features.shape[1]
        sv_proj, sv_stack, sv_head = {}, {}, {}
        # the embedding gradient's row sort depends on the token ids only: it runs now, on the (idle) weight-gradient
        # lane beside the forward pass, so that only the summation is left on the critical path at the end of backward
        V_emb, d_emb = m.embedding.weight.shape
        if self._embed_ws is None or self._embed_ws[0] != (B, T):
            self._embed_ws = ((B, T), ops.embed_bwd_workspace(B, T, d_emb, V_emb, features.device))
        embed_ws = self._embed_ws[1]
        plan_done = None
        if self.side is not None:
            self.side.run(lambda: ops.embed_bwd_plan(tgt_in, d_emb, V_emb, embed_ws), tgt_in)
            plan_done = torch.cuda.Event()
            plan_done.record(self.side.stream)
        else:
            ops.embed_bwd_plan(tgt_in, d_emb, V_emb, embed_ws)
        mem = engine.proj_fwd(self.P_proj, f2, dp, sv_proj)
        x = ops.embed_fwd(tgt_in, m.embedding.weight, m.pos_encoder.table(), 0, dp.p, dp.seed(engine.S_EMBED))
        dec = m.transformer_decoder
        kv_stack = None
        if engine.kv_batched(self.nl, mem.shape[0]):
            # the layers' cross-attention K | V weights and biases as one [L*2d, d] operand: a persistent block refreshed
            # by ONE launch (the parameters live apart in the flat block, in state_dict order)
            if self._kv_stack is None:
                d = mem.shape[1]
                self._kv_stack = (ops.StackedRows([P["multihead_attn.in_proj_weight"][d:] for P in self.P_layers]),
                                  ops.StackedRows([P["multihead_attn.in_proj_bias"][d:] for P in self.P_layers]))
            kv_stack = (self._kv_stack[0].refresh(), self._kv_stack[1].refresh())
        y = engine.stack_fwd(self.P_layers, dec.norm.weight, dec.norm.bias, x.view(B * T, -1), mem, B, T, S, self.nhead,
                             True, dp, sv_stack, kv_stack=kv_stack, plans=self._layer_plans(B, T, S, mem, dp.p))
        # the logits block (B*T x V fp32, the step's largest tensor) is owned by the trainer: one allocation
        # for the run instead of a 100+ MB request to the caching allocator every step
        z16 = self.logits_bf16 and self.dtype == "bf16s"
        if self._logits is None or tuple(self._logits.shape) != (B * T, self.P_head["3.weight"].shape[0]) \
                or (self._logits.dtype == torch.bfloat16) != z16:
            if z16:
                self._logits = ops.bf16_matrix(B * T, self.P_head["3.weight"].shape[0], features.device)
            else:
                self._logits = torch.empty((B * T, self.P_head["3.weight"].shape[0]), dtype=torch.float32,
                                           device=features.device)
        # label-smoothed CE fused into the vocabulary product: its epilogue leaves per-tile softmax partials, a tiny
        # finalize gives lse / loss per row, and the two backward products read the LOGITS through the gradient
        # transform in their loaders -- the [B*T, V] block is written once and never rewritten (fp32; the other
        # compute types apply the same transform as one in-place pass)
        V, K2 = self.P_head["3.weight"].shape
        tiles = ops.ce_tiles(B * T, V, K2)
        if self._ce_part is None or tuple(self._ce_part.shape) != (B * T, tiles, 4):
            self._ce_part = torch.empty((B * T, tiles, 4), dtype=torch.float32, device=features.device)
        targets = tgt_out.view(-1)
        logits = engine.head_fwd(self.P_head, y, dp, sv_head, logits_out=self._logits, ce_partials=self._ce_part,
                                 ce_targets=targets if z16 else None)
        keep = (logits.float() if z16 else logits.clone()) if keep_logits else None
        lse, rows = ops.ce_finalize(self._ce_part, logits, targets, self.smooth)
        loss = ops.reduce_sum(rows, 1.0 / (B * T))
        nb = len(self.bucket_ranges)
        ce = (lse, targets, self.smooth, 1.0 / (B * T * self.world))
        if self.dtype == "bf16s":  # the gradient goes straight to the bf16 copy both backward products read
            logits, ce = ops.ce_dlogits_bf16(logits, *ce), None
        elif self.dtype != "fp32" or V % 4 or self.ce_backward == "pass":  # the loader transform: exact fp32, float4 loaders
            ops.ce_dlogits(logits, *ce)
            ce = None
        dy = engine.head_bwd(self.P_head, self.G_head, sv_head, logits, dp, ce=ce)

        lanes = tuple(self.side.streams) if self.side is not None else ()

        def after_layer(l):
            # a bucket's collective waits for the weight-gradient lanes too; the critical path does not
            if l == self.nl - 1:
                self.reducer.reduce(nb - 1, after=lanes)  # head + final norm are complete once layer L-1 started
            self.reducer.reduce(1 + l, after=lanes)

        dx, dmem = engine.stack_bwd(self.P_layers, self.G_layers, dec.norm.weight, dec.norm.bias, dec.norm.weight.grad,
                                    dec.norm.bias.grad, sv_stack, dy, mem, B, T, S, self.nhead, True, dp, after_layer)
        engine.proj_bwd(self.P_proj, self.G_proj, sv_proj, dmem, dp)
        if plan_done is not None:
            torch.cuda.current_stream().wait_event(plan_done)
        ops.embed_bwd(tgt_in, dx.view(B, T, -1), V_emb, dp.p, dp.seed(engine.S_EMBED), dE=m.embedding.weight.grad,
                      planned_ws=embed_ws)
        self.reducer.reduce(0, after=lanes)
        self.reducer.finish()
        return loss, keep

    def tune(self, features, captions, train=True, verbose=False):
        """One-time set-up for a run of equally shaped steps: record the matrix products of one forward + backward
        pass, time every candidate GEMM plan of each and pin the fastest in the library (``ops.tune_gemm_plans``).
        Changes no parameter and no optimizer state; call it before the steps that matter are timed."""
        ops.TUNE_RECORD = {}
        keep = self.ce_backward
        V = self.P_head["3.weight"].shape[0]
        variants = ("loader", "pass") if self.dtype == "fp32" and V % 4 == 0 else (keep,)  # others have one form
        try:
            for variant in variants:  # both forms of the vocabulary projection's backward get their plans
                self.ce_backward = variant
                self.forward_backward(features, captions, train=train)
        finally:
            sigs, ops.TUNE_RECORD = ops.TUNE_RECORD, None
            self.ce_backward = keep
        torch.cuda.synchronize()
        out = ops.tune_gemm_plans(sigs, verbose=verbose)
        # ... and the faster of the two forms is kept (whole steps, timed with events on this stream)
        times = {}
        if len(variants) == 1:
            self.ce_backward = keep
            return out
        for variant in variants:
            self.ce_backward = variant
            self.forward_backward(features, captions, train=train)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                self.forward_backward(features, captions, train=train)
            e1.record()
            e1.synchronize()
            times[variant] = e0.elapsed_time(e1) / 3
        # the loader form (the block is written once, never rewritten) unless the pass form is clearly faster here
        self.ce_backward = "pass" if times["pass"] < 0.98 * times["loader"] else "loader"
        if verbose:
            import sys
            print(f"[capdec tune] vocabulary-projection backward: {times} ms per forward+backward -> {self.ce_backward}",
                  file=sys.stderr, flush=True)
        out["ce_backward"] = (self.ce_backward, times[self.ce_backward], times[keep if keep in times else "loader"])
        return out

    def check_ids(self):
        """Raise IndexError if a step since the last call saw token ids outside the vocabulary (synchronises: call it
        once per epoch or per logging interval, not per step)."""
        ops.check_ids()
        self._rows_guard.verify()

    def optimizer_step(self):
        self.step_count += 1
        ops.sumsq(self.flat_g, out=self.gnorm_sq)
        ops.adamw(self.flat_p, self.flat_g, self.flat_m, self.flat_v, self.step_count, self.lr, self.betas, self.eps,
                  self.wd, gnorm_sq=self.gnorm_sq, max_norm=self.max_norm)

    def step(self, features, captions, train=True):
        """Returns the (device-resident, un-synchronised) mean loss of this rank's batch."""
        loss, _ = self.forward_backward(features, captions, train=train)
        self.optimizer_step()
        return loss

    # ---- interchange with torch.optim.AdamW (checkpoint.py; ref train.py:485-495, main.py:140-142) ----------
    def _as_torch_adamw(self):
        return torch.optim.AdamW(self.model.parameters(), lr=self.lr, betas=tuple(self.betas), eps=self.eps,
                                 weight_decay=self.wd)

    def adamw_state_dict(self):
        """The optimizer state in ``torch.optim.AdamW.state_dict()`` format over ``model.parameters()``."""
        opt = self._as_torch_adamw()
        if self.step_count > 0:
            off = {id(p): o for p, o in self.flat.placed}
            for p in self.model.parameters():
                o, n = off[id(p)], p.numel()
                opt.state[p] = {"step": torch.tensor(float(self.step_count)),
                                "exp_avg": self.flat_m[o:o + n].view_as(p).clone(),
                                "exp_avg_sq": self.flat_v[o:o + n].view_as(p).clone()}
        return opt.state_dict()

    def load_adamw_state_dict(self, sd):
        opt = self._as_torch_adamw()
        opt.load_state_dict(sd)
        g = opt.param_groups[0]
        self.lr, self.betas, self.eps, self.wd = g["lr"], tuple(g["betas"]), g["eps"], g["weight_decay"]
        for k, v in g.items():  # scheduler bookkeeping such as 'initial_lr' travels with the group
            if k not in ("params", "lr", "betas", "eps", "weight_decay"):
                self._group_set(k, v)
        off = {id(p): o for p, o in self.flat.placed}
        self.flat_m.zero_()
        self.flat_v.zero_()
        steps = set()
        for p in self.model.parameters():
            st = opt.state.get(p)
            if not st:
                continue
            o, n = off[id(p)], p.numel()
            self.flat_m[o:o + n].view_as(p).copy_(st["exp_avg"])
            self.flat_v[o:o + n].view_as(p).copy_(st["exp_avg_sq"])
            steps.add(int(st["step"]))
        if len(steps) > 1:
            raise RuntimeError(f"CaptionTrainer keeps one step counter; the state has {sorted(steps)}")
        self.step_count = steps.pop() if steps else 0

    def grad_norm(self):
        return self.gnorm_sq.sqrt()
