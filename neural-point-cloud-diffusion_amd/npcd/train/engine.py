"""Denoiser training step for MI355X: flat parameter / gradient / optimizer-state buffers, bucketed
gradient all-reduce over RCCL overlapped with backward, AdamW + EMA over the flat buffers.

Reproduces one iteration of the reference loop (npcd/train/diffusion_training.py:143-174):
zero_grad -> autocast(compute_loss) -> backward -> AdamW step (lr 7e-5, wd 0.01 on every parameter,
:116) -> EMA update (decay 0.9999 on parameters, buffers copied; utils/ema.py:114-138).  The reference
is single-GPU; data parallelism over diffusion samples (one process per GPU, gradients averaged with
an all-reduce) is what this build adds.

Memory layout: all trainable parameters of `model` live in ONE contiguous fp32 buffer (each
nn.Parameter is a view), likewise gradients, Adam moments and the EMA copy.  The optimizer and the
EMA are then single elementwise passes over 310 M contiguous floats and the gradient all-reduce
works on slices of one buffer -- no per-tensor launches, no flatten/unflatten copies.
"""
import os
from typing import List, Optional

import torch
import torch.distributed as dist
import torch.nn as nn


def _none():
    return None


class _ParamWaitHook:
    """forward-pre / state_dict-pre hook a DiffusionTrainer installs on its model: completes pending parameter gathers.  It holds the
    trainer STRONGLY (a dropped trainer reference can never turn the waits into no-ops), and it does not travel: copy.deepcopy(model)
    and pickling (torch.save(model)) give the copy an inert hook, never a copy of the trainer's flat buffers, streams, ctypes handles
    and process groups."""

    def __init__(self, trainer, kind):
        self.trainer, self.kind = trainer, kind

    def __call__(self, module, *args):
        t = self.trainer
        if t is None:
            return None
        if self.kind == "forward":
            t._await_params_for_forward(module, args[0])
        else:
            t.wait_params()
        return None

    def __deepcopy__(self, memo):
        return _ParamWaitHook(None, self.kind)

    def __reduce__(self):
        return (_ParamWaitHook, (None, self.kind))


class FlatBuffers:
    """Re-home the trainable parameters of `module` into one flat buffer (+ flat grads)."""
    ALIGN = 256

    def __init__(self, module: nn.Module):
        self.params: List[nn.Parameter] = [p for p in module.parameters() if p.requires_grad]
        assert self.params, "no trainable parameters"
        dev, dt = self.params[0].device, self.params[0].dtype
        assert all(p.dtype == dt and p.device == dev for p in self.params)
        # every view starts on a multiple of ALIGN elements: 16-byte aligned for the vector kernels, and every bucket of the
        # gradient reducer (a union of whole parameters) then splits into equal, 16-byte aligned shards for up to 64 ranks
        self.offsets, n = [], 0
        for p in self.params:
            self.offsets.append(n)
            n += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.numel = n
        self.flat = torch.zeros(n, dtype=dt, device=dev)
        self.grad = torch.zeros(n, dtype=dt, device=dev)
        with torch.no_grad():
            for p, off in zip(self.params, self.offsets):
                view = self.flat[off:off + p.numel()].view_as(p)
                view.copy_(p.data)
                p.data = view
                p.grad = self.grad[off:off + p.numel()].view_as(p)

    def zero_grad(self):
        self.grad.zero_()
        for p, off in zip(self.params, self.offsets):     # re-attach in case something replaced .grad
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + off * self.grad.element_size():
                p.grad = self.grad[off:off + p.numel()].view_as(p)


class GradReducer:
    """Bucketed all-reduce(mean) of the flat gradient buffer, overlapped with backward.

    Buckets are contiguous slices of the flat gradient in REVERSE parameter order (the order backward
    produces them); a bucket's collective is launched from the post-accumulate-grad hook of its last
    parameter to become ready.  `finish()` waits for all of them."""

    def __init__(self, flat: FlatBuffers, group=None, bucket_bytes: int = 64 << 20, always_reduce: bool = False,
                 comm_dtype: Optional[torch.dtype] = None):
        """always_reduce: run the collectives also on a one-rank group (lets a single-GPU box exercise the RCCL code path:
        ReduceOp.AVG, async work handles, stream ordering -- RCCL refuses two ranks on one device).
        comm_dtype=torch.bfloat16: the gradient buckets travel as bf16 (half the bytes of the reduce-scatter / all-reduce: 0.62
        instead of 1.24 GB per step at cfg-D); a bucket is rounded once before the collective, the averaged result is widened
        back into the fp32 gradient buffer / shard, the optimizer stays fp32.  Default (None): fp32 on the wire."""
        self.flat, self.group = flat, group
        self.comm_dtype = comm_dtype if comm_dtype not in (None, torch.float32) else None
        self.wire_bytes = 0             # bytes this rank handed to gradient collectives in the current step (bookkeeping for DESIGN 6)
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.active = self.world > 1 or (always_reduce and dist.is_available() and dist.is_initialized())
        self.rank = dist.get_rank(group) if self.active else 0
        # sharded mode (set by the trainer): reduce-scatter the gradient buckets, every rank updates 1/world of each bucket,
        # all-gather the parameters -- the optimizer pass (38 bytes per parameter) shrinks by the number of ranks
        self.shard = False
        self.gshard = None
        self.handles = []
        self.buckets = []           # (start, end) element ranges
        self.param_bucket = {}      # id(param) -> bucket index
        self.pending: List[int] = []
        per = max(1, bucket_bytes // flat.grad.element_size())
        end = flat.numel
        members, start_of = [], end
        for p, off in reversed(list(zip(flat.params, flat.offsets))):
            members.append(p)
            start_of = off
            if end - start_of >= per:
                self._close(members, start_of, end)
                members, end = [], start_of
        if members:
            self._close(members, start_of, end)
        self._avg = None
        if self.active:
            for p in flat.params:
                p.register_post_accumulate_grad_hook(self._hook)

    def _close(self, members, start, end):
        b = len(self.buckets)
        self.buckets.append((start, end))
        for p in members:
            self.param_bucket[id(p)] = b
        self.pending.append(len(members))

    def enable_sharding(self) -> bool:
        """Needs equal 16-byte aligned shards: (bucket length) % (4 * world) == 0 for every bucket."""
        if not self.active or any((e - s) % (4 * self.world) for s, e in self.buckets):
            return False
        self.shard = True
        self.gshard = torch.empty(self.flat.numel // self.world, dtype=self.flat.grad.dtype, device=self.flat.grad.device)
        return True

    def shard_range(self, s, e):
        n = (e - s) // self.world
        return s + self.rank * n, s + (self.rank + 1) * n

    def start_step(self):
        self.handles = []
        self.launched = []          # bucket index of every handle, in launch order
        self._left = list(self.pending)
        self.wire_bytes = 0

    def _launch(self, b):
        """handles: (work, post) -- post() runs right after the wait: division for back ends without a native average, widening of
        a bf16 bucket back into the fp32 buffer"""
        self.launched.append(b)
        s, e = self.buckets[b]
        buf = self.flat.grad[s:e]
        if self._avg is None:       # RCCL has a native average; gloo (CPU tests) does not
            self._avg = dist.get_backend(self.group) == "nccl"
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        scale = None if self._avg else 1.0 / self.world
        wire = buf if self.comm_dtype is None else buf.to(self.comm_dtype)          # (one rounding of the local bucket)
        self.wire_bytes += wire.numel() * wire.element_size()
        if self.shard:
            out = self.gshard[s // self.world:e // self.world]
            if self.comm_dtype is None:
                h = dist.reduce_scatter_tensor(out, wire, op=op, group=self.group, async_op=True)
                post = (lambda o=out: o.mul_(scale)) if scale is not None else None
            else:
                out16 = torch.empty(out.shape, dtype=self.comm_dtype, device=out.device)
                h = dist.reduce_scatter_tensor(out16, wire, op=op, group=self.group, async_op=True)

                def post(o=out, o16=out16, w=wire):          # (w: keeps the send buffer alive until the collective is done)
                    # (a bucket may be launched from the backward's side stream: tell the allocator this stream reads the buffers too)
                    if o16.is_cuda:
                        o16.record_stream(torch.cuda.current_stream())
                    o.copy_(o16)
                    if scale is not None:
                        o.mul_(scale)
            self.handles.append((h, post))
            return
        h = dist.all_reduce(wire, op=op, group=self.group, async_op=True)
        if self.comm_dtype is None:
            post = (lambda t=buf: t.mul_(scale)) if scale is not None else None
        else:
            def post(t=buf, w=wire):
                if w.is_cuda:
                    w.record_stream(torch.cuda.current_stream())
                t.copy_(w)
                if scale is not None:
                    t.mul_(scale)
        self.handles.append((h, post))

    def mark_ready(self, p):
        """For gradients written outside autograd (the fused backbone backward)."""
        if self.active:
            self._hook(p)

    def _hook(self, p):
        b = self.param_bucket[id(p)]
        self._left[b] -= 1
        if self._left[b] == 0:
            self._launch(b)

    def finish(self, on_bucket_ready=None):
        """Wait for every bucket's collective.  `on_bucket_ready(start, end)` is called right after the wait of
        each bucket (in launch order): work it enqueues on the current stream only depends on THAT bucket's
        all-reduce, so it overlaps with the collectives still in flight (used to pipeline the optimizer)."""
        if not self.active:
            return False
        for b, left in enumerate(self._left):     # parameters that received no gradient this step
            if left > 0:
                self._launch(b)
                self._left[b] = 0
        self.pending_at_finish = 0      # collectives still running when the backward ended (bookkeeping: the exposed ones)
        for (h, post), b in zip(self.handles, self.launched):
            try:
                self.pending_at_finish += 0 if h.is_completed() else 1
            except (RuntimeError, AttributeError):
                pass
            h.wait()
            if post is not None:
                post()
            if on_bucket_ready is not None:
                on_bucket_ready(*self.buckets[b])
        self.handles = []
        return on_bucket_ready is not None


class DiffusionTrainer:
    """One-process-per-GPU trainer of `model.diffusion` (a DiffusionModel)."""

    def __init__(self, diffusion: nn.Module, lr: float = 7e-5, weight_decay: float = 0.01, ema_decay: Optional[float] = 0.9999,
                 dtype: Optional[torch.dtype] = torch.bfloat16, group=None, bucket_bytes: int = 64 << 20, max_grad_norm=None,
                 fused: bool = True, always_reduce: bool = False, shard_optimizer: Optional[bool] = None,
                 comm_dtype: Optional[torch.dtype] = None):
        """shard_optimizer (default: on whenever gradients are exchanged on the native path without clipping / loss scaling):
        ZeRO-1 style -- reduce-scatter instead of all-reduce, each rank runs AdamW + EMA on 1/world of every bucket, the updated
        parameters are all-gathered.  Same bytes on the wire as the all-reduce, optimizer pass divided by the number of ranks."""
        self.model = diffusion
        self.dtype = dtype
        # the training loop never looks at the pointwise losses: this package's DiffusionModel can skip materialising them
        # (a model with the reference's plain compute_loss signature is called as is)
        import inspect
        self._loss_kwargs = ({"want_pointwise": False}
                             if "want_pointwise" in inspect.signature(diffusion.compute_loss).parameters else {})
        self.flat = FlatBuffers(diffusion)
        if comm_dtype is None and os.environ.get("NPCD_COMM_BF16"):
            comm_dtype = torch.bfloat16
        self.reducer = GradReducer(self.flat, group, bucket_bytes, always_reduce, comm_dtype)
        self.lr, self.weight_decay, self.betas, self.eps = lr, weight_decay, (0.9, 0.999), 1e-8
        self.ema_decay = ema_decay
        self.ema = self.flat.flat.clone() if ema_decay is not None else None
        self.max_grad_norm = max_grad_norm
        self.iteration = 0                 # optimizer steps applied (AdamW bias correction)
        self.finished_iterations = 0       # loop iterations, including those a float16 overflow skipped (diffusion_training.py:190)
        # float16 autocast (the reference's default --dtype, train_diffusion.py:78) trains with dynamic loss scaling
        # (torch.cuda.amp.GradScaler defaults, diffusion_training.py:62,156,169-170): scale 2^16, halved when a gradient
        # overflows (that step is skipped), doubled after 2000 clean steps
        self.loss_scale = 65536.0 if dtype == torch.float16 else None
        self._clean_steps = 0
        self.skipped_steps = 0
        self.native = self.flat.flat.is_cuda and fused
        if self.native:
            # HIP path: one fused AdamW+EMA kernel over the flat buffers, a bf16 shadow of the parameters for
            # the GEMMs and the explicit backbone forward/backward (npcd.models.diffusion.fused)
            from ..hip import elementwise as ew
            from ..models.diffusion.fused import FusedBackboneEngine
            self._ew = ew
            self.exp_avg = torch.zeros_like(self.flat.flat)
            self.exp_avg_sq = torch.zeros_like(self.flat.flat)
            # the 16-bit shadow the GEMMs read is kept in the run's autocast type: bf16, or f16 (the reference's default --dtype,
            # train_diffusion.py:78 -- trained with the loss scaling below); any other dtype trains through the module path
            half = dtype if dtype in (torch.bfloat16, torch.float16) else torch.bfloat16
            self.shadow = torch.empty(self.flat.numel, dtype=half, device=self.flat.flat.device)
            ew.cast_f32_bf16(self.flat.flat, self.shadow)
            denoiser = getattr(diffusion, "denoiser", None)
            fused_ids = set()
            if (denoiser is not None and dtype in (torch.bfloat16, torch.float16)
                    and denoiser.backbone.width // denoiser.backbone.resblocks[0].attn.heads == 64):
                denoiser.backbone.fused_engine = FusedBackboneEngine(denoiser.backbone, self.flat, self.shadow, self.reducer)
                fused_ids = {id(p) for e in denoiser.backbone.fused_engine.blocks for p in e["params"]}
            # parameters whose gradients ACCUMULATE through autograd (everything outside the fused backbone, which overwrites):
            # their gradient ranges must be zeroed every step when the fused optimizer kernel only sees a shard of them
            ranges, align = [], FlatBuffers.ALIGN
            for p, off in zip(self.flat.params, self.flat.offsets):
                if id(p) in fused_ids:
                    continue
                end = off + (p.numel() + align - 1) // align * align        # (alignment padding never holds a gradient)
                if ranges and ranges[-1][0] + ranges[-1][1] == off:
                    ranges[-1] = (ranges[-1][0], ranges[-1][1] + end - off)  # merge neighbours: a handful of fills per step
                else:
                    ranges.append((off, end - off))
            self._accum_ranges = ranges
            want = shard_optimizer if shard_optimizer is not None else not os.environ.get("NPCD_NO_SHARD_OPTIMIZER")
            if want and self.reducer.active and max_grad_norm is None and self.loss_scale is None:
                self.reducer.enable_sharding()
            # Sharded optimizer: the updated parameters come back with one all-gather per bucket.  Those gathers are NOT awaited
            # at the end of the step: a bucket is awaited (and its bf16 shadow refreshed) right before the next forward first
            # reads a parameter of it -- the fused backbone asks block by block -- so that the 1.24 GB of parameter traffic runs
            # under the next step's forward instead of after the optimizer.  Buckets are filled in reverse parameter order, so
            # only the LAST bucket (block 0, ln_pre, time_embed) is needed immediately and stays exposed.
            self._pending = {}                   # bucket index -> (work handle, send buffer, start, end)
            self._bucket_of = {se: i for i, se in enumerate(self.reducer.buckets)}
            self.lazy_gather = not os.environ.get("NPCD_EAGER_PARAM_GATHER")
            eng = getattr(getattr(denoiser, "backbone", None), "fused_engine", None) if denoiser is not None else None
            if eng is not None:
                offs = {id(p): (o, o + p.numel()) for p, o in zip(self.flat.params, self.flat.offsets)}
                eng.block_ranges = [(min(offs[id(p)][0] for p in e["params"]), max(offs[id(p)][1] for p in e["params"])) for e in eng.blocks]
                eng.wait_range = self.wait_params
            # The waits live in the MODEL, not in step(): whoever runs a forward (step(), a custom compute_loss + apply_gradients
            # loop, generate() in any precision) or reads the weights through state_dict() first completes the gathers it depends
            # on.  In-place writes into parameters (copying EMA weights in) cannot be intercepted: call wait_params() first.
            self._fused_engine = eng
            # The hooks hold the trainer STRONGLY (the model keeps its trainer alive): as long as a lazily gathered parameter can
            # still be in flight, every forward and every state_dict() of the model completes it first -- a dropped trainer
            # reference can never turn the waits into no-ops and let a checkpoint read half-gathered weights.  A trainer leaves the
            # model only through close(); building another DiffusionTrainer on the same model close()s the attached one first
            # (no stacked hooks, its moments / EMA / shadow are released with it).
            prev = diffusion.__dict__.get("_npcd_trainer")
            if prev is not None and prev is not self:
                prev.close()
            self._hook_handles = []
            # (model -> hook -> trainer -> model is a reference cycle: the trainer's buffers -- moments, EMA, shadow: 16 bytes per parameter --
            # are released by close(), or by the cyclic collector some time after the last reference; call close() when done with a trainer)
            if denoiser is not None:
                self._hook_handles.append(denoiser.register_forward_pre_hook(_ParamWaitHook(self, "forward")))
                self._hook_handles.append(denoiser.register_state_dict_pre_hook(_ParamWaitHook(self, "state_dict")))   # denoiser.state_dict() read directly
            self._hook_handles.append(diffusion.register_state_dict_pre_hook(_ParamWaitHook(self, "state_dict")))
            diffusion.__dict__["_npcd_trainer"] = self
        else:
            # reference path (CPU tests / ablation): torch AdamW over ONE flat "parameter"
            self.master = nn.Parameter(self.flat.flat, requires_grad=True)
            self.master.grad = self.flat.grad
            self.optimizer = torch.optim.AdamW([self.master], lr=lr, weight_decay=weight_decay, fused=self.flat.flat.is_cuda)

    # A trainer never travels with its model: a deep copy or a pickle of a model with an attached trainer (EMA snapshots by
    # copy.deepcopy(model), torch.save(model)) carries None in its place and the module path instead of the fused engine.
    def __deepcopy__(self, memo):
        return None

    def __reduce__(self):
        return (_none, ())

    def comm_stats(self):
        """Communication bookkeeping of the LAST step of this rank (bench.py puts it into the line for n_gpus > 1, DESIGN.md
        section 6): bytes handed to the gradient collectives, bytes of the parameter all-gather, number of collectives, and how many
        gradient collectives were still running when the backward had finished (the exposed ones)."""
        red = self.reducer
        grad_elem = 2 if red.comm_dtype is not None else 4
        shard = bool(red.shard)
        gather = sum((e - s) // red.world * 4 for s, e in red.buckets) if shard else 0
        return {"mode": ("reduce_scatter+sharded_adamw+all_gather" if shard else "all_reduce") if red.active else "none",
                "world": red.world, "buckets": len(red.buckets), "bucket_bytes_fp32": [(e - s) * 4 for s, e in red.buckets],
                "gradient_wire_dtype": "bf16" if grad_elem == 2 else "fp32",
                "gradient_bytes_handed_to_collectives": int(red.wire_bytes),
                "parameter_all_gather_send_bytes": int(gather),
                "collectives_per_step": len(getattr(red, "launched", [])) * (2 if shard else 1),
                "gradient_collectives_pending_when_backward_ended": int(getattr(red, "pending_at_finish", 0)),
                "lazy_parameter_gather": bool(shard and getattr(self, "lazy_gather", False))}

    def close(self):
        """Detach this trainer from the model: complete pending parameter gathers, remove its forward / state_dict hooks and the
        fused engine it installed.  Call before building another DiffusionTrainer on the same model."""
        self.wait_params()
        for h in getattr(self, "_hook_handles", []):
            h.remove()
        self._hook_handles = []
        denoiser = getattr(self.model, "denoiser", None)
        eng = getattr(self, "_fused_engine", None)
        if denoiser is not None and eng is not None and getattr(denoiser.backbone, "fused_engine", None) is eng:
            denoiser.backbone.fused_engine = None
        if eng is not None:
            eng.wait_range = None                 # (a caller that still holds the engine no longer reaches this trainer)
        self._fused_engine = None
        if self.model.__dict__.get("_npcd_trainer") is self:
            del self.model.__dict__["_npcd_trainer"]

    def ema_state_dict(self, gathered: bool = False):
        """state_dict of the EMA model: same keys as the running model (utils/ema.py:80), buffers copied.  Collective with a
        sharded optimizer (all ranks call it) unless the caller has just gathered the EMA (`gathered=True`)."""
        self.wait_params()
        if not gathered:
            self.gather_ema()
        sd = {k: v.clone() for k, v in self.model.state_dict().items()}
        names = {id(p): n for n, p in self.model.named_parameters()}
        for p, off in zip(self.flat.params, self.flat.offsets):
            sd[names[id(p)]] = self.ema[off:off + p.numel()].view_as(p).clone()
        return sd

    def wait_params(self, lo: int = 0, hi: Optional[int] = None):
        """Complete the parameter all-gathers of the previous step for the flat range [lo, hi) (default: everything): wait for
        the collective of every still-pending bucket that overlaps the range and refresh its bf16 shadow.  Called lazily by
        the forward; call it without arguments before reading `flat.flat` / `shadow` directly."""
        pend = getattr(self, "_pending", None)
        if not pend:
            return
        hi = self.flat.numel if hi is None else hi
        for b in [b for b, (_, _, s0, e0) in pend.items() if s0 < hi and e0 > lo]:
            h, _, s0, e0 = pend.pop(b)
            h.wait()
            self._ew.cast_f32_bf16(self.flat.flat[s0:e0], self.shadow[s0:e0])
            self._shadow_written()

    def _await_params_for_forward(self, module, args):
        """forward-pre-hook of the denoiser.  The fused training forward asks for its blocks one by one (so that the gathers of
        the later blocks stay under the earlier blocks' compute): only the parameters read through ordinary modules (time_embed,
        ln_pre, ln_post, input / output projection) must be current before it starts.  Every other forward (module path in fp32 /
        f16, sampling) gets everything."""
        if not getattr(self, "_pending", None):
            return
        fused_training = (self._fused_engine is not None and torch.is_grad_enabled() and torch.is_autocast_enabled()
                          and torch.get_autocast_dtype("cuda") == self._fused_engine.dtype and args and args[0].is_cuda)
        if not fused_training:
            self.wait_params()
            return
        for off, n in self._accum_ranges:
            self.wait_params(off, off + n)

    def step(self, coords, feats, t=None, coords_noise=None, feats_noise=None):
        if not self.native or self.iteration == 0:
            self.flat.zero_grad()                 # afterwards the fused optimizer kernel leaves the gradients zeroed
        self.reducer.start_step()
        self.finished_iterations += 1
        dev_type = "cuda" if coords.is_cuda else "cpu"
        with torch.autocast(dev_type, dtype=self.dtype, enabled=self.dtype is not None):
            loss, sub, _ = self.model.compute_loss(coords, feats, t=t, coords_noise=coords_noise, feats_noise=feats_noise, **self._loss_kwargs)
        (loss if self.loss_scale is None else loss * self.loss_scale).backward()
        self.apply_gradients()
        return loss.detach(), sub

    def apply_gradients(self):
        """Second half of an iteration: finish the gradient exchange, then AdamW + EMA on what the flat gradient buffer holds
        (diffusion_training.py:169-174).  step() calls this after backward."""
        self.iteration += 1
        if self.native and self.reducer.shard:
            if self.reducer.finish(self._adamw_shard):
                self._finish_shards()
                return
        elif self.native and self.max_grad_norm is None and self.loss_scale is None:
            # multi-GPU: update each bucket's slice as soon as ITS all-reduce is done, under the remaining collectives
            if self.reducer.finish(self._adamw_range):
                return
        else:
            self.reducer.finish()
        if self.loss_scale is not None and not self._unscale_or_skip():
            return
        if self.max_grad_norm is not None:
            if self.native:
                self._clip_native()
            else:
                torch.nn.utils.clip_grad_norm_([self.master], self.max_grad_norm)
        if self.native:
            self._adamw_range(0, self.flat.numel, zero_grad=False)
            self._zero_accumulating()
        else:
            self.master.grad = self.flat.grad
            self.optimizer.step()
            if self.ema is not None:
                self.ema.lerp_(self.flat.flat, 1.0 - self.ema_decay)

    def _unscale_or_skip(self) -> bool:
        """GradScaler.step/update: returns False (step skipped, scale halved) when the reduced gradient holds an inf / nan
        -- every rank sees the same all-reduced values, hence takes the same decision."""
        g = self.flat.grad
        if not bool(torch.isfinite(g).all()):
            g.zero_()
            self.loss_scale *= 0.5
            self._clean_steps = 0
            self.skipped_steps += 1
            self.iteration -= 1                       # the optimizer's bias correction counts applied steps only
            if self.ema is not None:                  # the reference updates the EMA after every iteration (:172-174)
                self.ema.lerp_(self.flat.flat, 1.0 - self.ema_decay)
            return False
        g.mul_(1.0 / self.loss_scale)
        self._clean_steps += 1
        if self._clean_steps == 2000:
            self.loss_scale *= 2.0
            self._clean_steps = 0
        return True

    def _shadow_written(self):
        eng = getattr(self, "_fused_engine", None)
        if eng is not None:
            eng.shadow_written()

    def _adamw_range(self, s0, e0, zero_grad=True):
        self._shadow_written()
        ema = None if self.ema is None else self.ema[s0:e0]
        self._ew.adamw_ema(self.flat.flat[s0:e0], self.flat.grad[s0:e0], self.exp_avg[s0:e0], self.exp_avg_sq[s0:e0], ema,
                           self.shadow[s0:e0], self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay, self.iteration,
                           self.ema_decay, zero_grad=zero_grad)

    def _zero_accumulating(self):
        """Gradients that ACCUMULATE through autograd must start the next step at zero; the fused backbone overwrites its
        97 % of the buffer, so the optimizer pass does not write zeros over those (4 of its 42 bytes per parameter)."""
        for off, n in self._accum_ranges:
            self.flat.grad[off:off + n].zero_()

    # ---- sharded optimizer (ZeRO-1 style) ---------------------------------------------------------------------------
    def _adamw_shard(self, s0, e0):
        """called when bucket [s0, e0)'s reduce-scatter is done: update this rank's shard, start the parameter all-gather"""
        red = self.reducer
        self.wait_params(s0, e0)                      # (a no-op unless a step ran without any forward in between)
        a, b = red.shard_range(s0, e0)
        g = red.gshard[s0 // red.world:e0 // red.world]
        self._shadow_written()
        ema = None if self.ema is None else self.ema[a:b]
        self._ew.adamw_ema(self.flat.flat[a:b], g, self.exp_avg[a:b], self.exp_avg_sq[a:b], ema, self.shadow[a:b], self.lr, self.betas[0],
                           self.betas[1], self.eps, self.weight_decay, self.iteration, self.ema_decay, zero_grad=False)
        mine = self.flat.flat[a:b].clone()            # out-of-place gather: the output range contains the input range
        h = dist.all_gather_into_tensor(self.flat.flat[s0:e0], mine, group=red.group, async_op=True)
        self._pending[self._bucket_of[(s0, e0)]] = (h, mine, s0, e0)

    def _finish_shards(self):
        if not self.lazy_gather:
            self.wait_params()                        # parameters of the other ranks' shards have arrived: refresh the bf16 shadow
        self._zero_accumulating()

    def _gather(self, bufs):
        red = self.reducer
        if not red.shard or red.world == 1:
            return
        for buf in bufs:
            for s0, e0 in red.buckets:
                a, b = red.shard_range(s0, e0)
                dist.all_gather_into_tensor(buf[s0:e0], buf[a:b].clone(), group=red.group)

    def gather_ema(self):
        """EMA shards -> the full EMA vector on every rank (before exporting / evaluating the EMA model)."""
        if self.ema is not None:
            self._gather([self.ema])

    def gather_state(self):
        """EMA and Adam-moment shards -> full vectors on every rank (before writing a checkpoint; collective)."""
        self.wait_params()
        if self.native:
            self._gather([self.exp_avg, self.exp_avg_sq] + ([self.ema] if self.ema is not None else []))

    # ---- train-state checkpoints in the reference's layout (npcd.train.checkpoint) -----------------------------------
    def state_dict(self, full_model=None):
        from .checkpoint import trainer_state_dict
        return trainer_state_dict(self, full_model)

    def load_state_dict(self, ckpt):
        self.wait_params()
        from .checkpoint import load_trainer_state
        load_trainer_state(self, ckpt)

    def _clip_native(self):
        norm = torch.linalg.vector_norm(self.flat.grad)
        self.flat.grad.mul_(torch.clamp(self.max_grad_norm / (norm + 1e-6), max=1.0))
