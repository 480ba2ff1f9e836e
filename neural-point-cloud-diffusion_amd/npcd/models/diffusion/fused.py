"""Explicit forward/backward of the transformer backbone for the MI355X training step.

One autograd node for all L residual blocks (reference ResidualAttentionBlock, transformer.py:140-172)
instead of ~40 autograd nodes per block:
  * activations are saved explicitly, the backward is written out by hand;
  * GEMMs are hipBLASLt calls on a bf16 shadow copy of the flat fp32 parameter buffer (no per-step
    weight casts); weight gradients are produced in fp32 straight into the flat gradient buffer;
  * residual add + LayerNorm + cast, LayerNorm backward (+ bias-gradient column sums), GELU fwd/bwd are
    single HIP kernels (csrc/elementwise.hip); attention is csrc/attention.hip;
  * gradients never pass through AccumulateGrad; the gradient reducer is told explicitly when a
    block's gradients are final so the RCCL all-reduce overlaps with the rest of backward.
Numerics follow the reference's bf16-autocast step: bf16 GEMM inputs with fp32 accumulation, fp32
residual stream, fp32 LayerNorm statistics, exact-erf GELU.
"""
import math

import torch

from ...hip import arena as harena
from ...hip import attention as hattn
from ...hip import elementwise as ew
from ...hip import linear as hlin

_bf16, _f32 = torch.bfloat16, torch.float32
_BLOCK_PARAMS = ("ln_1.weight", "ln_1.bias", "attn.c_qkv.weight", "attn.c_qkv.bias", "attn.c_proj.weight", "attn.c_proj.bias",
                 "ln_2.weight", "ln_2.bias", "mlp.c_fc.weight", "mlp.c_fc.bias", "mlp.c_proj.weight", "mlp.c_proj.bias")


# ---- token-dimension split of the forward / data-gradient GEMMs ---------------------------------------------------------------
# The token count of a step is B x (N + 1): the time token makes it 64 x 513 = 32,832 = 128.25 tiles of 256 rows, and hipBLASLt
# loses 12 % on that quarter tile (tools/probes/gpu_dev_gemm_m.py: 1683 -> 1475 us per block for the eight GEMMs).  A GEMM does not
# care which rows it gets, so every [T, K] x [K, N] product is issued as one call on the last T % 256 rows and one on the rest.
# Measured in situ (same box, tuned solutions for both): 10.78 -> 11.13 steps/s.  (Tried: the small call on a second stream --
# its workgroups queue behind the large GEMM's and the join waits for them: no gain at either stream priority.  Tried: a
# hand-written kernel for the 64 left-over rows of the forward -- one pass over the weights straight into MFMA fragments,
# split-K over 8 waves: 11.5 us per call in situ against the tuned library's 12.8 us, 0.2-0.5 % of a step; both are bound by
# pulling 2-8 MB of cold weights through a launch that is over before it has filled the chip.  Not kept.)
import os

_SPLIT = 0 if os.environ.get("NPCD_NO_GEMM_SPLIT") else 256       # (the env switches exist for A/B measurements)
_SPLIT_MIN = int(os.environ.get("NPCD_GEMM_SPLIT_MIN", "16000"))
# the left-over rows go BEHIND the large call: they then find the weights in L2 / MALL (82.08 against 82.32 ms per step, two
# alternating rounds, tools/run_split_order_ab.sh; NPCD_GEMM_SPLIT_SMALL_FIRST=1 restores the old order)
_SPLIT_BIG_FIRST = not os.environ.get("NPCD_GEMM_SPLIT_SMALL_FIRST")
_ATTN_COLSUM = not os.environ.get("NPCD_NO_ATTN_COLSUM")      # c_qkv bias gradient from the attention backward itself (A/B switch)
_SUM_KERNEL = not os.environ.get("NPCD_NO_SUM_KERNEL")          # the weight-gradient partials summed by csrc/elementwise.hip (A/B switch)
# opt-in: the weight gradients on the own split-T kernel (csrc/gemm.hip: at parity with the library's row-split form -- both are
# bound by the L2 -> LDS stream of a 256 x 256 tile -- bitwise reproducible, no library call)
_OWN_WGRAD = bool(os.environ.get("NPCD_OWN_WGRAD"))
# opt-in: the data gradient of mlp.c_proj as the own NT product with the GELU backward and the c_fc bias-gradient column sums in its
# epilogue (csrc/gemm_nt.hip, docs/experiments.md R4.1) on the full 256-row tiles of the token range; the remainder rows keep the
# library product + the separate kernel.  Reads a TRANSPOSED 16-bit copy of the weight, refreshed per backward.
_OWN_DGELU = bool(os.environ.get("NPCD_OWN_DGELU"))


# Below _SPLIT_MIN only the products whose OUTPUT is at least this wide are split (round 5, R5.18): at T = 4,104 / 8,208 the left-over
# row of 256-row tiles turns the 4,096-wide products (c_fc forward, mlp.c_proj data gradient) from 256 / 512 tiles = one / two full
# rounds into 272 / 528.  Same box, alternating processes: a rank's step at per-GPU batch 8 18.51-18.57 -> 18.18-18.36 ms, at 16
# 28.71-28.75 -> 27.96-27.99 ms; with 3,072 (c_qkv too) 18.34.  NPCD_GEMM_SPLIT_WIDE_N=0 switches it off.
_SPLIT_WIDE_N = int(os.environ.get("NPCD_GEMM_SPLIT_WIDE_N", "4096"))


def _split_gemm(fn, T, N=0):
    """fn(rows) enqueues the product for a row range into a shared output."""
    # below ~16 k tokens the large call is short enough that the extra launch costs what the quarter tile did (measured at
    # per-GPU batch 8 and 16: 19.9 vs 20.2 ms and 30.8 vs 31.1 ms per step; at per-GPU batch 32 the split wins: 50.4 -> 48.9 ms)
    wide = _SPLIT_WIDE_N > 0 and N >= _SPLIT_WIDE_N and T > _SPLIT
    Tm = T - T % _SPLIT if (_SPLIT and (T >= _SPLIT_MIN or wide)) else T
    if Tm == 0 or Tm == T:
        fn(slice(0, T))
        return
    if _SPLIT_BIG_FIRST:
        fn(slice(0, Tm))
        fn(slice(Tm, T))
        return
    fn(slice(Tm, T))
    fn(slice(0, Tm))


# The token counts of one rank of a 4- / 8-GPU job (8,208 / 4,104 rows): the square product of the block (attn.c_proj, N = K = 1,024:
# 64-68 tiles of 256 x 256 for 256 CUs in the library's form) goes to the own 128 x 128-tile kernel, which measured 1.3 x the tuned
# library there (17.6-18.1 against 23-29 us at T = 4,104; docs/experiments.md R4.7) and loses on every other shape of the block --
# selected by shape, NPCD_NO_LIN128=1 switches it off (A/B).
_LIN128 = not os.environ.get("NPCD_NO_LIN128")
_LIN128_MAX_T = int(os.environ.get("NPCD_LIN128_MAX_T", "4200"))


def _linear(bias, x, w16):
    """x [T, K] bf16, w16 [N, K] bf16, bias [N] bf16 -> x @ w16^T + bias, [T, N] bf16."""
    T = x.shape[0]
    out = harena.empty((T, w16.shape[0]), x.dtype, x.device)
    N, K = w16.shape
    if (_LIN128 and N == 1024 and K == 1024 and T <= _LIN128_MAX_T and x.is_cuda and x.is_contiguous() and w16.is_contiguous()
            and bias.is_contiguous() and x.dtype == w16.dtype == bias.dtype and hlin.supported128(T, N, K)):
        return hlin.linear128_fwd(x, w16, bias, out)
    wt = w16.t()
    _split_gemm(lambda r: torch.addmm(bias, x[r], wt, out=out[r]), T, N)
    return out


def _dgrad(dy, w16):
    """dy [T, N] bf16, w16 [N, K] bf16 -> dy @ w16, [T, K] bf16."""
    T = dy.shape[0]
    out = harena.empty((T, w16.shape[1]), dy.dtype, dy.device)
    _split_gemm(lambda r: torch.mm(dy[r], w16, out=out[r]), T, w16.shape[1])
    return out


def _wgrad(dy, x, out):
    """out (fp32 view of the flat gradient) = dy^T @ x, fp32 accumulate AND fp32 output.

    The reduction dimension is the token count T (32,832 at cfg-D) while the output is only 1-4 M
    elements, i.e. 16-64 tiles of 256x256 for 256 CUs: the GEMM is split along T into S batched slices
    (hipBLASLt batched GEMM, S x more tiles in flight) whose fp32 partials are summed straight into the
    gradient view.  Measured on MI355X: c_qkv 350 -> 250 us, attn.c_proj 200 -> 87 us, c_fc 327 -> 259 us,
    mlp.c_proj 308 -> 255 us.
    (Tried: the weight gradients on a second HIP stream, so that they run beside the HBM-bound GELU / LayerNorm backward
    kernels of the critical path -- 84.39 -> 84.14 ms per step on the same box, i.e. nothing: the GEMM's workgroups hold
    every CU and the other stream's kernels are dispatched as they drain.)"""
    if _OWN_WGRAD and ew.wgrad(dy, x, out):
        return
    T = dy.shape[0]
    small = out.numel() <= (1 << 20)
    S = 8 if small else 4
    # keep >= 4096 rows per slice (2048 for the 1024 x 1024 output, 16 tiles: tools/probes/gpu_dev_wgrad_split.py at T = 4104 / 8208): short
    # reductions need no split
    S = min(S, max(1, T // (2048 if small else 4096)))
    while S > 1 and T % S:
        S //= 2
    if S == 1:
        torch.mm(dy.t(), x, out_dtype=_f32, out=out)
        return
    part = torch.bmm(dy.view(S, T // S, -1).transpose(1, 2), x.view(S, T // S, -1), out_dtype=_f32)
    if not (_SUM_KERNEL and ew.sum_slices(part, out)):
        torch.sum(part, dim=0, out=out)


# The weight gradients of a block on a second HIP stream, beside the data-gradient products and the HBM-bound GELU / LayerNorm backward
# kernels of the critical path; joined before the block's gradients are handed to the reducer.  Round 3 measured nothing at per-GPU
# batch 64 (docs/experiments.md R3.9); round 5 at the batches of a rank of the strong-scaling job, where a product is one partial
# round of tiles: 19.1 -> 18.6 ms per step at batch 8, 29.3 -> 28.7 at 16, 46.0 -> 45.6 at 32; at 64 it measures 84.4-84.5 against
# 82.9-84.3 ms (R5.11) -- the default below _WGRAD_STREAM_MAX_T token rows since.  NPCD_WGRAD_STREAM=0 / 1 forces it off / on at
# every size, NPCD_WGRAD_STREAM_PRIO=<priority> sets the side stream's priority.  Same kernels in the same order per tensor: the
# gradients are the same bits either way (tests/test_gpu_fused.py).
def _parse_wgrad_stream(value):
    """NPCD_WGRAD_STREAM -> (enabled, max token rows): unset = on below 20,000 rows, "0" = off, "1" = on at every size, any other
    integer = on up to that many rows.  Parsed ONCE, here (it had been read twice with two meanings)."""
    if value is None or value == "":
        return True, 20000
    if value == "0":
        return False, 0
    if value == "1":
        return True, 1 << 62
    return True, int(value)


_WGRAD_STREAM, _WGRAD_STREAM_MAX_T = _parse_wgrad_stream(os.environ.get("NPCD_WGRAD_STREAM"))
_side = {}


def _side_stream(device):
    s = _side.get(device)
    if s is None:
        s = _side[device] = torch.cuda.Stream(device=device, priority=int(os.environ.get("NPCD_WGRAD_STREAM_PRIO", "0")))
    return s


def _wgrad_side_ok(T):
    return _WGRAD_STREAM and T <= _WGRAD_STREAM_MAX_T


# Round 6: below _WGRAD_GROUP_MAX_T token rows the four weight gradients of a block are ONE launch of the own kernel (csrc/gemm.hip,
# npcd_wgrad_group: a workgroup per 256 x 256 tile over the whole token range; 192 tiles at width 1,024), beside the critical path on the
# side stream.  NPCD_WGRAD_GROUP=0 switches it off (the library's products), NPCD_WGRAD_GROUP_MAX_T sets the size limit.
_WGRAD_GROUP = os.environ.get("NPCD_WGRAD_GROUP", "1") != "0"
_WGRAD_GROUP_MAX_T = int(os.environ.get("NPCD_WGRAD_GROUP_MAX_T", "17000"))
# Round 6: the main stream no longer waits for the side stream once per block (a cross-queue dependency costs ~10 us of an idle chip per
# block even when it is already satisfied: tools/step_timeline.py on the rank step).  A block's gradients are handed to the reducer FROM the
# side stream instead -- behind the block's weight gradients in that stream's order, and behind the block's critical path through the fork's
# wait -- so a collective launched there depends on exactly what it reads; the main stream joins the side stream once, at the end of the
# backward.  NPCD_WGRAD_JOIN_PER_BLOCK=1 restores the per-block join (A/B).
_JOIN_PER_BLOCK = bool(os.environ.get("NPCD_WGRAD_JOIN_PER_BLOCK"))
# Round 6: below _STEP_ARENA_MAX_T token rows the ~740 buffers a step of the fused node allocates are kept and handed out again in the
# same order by the next step (npcd/hip/arena.py): 1.6 ms of host time per step at per-GPU batch 8.  The memory of a step then stays
# allocated between steps (at 4,104 rows ~8 GB; the caching allocator would keep most of it anyway).  NPCD_STEP_ARENA=0: off.
_STEP_ARENA = os.environ.get("NPCD_STEP_ARENA", "1") != "0"
_STEP_ARENA_MAX_T = int(os.environ.get("NPCD_STEP_ARENA_MAX_T", "20000"))


def _wgrad_fork(pending, device, before=None):
    """The weight-gradient products of ONE block -- (dy, x, out) triples -- on the side stream, behind everything the current stream has
    been given so far: one fork per block (per product it cost the host ~40 us of stream bookkeeping, 3.8 ms of a rank's step at
    per-GPU batch 8, which is host-bound: tools/probes/gpu_dev_b8_hostprofile.py).  `before()`: called on the side stream ahead of the
    products (the hand-over of the block above to the reducer)."""
    side = _side_stream(device)
    with torch.cuda.stream(side):
        if before is not None:
            before()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for dy, x, out in pending:
            dy.record_stream(side)
            x.record_stream(side)
        if not (_WGRAD_GROUP and pending and pending[0][0].shape[0] <= _WGRAD_GROUP_MAX_T and ew.wgrad_group(pending)):
            for dy, x, out in pending:
                _wgrad(dy, x, out)
    pending.clear()


def _wgrad_join(device):
    torch.cuda.current_stream().wait_stream(_side_stream(device))


def _no_engine():
    return None


import contextlib

_NULL = contextlib.nullcontext()


class FusedBackboneEngine:
    """Views into the flat fp32 parameter / gradient buffers and the bf16 shadow for every block."""

    # the engine belongs to ONE trainer's flat buffers: a deep copy / pickle of the backbone carries None (the module path) instead
    def __deepcopy__(self, memo):
        return None

    def __reduce__(self):
        return (_no_engine, ())

    def __init__(self, backbone, flat, shadow, reducer=None):
        self.heads = backbone.resblocks[0].attn.heads
        self.width = backbone.width
        self.reducer = reducer
        offset = {id(p): off for p, off in zip(flat.params, flat.offsets)}
        self.blocks = []
        for blk in backbone.resblocks:
            named = dict(blk.named_parameters())
            entry = {"params": [named[n] for n in _BLOCK_PARAMS]}
            for n in _BLOCK_PARAMS:
                p = named[n]
                off = offset[id(p)]
                key = n.replace(".", "_")
                entry[key] = p.data                                              # fp32 master (LN affine is used in fp32)
                entry[key + "_16"] = shadow[off:off + p.numel()].view_as(p)     # bf16 shadow (GEMM operands)
                entry[key + "_g"] = flat.grad[off:off + p.numel()].view_as(p)   # fp32 gradient
            self.blocks.append(entry)

        # The GEMMs read the bf16 SHADOW, which only the optimizer kernel (and load_trainer_state) refresh.  Any other in-place
        # write to a parameter -- model.load_state_dict, copying EMA weights in to sample from them -- bumps the parameter's
        # version counter (the optimizer kernel writes through raw pointers and does not): the stamp below notices that and the
        # shadow is re-cast from the fp32 masters before the next forward instead of silently running stale Linear weights
        # beside fresh LayerNorm ones.
        self.block_ranges, self.wait_range = None, None      # set by a trainer with lazily gathered parameters (engine.py)
        self._flat, self._shadow = flat, shadow
        self.dtype = shadow.dtype                            # the run's 16-bit activation type: bf16, or f16 (with loss scaling)
        self._stamped = [p for e in self.blocks for p in e["params"]]
        self._stamp = self._versions()
        # counts the writes to the 16-bit shadow (optimizer pass, parameter gather, re-cast): what a cached derivative of the shadow
        # -- the transposed mlp.c_proj weights of the NPCD_OWN_DGELU launch -- is stamped with
        self.shadow_epoch = 0
        # the buffers of a step, reused by the next one (npcd.hip.arena) -- for the token counts of a rank of the strong-scaling job,
        # where the host side of a step is as long as its GPU side; NPCD_STEP_ARENA=0 switches it off
        self.arena = harena.StepArena() if _STEP_ARENA else None

    def shadow_written(self):
        self.shadow_epoch += 1

    def _versions(self):
        return [p._version for p in self._stamped]

    def sync_shadow(self, force=False):
        """Re-cast fp32 masters -> bf16 shadow if a parameter was written outside the optimizer since the last check."""
        v = self._versions()
        if force or v != self._stamp:
            if self.wait_range is not None:
                self.wait_range()                                # every pending parameter gather first: the whole shadow is re-cast
            for p, off in zip(self._flat.params, self._flat.offsets):
                if p.data_ptr() != self._flat.flat.data_ptr() + off * self._flat.flat.element_size():
                    raise RuntimeError("a parameter of the fused backbone was re-homed outside the trainer's flat buffer "
                                       "(p.data replaced): build a new DiffusionTrainer for this model")
            ew.cast_f32_bf16(self._flat.flat, self._shadow)
            self.shadow_written()
            self._stamp = v

    def __call__(self, x):
        self.sync_shadow()
        return _BackboneFn.apply(x, self)


class InferenceWeights:
    """bf16 copies of the block weights for the forward-only path of a backbone that is not attached to a trainer
    (sampling from a loaded checkpoint).  Rebuilt when a parameter was written to or replaced since the copy was taken."""

    def __init__(self, backbone):
        self.heads = backbone.resblocks[0].attn.heads
        self.params = [[dict(blk.named_parameters())[n] for n in _BLOCK_PARAMS] for blk in backbone.resblocks]
        self.stamp, self.blocks = None, None

    def current(self):
        stamp = [(p._version, p.data_ptr()) for ps in self.params for p in ps]
        if stamp != self.stamp:
            self.blocks = []
            for ps in self.params:
                e = {}
                for n, p in zip(_BLOCK_PARAMS, ps):
                    key = n.replace(".", "_")
                    e[key] = p.data
                    if not n.startswith("ln_"):
                        e[key + "_16"] = p.data.to(_bf16)
                self.blocks.append(e)
            self.stamp = stamp
        return self.blocks


class InferenceWeightsX2:
    """The block weights of a backbone in the split-operand form of the fp32-class forward (csrc/split.hip): per Linear layer
    [Wh | Wh | Wl] bf16 [N, 3 K] with W = Wh + Wl, the biases and LayerNorm parameters in fp32.  Rebuilt when a parameter was written
    to or replaced since the copy was taken."""

    def __init__(self, backbone):
        self.heads = backbone.resblocks[0].attn.heads
        self.params = [[dict(blk.named_parameters())[n] for n in _BLOCK_PARAMS] for blk in backbone.resblocks]
        self.stamp, self.blocks = None, None

    def current(self):
        stamp = [(p._version, p.data_ptr()) for ps in self.params for p in ps]
        if stamp != self.stamp:
            self.blocks = []
            for ps in self.params:
                e = {}
                for n, p in zip(_BLOCK_PARAMS, ps):
                    key = n.replace(".", "_")
                    w = p.data.to(_f32)
                    if n.endswith("weight") and not n.startswith("ln_"):
                        hi = w.to(_bf16)
                        lo = (w - hi.float()).to(_bf16)
                        e[key + "_x3"] = torch.cat((hi, hi, lo), dim=1).contiguous()
                    else:
                        e[key] = w.contiguous()
                self.blocks.append(e)
            self.stamp = stamp
        return self.blocks


def backbone_forward_x2(x, blocks, heads):
    """Forward only, in the reference's fp32 class (the sampler, diffusion_model.py:108-133, runs the denoiser in fp32): residual stream,
    LayerNorm, attention (the fp32 matrix-instruction kernels of csrc/attention.hip) and GELU in fp32; the four Linear layers of a
    block as ONE bf16 library GEMM each over the three cross products of split operands (csrc/split.hip, npcd_split3_bf16), fp32
    accumulation and output -- 3e-6 relative per product against float64 (an fp32 GEMM: 4e-7), at 2.3-3.5 x the fp32 GEMM's rate.
    x [B, n, W] fp32 -> [B, n, W] fp32.  `blocks`: InferenceWeightsX2.current()."""
    import torch.nn.functional as F
    B, n, W = x.shape
    T, d = B * n, W // heads
    scale = 1.0 / math.sqrt(d)
    f32 = _f32

    def lin(a3, e, name):           # [T, 3 K] bf16 x [N, 3 K]^T -> [T, N] fp32 (bias added by the consumer)
        return torch.mm(a3, e[name + "_weight_x3"].t(), out_dtype=f32)
    def ln_split(xs, e, ln, o=None, bias=None):      # (xs + o + bias, split(LayerNorm(.))): one kernel, or torch + split3 for other widths
        r = ew.add_ln_split3(xs, e[ln + "_weight"], e[ln + "_bias"], o, bias)
        if r is not None:
            return r
        if o is not None:
            xs = xs + o + bias
        return xs, ew.split3(F.layer_norm(xs, (W,), e[ln + "_weight"], e[ln + "_bias"]))
    with torch.autocast("cuda", enabled=False):
        xs = x.reshape(T, W).contiguous()
        o, ob = None, None               # the previous block's mlp.c_proj output and bias: added by the next LayerNorm kernel
        for e in blocks:
            xs, y1 = ln_split(xs, e, "ln_1", o, ob)
            qkv = lin(y1, e, "attn_c_qkv").add_(e["attn_c_qkv_bias"]).view(B, n, heads, 3 * d)
            a, _ = hattn._fwd(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], scale)
            xs, y2 = ln_split(xs, e, "ln_2", lin(ew.split3(a.reshape(T, W)), e, "attn_c_proj"), e["attn_c_proj_bias"])
            h = lin(y2, e, "mlp_c_fc")
            o, ob = lin(ew.split3(h, bias=e["mlp_c_fc_bias"], gelu=True), e, "mlp_c_proj"), e["mlp_c_proj_bias"]
        xs = xs + o + ob
    return xs.view(B, n, W)


def backbone_forward(x, blocks, heads, dtype=_bf16):
    """Forward only (sampler, evaluation): the kernels of the training forward, nothing kept for a backward.
    x [B, n, W] fp32 -> [B, n, W] fp32.  `dtype`: the 16-bit type of the weights in `blocks`."""
    B, n, W = x.shape
    T, d = B * n, W // heads
    scale = 1.0 / math.sqrt(d)
    with torch.autocast("cuda", enabled=False):
        xs = x.reshape(T, W).contiguous()
        delta = None
        for e in blocks:
            x1, y1, _, _ = ew.add_ln_fwd(xs, delta, e["ln_1_weight"], e["ln_1_bias"], dtype=dtype)
            x_cur = xs if x1 is None else x1
            q4 = _linear(e["attn_c_qkv_bias_16"], y1, e["attn_c_qkv_weight_16"]).view(B, n, heads, 3 * d)
            a, _ = hattn._fwd(q4[..., :d], q4[..., d:2 * d], q4[..., 2 * d:], scale)
            o = _linear(e["attn_c_proj_bias_16"], a.view(T, W), e["attn_c_proj_weight_16"])
            xs, y2, _, _ = ew.add_ln_fwd(x_cur, o, e["ln_2_weight"], e["ln_2_bias"])
            g = ew.gelu_fwd(_linear(e["mlp_c_fc_bias_16"], y2, e["mlp_c_fc_weight_16"]))
            delta = _linear(e["mlp_c_proj_bias_16"], g, e["mlp_c_proj_weight_16"])
        out = xs + delta
    return out.view(B, n, W)


class _BackboneFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, eng):
        B, n, W = x.shape
        T = B * n
        arena = eng.arena if (eng.arena is not None and x.is_cuda and T <= _STEP_ARENA_MAX_T) else None
        token = arena.begin() if arena is not None else 0
        if not token:
            arena = None                       # (another step of this engine still holds its buffers: plain allocations for this one)
        ctx.arena, ctx.arena_token = arena, token
        if arena is None:
            return _BackboneFn._forward(ctx, x, eng)
        ctx.arena_guard = harena.StepGuard(arena, token)      # ends the step if the graph is dropped without a backward
        try:
            with arena.active():
                return _BackboneFn._forward(ctx, x, eng)
        except BaseException:
            arena.end(token)
            raise

    @staticmethod
    def _forward(ctx, x, eng):
        B, n, W = x.shape
        T, H = B * n, eng.heads
        d = W // H
        scale = 1.0 / math.sqrt(d)
        saved = []
        with torch.autocast("cuda", enabled=False):
            xs = x.reshape(T, W).contiguous()
            delta = None
            for bi, e in enumerate(eng.blocks):
                if eng.wait_range is not None:
                    eng.wait_range(*eng.block_ranges[bi])        # this block's parameters (gathered lazily by the sharded optimizer)
                x1, y1, mean1, rstd1 = ew.add_ln_fwd(xs, delta, e["ln_1_weight"], e["ln_1_bias"], dtype=eng.dtype)
                x_cur = xs if x1 is None else x1
                qkv = _linear(e["attn_c_qkv_bias_16"], y1, e["attn_c_qkv_weight_16"])
                q4 = qkv.view(B, n, H, 3 * d)
                a, lse = hattn._fwd(q4[..., :d], q4[..., d:2 * d], q4[..., 2 * d:], scale)
                a = a.view(T, W)
                o = _linear(e["attn_c_proj_bias_16"], a, e["attn_c_proj_weight_16"])
                x2, y2, mean2, rstd2 = ew.add_ln_fwd(x_cur, o, e["ln_2_weight"], e["ln_2_bias"])
                h = _linear(e["mlp_c_fc_bias_16"], y2, e["mlp_c_fc_weight_16"])
                g = ew.gelu_fwd(h)
                delta = _linear(e["mlp_c_proj_bias_16"], g, e["mlp_c_proj_weight_16"])
                saved.append((x_cur, mean1, rstd1, y1, qkv, a, lse, x2, mean2, rstd2, y2, h, g))
                xs = x2
            out = xs + delta                       # fp32 + bf16 -> fp32  (a torch operator: the node's output is never an arena buffer)
        ctx.eng, ctx.saved, ctx.dims, ctx.scale = eng, saved, (B, n, W, H, d), scale
        return out.view(B, n, W)

    @staticmethod
    def backward(ctx, dout):
        arena = ctx.arena
        if arena is None:
            return _BackboneFn._backward(ctx, dout)
        try:
            with arena.active():           # (the engine's worker thread: the arena context is per thread)
                return _BackboneFn._backward(ctx, dout)
        finally:
            arena.end(ctx.arena_token)

    @staticmethod
    def _backward(ctx, dout):
        eng, (B, n, W, H, d), scale = ctx.eng, ctx.dims, ctx.scale
        T = B * n
        with torch.autocast("cuda", enabled=False):
            dx = dout.reshape(T, W).contiguous().float()
            dxb = dx.to(eng.dtype)
            last = eng.blocks[-1]
            last["mlp_c_proj_bias_g"].copy_(dx.sum(dim=0))
            side, pending = _wgrad_side_ok(T), []

            reducing = eng.reducer is not None and eng.reducer.active

            def ready(entry):
                if reducing:
                    for p in entry["params"]:
                        eng.reducer.mark_ready(p)
            for bi in range(len(eng.blocks) - 1, -1, -1):
                e = eng.blocks[bi]
                x_cur, mean1, rstd1, y1, qkv, a, lse, x2, mean2, rstd2, y2, h, g = ctx.saved[bi]
                ctx.saved[bi] = None
                sums = ew.ColsumBatch()            # this block's 8 bias / LN-affine column sums: one finalize
                wg = pending.append if side else (lambda t: _wgrad(*t))      # weight gradients: now, or queued for the side stream
                # ---- MLP branch: x3 = x2 + c_proj(gelu(c_fc(ln_2(x2)))) ------------------------------
                Tm = T - T % 256
                w2 = e["mlp_c_proj_weight_16"]
                # (the own launch only for what npcd_linear_dgelu_bwd takes -- contiguous 16-bit operands of one type; anything else
                # goes through the library product + gelu_bwd below instead of raising, like NPCD_ERR_UNSUPPORTED would, ADVICE r4.
                # The transposed 16-bit copy of the weight is CACHED per block and stamped with the engine's shadow epoch: it is rebuilt
                # (into the same 8-MB buffer) only after the shadow was written -- once per optimizer step in a plain loop, not at
                # all between the backwards of a gradient-accumulation loop or of repeated evaluations of one model state.)
                if (_OWN_DGELU and Tm > 0 and hlin.supported(Tm, 4 * W, W) and dxb.is_contiguous() and h.is_contiguous()
                        and w2.is_contiguous() and dxb.dtype == w2.dtype == h.dtype and dxb.dtype in (torch.bfloat16, torch.float16)):
                    wT = e.get("mlp_c_proj_weight_16T")
                    if wT is None or e.get("mlp_c_proj_weight_16T_epoch") != eng.shadow_epoch or wT.dtype != w2.dtype:
                        wT = e["mlp_c_proj_weight_16T"] = hlin.transpose16(w2, out=wT if wT is not None and wT.dtype == w2.dtype else None)
                        e["mlp_c_proj_weight_16T_epoch"] = eng.shadow_epoch
                    dh = harena.empty_like(h)
                    extra = ew.lib().npcd_colsum_blocks(T - Tm) if Tm < T else 0
                    _, part, rows = hlin.linear_dgelu_bwd(dxb[:Tm], wT, h[:Tm], out=dh[:Tm], extra_part_rows=extra)
                    if Tm < T:       # remainder rows: library product + the separate kernel, partial rows behind the own kernel's
                        dg = torch.mm(dxb[Tm:], e["mlp_c_proj_weight_16"])
                        ew.gelu_bwd(dg, h[Tm:], None, out=dh[Tm:], part_rows=part[rows:])
                        del dg
                    sums.add(part, rows + extra, 4 * W, e["mlp_c_fc_bias_g"])
                    wg((dxb, g, e["mlp_c_proj_weight_g"]))
                    del g, h
                else:
                    dg = _dgrad(dxb, e["mlp_c_proj_weight_16"])
                    wg((dxb, g, e["mlp_c_proj_weight_g"]))
                    dh = ew.gelu_bwd(dg, h, e["mlp_c_fc_bias_g"], batch=sums)
                    del dg, g, h
                dy2 = _dgrad(dh, e["mlp_c_fc_weight_16"])
                wg((dh, y2, e["mlp_c_fc_weight_g"]))
                del dh, y2
                dx2, dx2b = ew.ln_bwd(dy2, x2, mean2, rstd2, e["ln_2_weight"], dx, e["ln_2_weight_g"], e["ln_2_bias_g"],
                                      e["attn_c_proj_bias_g"], batch=sums)
                del dy2, x2, dx, dxb
                # ---- attention branch: x2 = x + c_proj(attn(c_qkv(ln_1(x)))) ---------------------------
                da = _dgrad(dx2b, e["attn_c_proj_weight_16"])
                wg((dx2b, a, e["attn_c_proj_weight_g"]))
                dqkv = harena.empty_like(qkv)
                q4, g4 = qkv.view(B, n, H, 3 * d), dqkv.view(B, n, H, 3 * d)
                # the c_qkv bias gradient (column sums of dqkv) is a by-product of the attention backward's row stores
                cpart = hattn.colsum_part_for(g4[..., :d]) if _ATTN_COLSUM else None
                hattn._bwd(q4[..., :d], q4[..., d:2 * d], q4[..., 2 * d:], a.view(B, n, H, d), da.view(B, n, H, d), lse,
                           g4[..., :d], g4[..., d:2 * d], g4[..., 2 * d:], scale, colsum_part=cpart[0] if cpart else None)
                del da, a, qkv, dx2b
                if cpart:
                    sums.add(cpart[0], cpart[1], 3 * W, e["attn_c_qkv_bias_g"])
                else:
                    ew.colsum_bf16(dqkv, e["attn_c_qkv_bias_g"], batch=sums)
                dy1 = _dgrad(dqkv, e["attn_c_qkv_weight_16"])
                wg((dqkv, y1, e["attn_c_qkv_weight_g"]))
                del dqkv, y1
                prev_bias_g = eng.blocks[bi - 1]["mlp_c_proj_bias_g"] if bi > 0 else None
                # (the last block's dx is what the node returns: allocated outside the arena, like everything that outlives the step)
                with (harena.paused() if bi == 0 else _NULL):
                    dx, dxb = ew.ln_bwd(dy1, x_cur, mean1, rstd1, e["ln_1_weight"], dx2, e["ln_1_weight_g"], e["ln_1_bias_g"],
                                        prev_bias_g, want_bf16=bi > 0, batch=sums)
                del dy1, dx2
                sums.flush()
                if side and _JOIN_PER_BLOCK:
                    # (the round-5 order, A/B) the block ABOVE had this block's critical path to finish its weight gradients beside: join
                    # them, hand its gradients on, then start this block's four products on the side stream
                    if bi + 1 < len(eng.blocks):
                        _wgrad_join(dx.device)
                        ready(eng.blocks[bi + 1])
                    _wgrad_fork(pending, dx.device)
                elif side:
                    # the block ABOVE is handed to the reducer ON the side stream, behind its own weight gradients (and, through the
                    # wait of the fork that started them, behind its critical path): the main stream does not wait; then this block's
                    # four products start there
                    above = eng.blocks[bi + 1] if (reducing and bi + 1 < len(eng.blocks)) else None
                    _wgrad_fork(pending, dx.device, before=None if above is None else (lambda: ready(above)))
                else:
                    ready(e)       # this block's gradients are final (mlp.c_proj.bias was finished by the block above / the tail)
            if side:
                _wgrad_join(dx.device)
                ready(eng.blocks[0])
        return dx.view(B, n, W), None
