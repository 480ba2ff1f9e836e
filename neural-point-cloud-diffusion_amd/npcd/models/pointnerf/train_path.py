"""Training-mode rendering (stage-1 autodecoder training): a few hundred randomly chosen rays
per view with jittered depth samples, differentiable w.r.t. the neural-point features and the field weights.

Where things run: ray generation with its cube limits, the neighbour query, the MLP input of every (point, neighbour) pair,
the inverse-distance aggregation, the LeakyReLU backward + bias gradients of the per-pair layers and the ray march are HIP
kernels with hand-written backward (npcd.hip.render, csrc/geometry.hip, csrc/pairs.hip); the Linear layers are library GEMMs
(weight gradients split along the ~1e6 rows, _RowSplitLinear); the point-level heads, the ray selection and the losses are
torch operators on the device.  `positional_encoding`, `depths_from_points` and `ray_march` below are the torch formulations
the kernels are tested against.  Nothing here runs on the CPU: the neighbour query fails loudly without the HIP library.

Reference: renderers/renderer.py:49-77,96-110,120-185,202-268; renderers/volume_renderer.py:23-92; fields/field.py:77-152;
fields/aggregators/aggregator.py:78-119; fields/aggregators/mlp.py:36-125; fields/positional_encoder.py:7-23;
renderers/math_utils.py:46-97.  The reference draws its random numbers inline; every draw can be injected (`rng`
dictionary: ray_perm, jitter, valid_perm) so that tests can replay the reference's numbers.
"""
import math
import os
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from ...hip import elementwise as ew
from ...hip import render as hr
from ...utils import AttrDict


# ------------------------------------------------------------------------------------------------- rays
def jittered_depths(start: torch.Tensor, end: torch.Tensor, S: int, jitter: Optional[torch.Tensor]):
    """start/end [..., 1] -> depths [..., S]: S evenly spaced samples, each moved forward by U(0,1) of one spacing."""
    steps = torch.arange(S, dtype=torch.float32, device=start.device) / (S - 1)
    dep = start + steps * (end - start)
    if jitter is not None:
        dep = dep + jitter * ((end - start) / (S - 1))
    return dep


# ------------------------------------------------------------------------------------------------- field (autograd)
def positional_encoding(x: torch.Tensor, n_freqs: int) -> torch.Tensor:
    """[x, sin(x_c 2^i pi) (i < n), cos(x_c 2^i pi) (i < n) for c in x, y, z] -> 3 + 6 n columns."""
    bands = (2 ** torch.arange(n_freqs, device=x.device)) * torch.pi
    spec = x[..., None] * bands
    return torch.cat((x, torch.cat((spec.sin(), spec.cos()), dim=-1).flatten(start_dim=-2)), dim=-1)


_SMALL_WGRAD = os.environ.get("NPCD_NO_SMALL_WGRAD", "") != "1"     # A/B switch: the library GEMM for the heads' weight gradients
_ROWSPLIT_MIN = int(os.environ.get("NPCD_ROWSPLIT_MIN", "32768"))      # rows from which a Linear layer takes the row-split weight gradient


class _RowSplitLinear(torch.autograd.Function):
    """y = x W^T + b for x of ~10^6 rows and a few hundred columns.  The forward and the data gradient are plain library GEMMs;
    the WEIGHT gradient dW = dy^T x reduces over the ~10^6 rows into a 256 x 256 output, which the library runs on 16 tiles
    (830 us per layer, a third of the whole step): here the rows are split into slices of 16384, the slices go through one
    batched GEMM with fp32 partial outputs, and the partials are summed (150 us).  `dtype` = None: fp32 like the reference;
    torch.bfloat16: operands rounded to bf16, fp32 accumulation, fp32 weight gradient (what autocast would compute)."""
    SLICE = 16384

    @staticmethod
    def forward(ctx, x, weight, bias, dtype, slope=None):
        """slope: negative slope of a LeakyReLU applied to the output (in place), or None"""
        xx = x if dtype is None else x.to(dtype)
        ww = weight if dtype is None else weight.to(dtype)
        bb = bias if dtype is None else bias.to(dtype)
        y = torch.addmm(bb, xx, ww.t())
        if slope is not None:
            y = F.leaky_relu(y, slope, inplace=True)
            ctx.save_for_backward(xx, ww, y)         # the activation OUTPUT carries the sign its backward needs
        else:
            ctx.save_for_backward(xx, ww)
        ctx.in_dtype, ctx.slope = x.dtype, slope
        return y

    @staticmethod
    def backward(ctx, dy):
        xx, ww = ctx.saved_tensors[:2]
        z = ctx.saved_tensors[2] if ctx.slope is not None else None
        dx, dw, db = _RowSplitLinear.linear_backward(dy, xx, ww, z, ctx.slope, ctx.in_dtype, ctx.needs_input_grad[0])
        return dx, dw, db, None, None

    @staticmethod
    def linear_backward(dy, xx, ww, z, slope, in_dtype, need_dx=True):
        """(dx, dW, db) of y = leaky(x W^T + b) given dy, the layer's input xx, weight ww and (with a slope) its OUTPUT z."""
        dy = dy.to(xx.dtype).contiguous()
        db = None
        if slope is not None:
            fused = hr.leaky_bwd_colsum(dy, z, slope) if dy.is_cuda else None    # activation backward + bias gradient: one pass
            if fused is not None:
                dy, db = fused
            else:
                dy = dy * torch.where(z > 0, 1.0, slope).to(dy.dtype)
        dx = torch.mm(dy, ww).to(in_dtype) if need_dx else None
        rows, c = dy.shape[0], _RowSplitLinear.SLICE
        S = rows // c
        f32 = torch.float32
        mixed = xx.dtype != f32
        if dy.shape[1] < 16 and mixed:
            # the heads' last layers (256 -> 1, 256 -> 3): one pass over x with fp32 sums (csrc/elementwise.hip small_wgrad_kernel).
            # Shapes it does not cover (fp16 operands, K not a power of two, NPCD_NO_SMALL_WGRAD=1) keep fp32 accumulation AND
            # fp32 output through the library GEMM -- slow on the host for a handful of output rows
            # (tools/probes/gpu_dev_stage1_hosttrace.py: 10-400 ms per call), but the same numerics as every other weight gradient here
            dw = ew.small_wgrad(dy, xx) if dy.is_cuda and _SMALL_WGRAD else None
            if dw is None:
                dw = torch.mm(dy.t(), xx, out_dtype=f32)
        elif S >= 2:
            head, tail = S * c, rows > S * c
            part = torch.empty((S + int(tail), dy.shape[1], xx.shape[1]), dtype=f32, device=dy.device)
            a, b = dy[:head].view(S, c, -1).transpose(1, 2), xx[:head].view(S, c, -1)
            if mixed:
                torch.bmm(a, b, out_dtype=f32, out=part[:S])
            else:
                torch.bmm(a, b, out=part[:S])
            if tail:
                if mixed:
                    torch.mm(dy[head:].t(), xx[head:], out_dtype=f32, out=part[S])
                else:
                    torch.mm(dy[head:].t(), xx[head:], out=part[S])
            dw = part.sum(dim=0)
        else:
            dw = torch.mm(dy.t(), xx, out_dtype=f32) if mixed else torch.mm(dy.t(), xx)
        if db is None:
            db = dy.sum(dim=0, dtype=f32)
        return dx, dw, db


class _PointLayersX2(torch.autograd.Function):
    """The point-level layers of the stage-1 forward -- last aggregator layer, shape_net, channel_net (8 Linear layers) -- as ONE launch
    in the fp32-class numerics of the per-pair kernels (csrc/points_x2.hip, `save` mode: every hidden activation is written out in
    fp32 for the backward), where the fp32 library path runs eight GEMMs + activation passes over ~2 x 10^5 points.  The backward is
    the chain the separate layers had (_RowSplitLinear.linear_backward: fp32 library GEMMs, fused LeakyReLU backward + bias sums).
    Inputs: G [P, 256] and the 16 parameters in module order; outputs: the heads' PRE-activations [P], [P, 3]."""

    @staticmethod
    def forward(ctx, G, wpack, *params):
        pre, saved = hr.points_x2(wpack, G, save=True)
        ctx.save_for_backward(G, saved, *params[0::2])
        ctx.need_dG = G.requires_grad
        return pre[:, 3].contiguous(), pre[:, :3].contiguous()

    @staticmethod
    def backward(ctx, d_sig, d_rgb):
        G, saved, w8, ws0, ws1, wc0, wc1, wc2, wc3, wc4 = ctx.saved_tensors
        feat, s0, c0, c1, c2, c3 = saved.unbind(0)
        lb, f32, slope = _RowSplitLinear.linear_backward, torch.float32, 0.01
        # colour head: C4 (linear), C3..C0 (LeakyReLU)
        d3, dwc4, dbc4 = lb(d_rgb, c3, wc4, None, None, f32)
        d2, dwc3, dbc3 = lb(d3, c2, wc3, c3, slope, f32)
        d1, dwc2, dbc2 = lb(d2, c1, wc2, c2, slope, f32)
        d0, dwc1, dbc1 = lb(d1, c0, wc1, c1, slope, f32)
        dfc, dwc0, dbc0 = lb(d0, feat, wc0, c0, slope, f32)
        # density head: S1 (linear), S0 (LeakyReLU)
        ds0, dws1, dbs1 = lb(d_sig[:, None], s0, ws1, None, None, f32)
        dfs, dws0, dbs0 = lb(ds0, feat, ws0, s0, slope, f32)
        dG, dw8, db8 = lb(dfc + dfs, G, w8, None, None, f32, ctx.need_dG)
        return (dG, None, dw8, db8, dws0, dbs0, dws1, dbs1, dwc0, dbc0, dwc1, dbc1, dwc2, dbc2, dwc3, dbc3, dwc4, dbc4)


def split2(t: torch.Tensor):
    """fp32 -> (hi, lo) bf16 with t ~ hi + lo: 16 mantissa bits in two halves (the operand form of the fp32-class kernels)."""
    hi = t.to(torch.bfloat16)
    return hi, (t - hi.float()).to(torch.bfloat16)


def x2_linear_forward(x, weight, bias, save: bool):
    """y = x W^T + b in the fp32 class on the bf16 matrix rate: x W^T ~ xh Wh^T + xl Wh^T + xh Wl^T as ONE library GEMM over a three
    times longer contraction, [xh | xl | xh] [Wh | Wh | Wl]^T, with fp32 accumulation and output (the lo x lo term, 2^-18 relative,
    is dropped: ~1e-5 relative per product, like csrc/pairs_mlp.hip at precision 1).  `save` (the caller says so explicitly: inside
    an autograd.Function's forward grad mode is off, so it cannot be probed): True = return the split operands a backward needs,
    False = forward only (rendering), the activation side in one pass.  Returns (y, saved operands or None)."""
    wh, wl = split2(weight)
    if not save and x.is_cuda and x.dtype == torch.float32 and x.shape[1] % 8 == 0:
        # forward only (rendering): the activation side in one pass (csrc/split.hip)
        y = torch.mm(ew.split3(x), torch.cat((wh, wh, wl), dim=1).t(), out_dtype=torch.float32)
        return y + bias, None
    xh, xl = split2(x)
    y = torch.mm(torch.cat((xh, xl, xh), dim=1), torch.cat((wh, wh, wl), dim=1).t(), out_dtype=torch.float32)
    return y + bias, (xh, xl, wh, wl)


class _X2Linear(torch.autograd.Function):
    """A Linear (+ LeakyReLU) layer of the point-level networks in the fp32-class mode of the stage-1 trainer: forward, data gradient
    and weight gradient each as one bf16 library GEMM over the three cross products of the split operands (x2_linear_forward), fp32
    accumulation and fp32 results; the weight gradient's long reduction (3 x ~2e5 rows into 256 x 256) is split into slices like
    _RowSplitLinear's.  OPT-IN (NPCD_STAGE1_X2_HEADS=1): on paper 3 x the rate of fp32 library GEMMs at ~1e-5 relative; measured, the
    step got slower (the split / concatenation passes and the library's fp32-output bf16 kernels at these shapes cost more than the
    fp32 GEMMs they replace): 20.6-21.1 against 16.1 ms."""
    SLICE = 16384

    @staticmethod
    def forward(ctx, x, weight, bias, slope=None):
        y, (xh, xl, wh, wl) = x2_linear_forward(x, weight, bias, save=True)
        if slope is not None:
            y = F.leaky_relu(y, slope, inplace=True)
            ctx.save_for_backward(xh, xl, wh, wl, y)
        else:
            ctx.save_for_backward(xh, xl, wh, wl)
        ctx.slope = slope
        return y

    @staticmethod
    def backward(ctx, dy):
        xh, xl, wh, wl = ctx.saved_tensors[:4]
        f32 = torch.float32
        dy = dy.to(f32).contiguous()
        db = None
        if ctx.slope is not None:
            z = ctx.saved_tensors[4]
            fused = hr.leaky_bwd_colsum(dy, z, ctx.slope) if dy.is_cuda else None
            if fused is not None:
                dy, db = fused
            else:
                dy = dy * torch.where(z > 0, 1.0, ctx.slope)
        if db is None:
            db = dy.sum(dim=0)
        dh, dl = split2(dy)
        dx = None
        if ctx.needs_input_grad[0]:      # dy W ~ dh Wh + dl Wh + dh Wl
            dx = torch.mm(torch.cat((dh, dl, dh), dim=1), torch.cat((wh, wh, wl), dim=0), out_dtype=f32)
        # dW = dy^T x ~ dh^T xh + dl^T xh + dh^T xl: per row slice one batched GEMM over [dh | dl | dh]^T [xh ; xh ; xl], fp32 partials
        rows, c = dy.shape[0], _X2Linear.SLICE
        S = rows // c
        if S >= 2:
            head = S * c
            a = torch.cat((dh[:head].view(S, c, -1), dl[:head].view(S, c, -1), dh[:head].view(S, c, -1)), dim=1).transpose(1, 2)
            b = torch.cat((xh[:head].view(S, c, -1), xh[:head].view(S, c, -1), xl[:head].view(S, c, -1)), dim=1)
            dw = torch.bmm(a, b, out_dtype=f32).sum(dim=0)
            if rows > head:
                dw = dw + torch.mm(torch.cat((dh[head:], dl[head:], dh[head:]), dim=0).t(), torch.cat((xh[head:], xh[head:], xl[head:]), dim=0),
                                   out_dtype=f32)
        else:
            dw = torch.mm(torch.cat((dh, dl, dh), dim=0).t(), torch.cat((xh, xh, xl), dim=0), out_dtype=f32)
        return dx, dw, db, None


def _mlp(seq, x, dtype):
    """nn.Sequential of Linear / LeakyReLU (utils/model.py:22-36).  From a few ten thousand rows on (the per-pair network, and the
    point-level layers of a training batch: ~2 x 10^5 shading points) the Linear layers run as _RowSplitLinear; below that the
    plain modules (under autocast for the bf16 opt-in).  (Round 1 had the threshold at 262,144 rows because lower values cost
    10 ms per step in the bf16 mode: that was the library's fp32-output GEMM for the heads' 1- and 3-row weight gradients, see
    _RowSplitLinear.backward; with those on the bf16-output path the point-level layers gain 1.1 ms of a 9.7 ms step.)"""
    x2 = isinstance(dtype, str) and dtype == "x2"      # fp32-class: wide layers as _X2Linear, the heads' 1- / 3-row layers in plain fp32
    if x2 and (not x.is_cuda or x.shape[0] < 4096):
        x2, dtype = False, None
    if not x2 and x.shape[0] < _ROWSPLIT_MIN:
        with torch.autocast("cuda", dtype=dtype or torch.bfloat16, enabled=dtype is not None):
            return seq(x)
    mods, i = list(seq), 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, torch.nn.Linear):
            act = mods[i + 1] if i + 1 < len(mods) and isinstance(mods[i + 1], torch.nn.LeakyReLU) else None
            slope = None if act is None else act.negative_slope
            if x2 and m.out_features >= 16 and m.in_features % 8 == 0:
                x = _X2Linear.apply(x, m.weight, m.bias, slope)
            else:
                x = _RowSplitLinear.apply(x, m.weight, m.bias, None if x2 else dtype, slope)
            i += 2 if act is not None else 1
        else:
            x = m(x)
            i += 1
    return x


FP32_CLASS = "fp32_class"          # PointNeRFTrainer(mlp_dtype=FP32_CLASS): the split-operand matrix-core mode, an explicit opt-in


def fused_pair_mlp_precision(field, mlp_dtype):
    """Which mode of the matrix-core pair MLP (csrc/pairs_mlp.hip) runs the four non-linear per-pair layers, or None for the library
    GEMMs.  The fused kernels cover the published network (pointnerf.py:174-179: four LeakyReLU layers of width 256 + a linear
    one, 10 frequency bands, feature width 32 or 128).
      mlp_dtype None / torch.float32 / "library": None -- TRUE fp32 like the reference (train_pointnerf.py has no autocast): fp32
          operands and accumulation on library GEMMs for every Linear layer (~6e-8 per operand);
      mlp_dtype "fp32_class": PAIR_MLP_X2 -- every operand as two bf16 halves hi + lo, three matrix instructions per product,
          ~1e-5 relative per product (16 mantissa bits, fp32's range): faster, NOT the reference's arithmetic -- an explicit opt-in
          (it was the default for one round; ADVICE r5: a default below the reference's precision needs the caller's say-so);
      mlp_dtype torch.bfloat16: PAIR_MLP_BF16 -- bf16 operands, narrower still (opt-in)."""
    agg = field.aggregator
    lf = agg.local_field
    if os.environ.get("NPCD_NO_FUSED_PAIR_MLP"):
        return None
    if mlp_dtype is None or mlp_dtype == torch.float32 or mlp_dtype == "library":
        return None
    if isinstance(mlp_dtype, str) and mlp_dtype != FP32_CLASS:
        raise ValueError(f"mlp_dtype {mlp_dtype!r}: None / torch.float32 / 'library' (fp32), 'fp32_class', or torch.bfloat16")
    covered = (agg.in_dim in (32, 128) and agg.n_freqs == 10
               and len(lf) == 9 and all(isinstance(lf[i], torch.nn.Linear) and lf[i].out_features == 256 for i in (0, 2, 4, 6, 8))
               and all(isinstance(lf[i], torch.nn.LeakyReLU) and lf[i].negative_slope == 0.01 for i in (1, 3, 5, 7)))
    if not covered:
        return None
    if mlp_dtype == torch.bfloat16:
        return hr.PAIR_MLP_BF16
    if mlp_dtype == FP32_CLASS:
        return hr.PAIR_MLP_X2
    return None


def point_layers_fused(field, mlp_dtype, n_points=None) -> bool:
    """Does the stage-1 forward run its eight point-level layers as ONE fp32-class launch (csrc/points_x2.hip, _PointLayersX2)?  The
    single predicate shade_autograd, PointNeRFTrainer.describe() and bench.py's labels share (n_points None: the batch-size
    condition, >= 4096 compact points, is left out -- the answer for a training-size batch)."""
    return (fused_pair_mlp_precision(field, mlp_dtype) == hr.PAIR_MLP_X2 and not os.environ.get("NPCD_STAGE1_X2_HEADS")
            and not field.use_dir and not os.environ.get("NPCD_STAGE1_LIBRARY_HEADS") and _point_layers_fusable(field)
            and (n_points is None or n_points >= 4096))


def _point_layers_fusable(field) -> bool:
    """shape_net = Linear(256, 256), LeakyReLU(0.01), Linear(256, 1); channel_net = 4 x [Linear(256, 256), LeakyReLU(0.01)], Linear(256, 3)
    (pointnerf.py:161-162): what csrc/points_x2.hip is built for."""
    def ok(seq, widths):
        lin = [m for m in seq if isinstance(m, torch.nn.Linear)]
        act = [m for m in seq if isinstance(m, torch.nn.LeakyReLU)]
        return ([(m.in_features, m.out_features) for m in lin] == widths and len(act) == len(lin) - 1 and len(seq) == 2 * len(lin) - 1
                and all(a.negative_slope == 0.01 for a in act))
    return ok(field.shape_net, [(256, 256), (256, 1)]) and ok(field.channel_net, [(256, 256)] * 4 + [(256, 3)])


def fused_pair_mlp_ok(field, mlp_dtype) -> bool:
    return fused_pair_mlp_precision(field, mlp_dtype) is not None


def shade_autograd(field, nb_idx: torch.Tensor, pts: torch.Tensor, kp_pos: torch.Tensor, kp_feat: torch.Tensor,
                   point_dir: Optional[torch.Tensor] = None):
    """Compact shading points -> sigma [P] (softplus(x - 1)), rgb [P, 3] (sigmoid), differentiable w.r.t. kp_feat and the
    field's parameters.  nb_idx [P, k] global point indices (-1 pad), pts [P, 3]; positions are constants (the point
    coordinates are frozen in stage 1: pointnerf.py:24,68, aggregators/mlp.py:58-59).  point_dir [P, 3]: the ray direction of
    every point, needed with use_view_dir (fields/mlp.py:67-70: the colour head sees [feat | enc(direction)])."""
    agg = field.aggregator
    P, k = nb_idx.shape
    valid = nb_idx >= 0
    owner, col = torch.nonzero(valid, as_tuple=True)              # (point, neighbour) pairs, row-major; one host round trip
    flat = nb_idx[owner, col]
    cnt = valid.sum(dim=1)
    off = torch.cumsum(cnt, 0) - cnt                             # a point's pairs are rows off[p] .. off[p] + cnt[p]
    # the reference trains stage 1 in fp32 and so does the default here (fp32 library GEMMs); `field.train_mlp_dtype`
    # (PointNeRFTrainer(mlp_dtype=...)) = "fp32_class" / torch.bfloat16 are the opt-ins that put the MLPs on the 16-bit matrix rate
    mlp_dtype = getattr(field, "train_mlp_dtype", None)
    precision = fused_pair_mlp_precision(field, mlp_dtype)
    lib_dtype = None if (mlp_dtype is None or isinstance(mlp_dtype, str) or mlp_dtype == torch.float32) else mlp_dtype     # operand type of the library layers (None: fp32)
    if precision == hr.PAIR_MLP_X2 and os.environ.get("NPCD_STAGE1_X2_HEADS"):
        # opt-in (measured SLOWER: 20.6-21.1 against 16.1 ms per step, docs/experiments.md R5.4): the point-level layers in the same fp32
        # class as one bf16 library GEMM over the three cross products of the split operands (_X2Linear) instead of fp32 library GEMMs
        lib_dtype = "x2"
    if precision is not None:
        # the four non-linear per-pair layers and the weighted mean run as ONE forward launch and one backward launch per layer on
        # the matrix cores (csrc/pairs_mlp.hip; fp32-class by default, bf16 operands as the opt-in); the network's last, linear
        # layer commutes with the mean (the normalised weights of a point sum to one and every compact point has a pair) and runs
        # on the points
        lf = agg.local_field
        G = hr.pair_mlp(kp_feat.reshape(-1, kp_feat.shape[-1]), [(lf[i].weight, lf[i].bias) for i in (0, 2, 4, 6)], nb_idx, pts,
                        kp_pos.detach().reshape(-1, 3), off, owner, flat, precision)
        if G.is_cuda and point_layers_fused(field, mlp_dtype, G.shape[0]):
            # the eight point-level layers in one launch of the same numerics class (csrc/points_x2.hip), activations saved for the
            # backward; NPCD_STAGE1_LIBRARY_HEADS=1 = the fp32 library layers below
            mods = ([lf[8]] + [m for m in field.shape_net if isinstance(m, torch.nn.Linear)]
                    + [m for m in field.channel_net if isinstance(m, torch.nn.Linear)])
            params = [t for m in mods for t in (m.weight, m.bias)]
            key = (str(G.device),) + tuple((t.data_ptr(), t._version) for t in params)
            if getattr(field, "_train_packx2_key", None) != key:          # (every optimizer step: packed on the device, one launch)
                field._train_packx2 = hr.points_x2_pack_device(mods, getattr(field, "_train_packx2", None))
                field._train_packx2_key = key
            shape_pre, chan_pre = _PointLayersX2.apply(G, field._train_packx2, *params)
            return F.softplus(shape_pre - 1.0), torch.sigmoid(chan_pre)
        if precision == hr.PAIR_MLP_BF16:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                agg_feat = lf[8](G).float()
        else:
            agg_feat = _mlp(torch.nn.Sequential(lf[8]), G, lib_dtype).float()
    else:
        # MLP input of every pair and its inverse-distance weight: one HIP kernel forward, one backward (csrc/pairs.hip)
        x0, w = hr.pair_input(kp_feat.reshape(-1, kp_feat.shape[-1]), flat, owner, pts, kp_pos.detach().reshape(-1, 3), agg.n_freqs)
        local = _mlp(agg.local_field, x0, lib_dtype).float()
        agg_feat = hr.pair_aggregate(local, w, off, cnt)         # weighted mean over each point's pairs (HIP fwd + bwd)
    chan_in = agg_feat
    if field.use_dir:
        from .field import encode_dir
        chan_in = torch.cat((agg_feat, encode_dir(point_dir, field.dir_freqs)), dim=-1)
    shape, chan = _mlp(field.shape_net, agg_feat, lib_dtype).float(), _mlp(field.channel_net, chan_in, lib_dtype).float()
    sigma = F.softplus(shape - 1.0)[:, 0]
    rgb = torch.sigmoid(chan)
    return sigma, rgb


# ------------------------------------------------------------------------------------------------- ray march (autograd)
def depths_from_points(pts, mask, o, d, ray_end):
    """pts [Nr, M, 3] (zeros at invalid slots), mask [Nr, M], o / d [Nr, 3], ray_end [Nr, 1] -> depth per slot [Nr, M]:
    invalid slots repeat the last valid depth before them, leading invalid slots take the ray end."""
    dep = torch.nanmean((pts - o[:, None, :]) / d[:, None, :], dim=-1)
    dep = torch.where(mask, dep, torch.full_like(dep, -math.inf))
    dep = torch.cummax(dep, dim=1).values
    return torch.where(dep == -math.inf, ray_end.expand_as(dep), dep)


def ray_march(sigma, depths, rgb, mask, white_back: bool):
    """sigma / depths / mask [Nr, M] dense, rgb [Nr, M, 3] -> opacity [Nr, 1], expected depth [Nr, 1], colour [Nr, 3]."""
    delta = torch.cat((depths[:, 1:] - depths[:, :-1], torch.zeros_like(depths[:, :1])), dim=1)
    alpha = 1.0 - torch.exp(-(sigma * delta))
    trans = torch.cumprod(torch.cat((torch.ones_like(alpha[:, :1]), 1.0 - alpha + 1e-10), dim=1), dim=1)[:, :-1]
    w = alpha * trans
    total = w.sum(dim=1, keepdim=True)
    depth = torch.nan_to_num((w * depths).sum(dim=1, keepdim=True) / total, float("inf"))
    if depth.numel() > 0:
        depth = torch.clamp(depth, depths.min(), depths.max())
    chan = ((w * mask)[..., None] * rgb).sum(dim=1)
    if white_back:
        chan = chan + 1.0 - total
    return total, depth, chan


# ------------------------------------------------------------------------------------------------- the training-mode forward
def render_train(renderer, kp_pos, kp_feat, extr, intr, resolution: int, sample: bool, rng: Optional[Dict] = None,
                 knn_mode: int = 0) -> AttrDict:
    """VolumeRenderer.forward for sample=True and / or jittered depths.  kp_pos [B,N,3], kp_feat [B,N,F] (may require grad),
    extr [B,T,4,4], intr [B,T,3,3] -> AttrDict(mask [B,T,n,1], depth [B,T,n,1], channels [B,T,n,3], ray_idx [B,T,n,1])."""
    rng = rng or {}
    field, agg = renderer.field, renderer.field.aggregator
    B, T = extr.shape[:2]
    dev = kp_pos.device
    R = resolution * resolution
    # rays and their cube limits (misses filled with the global limits of the generated set) come out of one kernel
    if renderer.ray_subsamples and sample:          # the same random rays for every (object, view) instance
        perm = rng["ray_perm"].to(dev) if "ray_perm" in rng else torch.randperm(R, device=dev)
        ray_ids = perm[:renderer.ray_subsamples].long()
        o, d, t0, t1 = hr.ray_gen(extr.flatten(0, 1), intr.flatten(0, 1), resolution, renderer.cube_scale, pixel_ids=ray_ids)
    else:
        ray_ids = torch.arange(R, device=dev)
        o, d, t0, t1 = hr.ray_gen(extr.flatten(0, 1), intr.flatten(0, 1), resolution, renderer.cube_scale)
    o, d = o.view(B, T, -1, 3), d.view(B, T, -1, 3)
    Rs = o.shape[2]
    t0, t1 = renderer.limits(t0, t1)                # fixed (near, far) when the renderer has ray_limits
    start, end = t0.view(B, T, Rs, 1), t1.view(B, T, Rs, 1)
    S = renderer.depth_resolution
    jitter = None
    if renderer.randomize_depth_samples:
        jitter = rng["jitter"].to(dev).reshape(B, T, Rs, S) if "jitter" in rng else torch.rand(B, T, Rs, S, device=dev)
    if renderer.disparity_space_sampling:
        # renderer.py:60-75: evenly spaced in 1 / depth with a jitter of its own (always), then the training jitter on top
        jd = rng["jitter_disp"].to(dev).reshape(B, T, Rs, S) if "jitter_disp" in rng else None
        dep = renderer.disparity_depths(start[..., 0], end[..., 0], jd)
        if jitter is not None:
            dep = dep + jitter * ((end - start) / (S - 1))
    else:
        dep = jittered_depths(start, end, S, jitter)
    x = o[..., None, :] + dep[..., None] * d[..., None, :]                       # [B,T,Rs,S,3]
    M = agg.max_shading_pts
    grid = agg.voxel_grid
    idx, loc, _, _ = grid.query_dense(agg.k, agg.scaled_r if knn_mode else agg.r, M, x=x.reshape(B, T * Rs, S, 3).contiguous(),
                                      mode=knn_mode, points=kp_pos.detach())
    idx = idx.view(B * T, Rs, M, agg.k).long()
    loc = loc.view(B * T, Rs, M, 3)
    slot_valid = idx[..., 0] >= 0                                                # lists are sorted: valid iff first entry valid
    if sample:
        ray_sel = agg.select_valid_rays(slot_valid, rng.get("valid_perm"))      # [B*T, Rs] bool, same count per instance
    else:
        ray_sel = torch.ones(B * T, Rs, dtype=torch.bool, device=dev)
    # compaction with ONE host round trip per list (boolean indexing would take one per indexed tensor)
    sel = torch.nonzero(ray_sel.view(-1)).squeeze(1)                             # selected rays, ascending
    n = sel.numel() // (B * T)
    idx_s, loc_s, valid_s = idx.view(-1, M, agg.k)[sel], loc.view(-1, M, 3)[sel], slot_valid.view(-1, M)[sel]
    rows = torch.nonzero(valid_s, as_tuple=True)                                 # valid (ray, slot) cells, row-major
    nb, pts = idx_s[rows], loc_s[rows]
    point_dir = d.reshape(-1, 3)[sel][rows[0]] if field.use_dir else None       # the ray direction of every compact point
    sigma_c, rgb_c = shade_autograd(field, nb, pts, kp_pos, kp_feat, point_dir)
    # ray march on the compact densities / colours, HIP forward and backward (no dense scatter, no per-slot depth tensors)
    per_ray = valid_s.sum(dim=1, dtype=torch.int32)
    base = torch.cumsum(per_ray, 0, dtype=torch.int32) - per_ray
    o_s, d_s, end_s = o.reshape(-1, 3)[sel], d.reshape(-1, 3)[sel], end.reshape(-1)[sel]
    total, cdepth, chan = hr.ray_march_train(sigma_c, rgb_c, valid_s, loc_s, base, o_s, d_s, end_s, renderer.white_back)
    out = AttrDict(mask=total.view(B, T, n, 1), depth=cdepth.view(B, T, n, 1), channels=chan.view(B, T, n, 3))
    if sample:
        out["ray_idx"] = ray_ids[None, :].expand(B * T, Rs).reshape(-1)[sel].view(B, T, n, 1)
    out["num_shading_points"] = int(nb.shape[0])
    out["num_pairs"] = (nb >= 0).sum()                                           # device scalar (no host round trip here)
    out["grid_level"] = "brute_force" if knn_mode else getattr(renderer.field.aggregator.voxel_grid, "grid_level", None)
    return out
