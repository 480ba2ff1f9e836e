"""Transformer denoiser over neural-point-cloud latents (MI355X build).

Module tree and parameter names follow the reference's NPCDTransformer
(npcd/models/diffusion/denoisers/transformer.py:211-274) one to one, so reference checkpoints
load unchanged.  The attention operator is the gfx950 HIP kernel
(npcd.hip.attention); the Linear GEMMs run on hipBLASLt through torch.
"""
import math
import os
from typing import Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from ...hip.attention import attention_qkvpacked

_NO_FAST_GLUE = bool(os.environ.get("NPCD_NO_FAST_GLUE"))       # A/B switch: the reference's expressions around the backbone as they are


def timestep_embedding(t: torch.Tensor, dim: int, max_period: float = 10000.0) -> torch.Tensor:
    """[cos(t f_j) | sin(t f_j)], f_j = max_period^(-j/half)   (reference transformer.py:33-48)."""
    half = dim // 2
    j = torch.arange(half, dtype=torch.float32, device=t.device)
    freqs = torch.exp(j * (-math.log(max_period)) / half)
    ang = t.reshape(-1, 1) * freqs.reshape(1, -1)
    emb = torch.cat((torch.cos(ang), torch.sin(ang)), dim=1)
    if dim & 1:
        emb = F.pad(emb, (0, 1))
    return emb


def _normal_init(linear: nn.Linear, std: float):
    nn.init.normal_(linear.weight, std=std)
    nn.init.zeros_(linear.bias)


class QKVMultiheadAttention(nn.Module):
    """softmax(q k^T / sqrt(d)) v on the packed c_qkv output (reference :51-84).  `use_flash_attn`
    is kept for constructor compatibility; both settings run the HIP kernel."""

    def __init__(self, *, heads: int, dropout: float = 0.0, use_flash_attn: bool = True):
        super().__init__()
        if dropout:
            raise NotImplementedError("attention dropout is 0 everywhere in the reference")
        self.heads = heads
        self.use_flash_attn = use_flash_attn

    def forward(self, qkv: torch.Tensor) -> torch.Tensor:
        return attention_qkvpacked(qkv, self.heads)


class MultiheadAttention(nn.Module):
    def __init__(self, *, width: int, heads: int, init_scale: float = 1.0, use_flash_attn: bool = True):
        super().__init__()
        self.width, self.heads = width, heads
        self.c_qkv = nn.Linear(width, 3 * width)
        self.c_proj = nn.Linear(width, width)
        self.attention = QKVMultiheadAttention(heads=heads, use_flash_attn=use_flash_attn)
        _normal_init(self.c_qkv, init_scale)
        _normal_init(self.c_proj, init_scale)

    def forward(self, x):
        return self.c_proj(self.attention(self.c_qkv(x)))


class MLP(nn.Module):
    def __init__(self, *, width: int, init_scale: float = 1.0):
        super().__init__()
        self.width = width
        self.c_fc = nn.Linear(width, 4 * width)
        self.c_proj = nn.Linear(4 * width, width)
        _normal_init(self.c_fc, init_scale)
        _normal_init(self.c_proj, init_scale)

    def forward(self, x):
        return self.c_proj(F.gelu(self.c_fc(x)))       # exact-erf GELU (reference :131)


class ResidualAttentionBlock(nn.Module):
    """pre-LN block: x += attn(ln_1 x); x += mlp(ln_2 x)   (reference :140-172)."""

    def __init__(self, *, width: int, heads: int, init_scale: float = 1.0, use_flash_attn: bool = True):
        super().__init__()
        self.attn = MultiheadAttention(width=width, heads=heads, init_scale=init_scale, use_flash_attn=use_flash_attn)
        self.ln_1 = nn.LayerNorm(width)
        self.ln_2 = nn.LayerNorm(width)
        self.mlp = MLP(width=width, init_scale=init_scale)

    def forward(self, x):
        x = x + self.attn(self.ln_1(x))
        return x + self.mlp(self.ln_2(x))


class Transformer(nn.Module):
    def __init__(self, *, width: int, layers: int, heads: int, init_scale: float = 0.25, use_flash_attn: bool = True):
        super().__init__()
        self.width, self.layers = width, layers
        std = init_scale * math.sqrt(1.0 / width)
        self.resblocks = nn.ModuleList(
            ResidualAttentionBlock(width=width, heads=heads, init_scale=std, use_flash_attn=use_flash_attn)
            for _ in range(layers))
        self.fused_engine = None        # set by npcd.train.DiffusionTrainer (explicit fwd/bwd over flat buffers)
        self._infer_weights = None      # bf16 weight copies of the forward-only path (sampler), built on first use
        self._infer_weights_x2 = None   # split-operand weight copies of the fp32-class forward-only path
        self.fp32_class = False         # set by DiffusionModel.generate(dtype="fp32_class"): fp32 forward with split-operand GEMMs

    def forward(self, x):
        eng = self.fused_engine
        if x.is_cuda and x.dtype == torch.float32 and torch.is_autocast_enabled():
            act = torch.get_autocast_dtype("cuda")
            if torch.is_grad_enabled():
                if eng is not None and eng.dtype == act:       # bf16, or f16 with the trainer's loss scaling
                    return eng(x)
            elif eng is not None and eng.dtype == act:
                # sampling / evaluation under the trainer's autocast type: fused forward-only kernels on its 16-bit shadow
                from . import fused
                if eng.wait_range is not None:
                    eng.wait_range()
                eng.sync_shadow()
                return fused.backbone_forward(x, eng.blocks, eng.heads, eng.dtype)
            elif eng is None and act == torch.bfloat16:
                # sampling / evaluation of a model without a trainer under bf16 autocast (fused.backbone_forward on cached weights)
                from . import fused
                if self._infer_weights is None:
                    self._infer_weights = fused.InferenceWeights(self)
                return fused.backbone_forward(x, self._infer_weights.current(), self._infer_weights.heads)
        if eng is not None and eng.wait_range is not None:
            eng.wait_range()            # module path while a trainer's parameter gathers may still be in flight (engine.py)
        if (self.fp32_class and x.is_cuda and x.dtype == torch.float32 and not torch.is_autocast_enabled() and not torch.is_grad_enabled()
                and self.width // self.resblocks[0].attn.heads == 64 and self.width % 8 == 0):
            # sampling / evaluation in the reference's fp32 class on the bf16 matrix rate (fused.backbone_forward_x2)
            from . import fused
            if self._infer_weights_x2 is None:
                self._infer_weights_x2 = fused.InferenceWeightsX2(self)
            return fused.backbone_forward_x2(x, self._infer_weights_x2.current(), self._infer_weights_x2.heads)
        for blk in self.resblocks:
            x = blk(x)
        return x


class _TimeTokenCat(torch.autograd.Function):
    """h = cat(temb[:, None], tokens) for bf16 tokens [B, N, W] = input_proj(x) on the GPU, carrying the BIAS GRADIENT of the
    input projection: the Linear is called with its bias detached (same forward values: the library GEMM adds the bias to the fp32
    accumulator), and this function returns the bias gradient from the fused backbone's column-sum kernel over the whole contiguous
    dh minus the B time-token rows (25 us) -- torch reduces the [B, N, W] bf16 slice with a 360 us column reduction
    (reference expression: transformer.py:246-248)."""

    @staticmethod
    def forward(ctx, temb, tokens, bias):
        B, N, W = tokens.shape
        h = torch.empty((B, N + 1, W), dtype=tokens.dtype, device=tokens.device)
        h[:, 0] = temb
        h[:, 1:] = tokens
        ctx.bias_dtype = bias.dtype
        return h

    @staticmethod
    def backward(ctx, dh):
        from ...hip import elementwise as ew
        dh = dh.contiguous()
        B, n, W = dh.shape
        db = None
        if ctx.needs_input_grad[2]:
            db = torch.empty(W, dtype=torch.float32, device=dh.device)
            ew.colsum_bf16(dh.view(B * n, W), db)
            db -= dh[:, 0].float().sum(dim=0)
            db = db.to(ctx.bias_dtype)
        return dh[:, 0], dh[:, 1:], db


class _LayerNormToBF16(torch.autograd.Function):
    """ln_post on the fused backbone's LayerNorm kernels (csrc/elementwise.hip): fp32 statistics and normalisation like the
    reference's fp32 nn.LayerNorm under autocast (transformer.py:249), the result rounded to the autocast type (bf16 / f16) once -- exactly what the
    following Linear's autocast does to the fp32 output -- and a backward of 92 us instead of torch's 190 us pair of kernels."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        from ...hip import elementwise as ew
        B, n, W = x.shape
        x2 = x.reshape(B * n, W).contiguous()
        _, y, mean, rstd = ew.add_ln_fwd(x2, None, gamma, beta, eps=eps, dtype=torch.get_autocast_dtype("cuda"))
        ctx.save_for_backward(x2, mean, rstd, gamma)
        return y.view(B, n, W)

    @staticmethod
    def backward(ctx, dy):
        from ...hip import elementwise as ew
        x2, mean, rstd, gamma = ctx.saved_tensors
        B, n, W = dy.shape
        dgamma, dbeta = torch.empty(W, dtype=torch.float32, device=dy.device), torch.empty(W, dtype=torch.float32, device=dy.device)
        dx, _ = ew.ln_bwd(dy.reshape(B * n, W).contiguous(), x2, mean, rstd, gamma, None, dgamma, dbeta, want_bf16=False)
        return dx.view(B, n, W), dgamma.to(gamma.dtype), dbeta.to(gamma.dtype), None


class NPCDTransformer(nn.Module):
    """eps-prediction network: (coords [B,3,N], feats [B,F,N], t [B]) -> (eps_coords, eps_feats)."""

    def __init__(self, *, coords_dim: int, feats_dim: int, width: int = 512, layers: int = 12, heads: int = 8,
                 init_scale: float = 0.25, use_flash_attn: bool = True):
        super().__init__()
        self.coords_dim, self.feats_dim = coords_dim, feats_dim
        self.input_channels = self.output_channels = coords_dim + feats_dim
        self.time_embed = MLP(width=width, init_scale=init_scale * math.sqrt(1.0 / width))
        self.ln_pre = nn.LayerNorm(width)
        self.backbone = Transformer(width=width, layers=layers, heads=heads, init_scale=init_scale,
                                    use_flash_attn=use_flash_attn)
        self.ln_post = nn.LayerNorm(width)
        self.input_proj = nn.Linear(self.input_channels, width)
        self.output_proj = nn.Linear(width, self.output_channels)
        nn.init.zeros_(self.output_proj.weight)      # reference :242-244
        nn.init.zeros_(self.output_proj.bias)

    def forward(self, coords: torch.Tensor, feats: torch.Tensor, t: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        x = torch.cat((coords, feats), dim=1).transpose(1, 2)
        temb = self.time_embed(timestep_embedding(t, self.backbone.width))               # [B,W]
        # (the column-sum kernel behind _TimeTokenCat's backward takes bf16 rows of a multiple of 8 columns)
        if (x.is_cuda and torch.is_grad_enabled() and torch.is_autocast_enabled()
                and torch.get_autocast_dtype("cuda") in (torch.bfloat16, torch.float16) and self.backbone.width % 8 == 0 and not _NO_FAST_GLUE):
            tokens = F.linear(x, self.input_proj.weight, self.input_proj.bias.detach())  # [B,N,W] bf16; the bias gradient comes from _TimeTokenCat
            h = _TimeTokenCat.apply(temb.to(tokens.dtype), tokens, self.input_proj.bias)
        else:
            tokens = self.input_proj(x)                                                  # [B,N,W]
            h = torch.cat((temb.unsqueeze(1).to(tokens.dtype), tokens), dim=1)           # time token first
        h = self.backbone(self.ln_pre(h))
        W = h.shape[-1]
        if (h.is_cuda and h.dtype == torch.float32 and torch.is_grad_enabled() and torch.is_autocast_enabled() and W % 4 == 0 and W <= 2048
                and torch.get_autocast_dtype("cuda") in (torch.bfloat16, torch.float16) and self.ln_post.weight.dtype == torch.float32
                and not _NO_FAST_GLUE):
            h = _LayerNormToBF16.apply(h, self.ln_post.weight, self.ln_post.bias, self.ln_post.eps)
        else:
            h = self.ln_post(h)
        eps = self.output_proj(h[:, 1:]).transpose(1, 2)                                 # [B,C,N]
        return eps[:, :self.coords_dim], eps[:, self.coords_dim:]
