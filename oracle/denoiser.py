"""Oracle (CPU, fp32) restatement of the NPCD transformer denoiser.  TEST INFRASTRUCTURE.

Functional style: every function takes a flat ``dict[str, Tensor]`` of parameters whose
keys are the reference's ``state_dict`` keys of ``NPCDTransformer``
(npcd/models/diffusion/denoisers/transformer.py:211-244), so the same weights can be fed to
the reference, to this oracle and to the HIP product.

Pinned by tests/golden/denoiser_*.npz and attention_*.npz (generated from the reference).
"""
import math
from typing import Dict, Tuple

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]


def timestep_embedding(t: torch.Tensor, dim: int, max_period: float = 10000.0) -> torch.Tensor:
    """transformer.py:33-48.  cos block first, then sin; zero pad when dim is odd."""
    half = dim // 2
    k = torch.arange(half, dtype=torch.float32)
    freqs = torch.exp(k * (-math.log(max_period)) / half).to(t.device)
    # the reference multiplies the (usually int64) timesteps with fp32 freqs -> fp32
    ang = t.reshape(-1, 1) * freqs.reshape(1, -1)
    out = torch.cat((ang.cos(), ang.sin()), dim=1)
    if dim % 2 == 1:
        out = torch.cat((out, out.new_zeros(out.shape[0], 1)), dim=1)
    return out


def attention_qkvpacked(qkv: torch.Tensor, heads: int) -> torch.Tensor:
    """transformer.py:68-84 (the einsum branch, which defines what flash_attn_func computes).

    qkv: [B, n, 3*W] where head h owns columns [3d*h, 3d*(h+1)) laid out q|k|v.
    returns [B, n, W] with head-major channels.
    """
    B, n, three_w = qkv.shape
    d = three_w // heads // 3
    x = qkv.reshape(B, n, heads, 3 * d)
    q, k, v = x[..., :d], x[..., d:2 * d], x[..., 2 * d:]
    s = 1.0 / math.sqrt(math.sqrt(d))
    # scores[b,h,i,j] = <q_i, k_j> / sqrt(d), scale split over both operands
    scores = torch.matmul((q * s).permute(0, 2, 1, 3), (k * s).permute(0, 2, 3, 1))
    prob = torch.softmax(scores, dim=-1)
    out = torch.matmul(prob, v.permute(0, 2, 1, 3))          # [B,H,n,d]
    return out.permute(0, 2, 1, 3).reshape(B, n, heads * d)


def _linear(p: Params, name: str, x: torch.Tensor) -> torch.Tensor:
    return F.linear(x, p[name + ".weight"], p[name + ".bias"])


def _ln(p: Params, name: str, x: torch.Tensor) -> torch.Tensor:
    w = p[name + ".weight"]
    return F.layer_norm(x, (w.shape[0],), w, p[name + ".bias"], 1e-5)


def _mlp(p: Params, name: str, x: torch.Tensor) -> torch.Tensor:
    """transformer.py:118-137: c_proj(GELU_erf(c_fc(x)))."""
    return _linear(p, name + ".c_proj", F.gelu(_linear(p, name + ".c_fc", x)))


def resblock(p: Params, prefix: str, x: torch.Tensor, heads: int) -> torch.Tensor:
    """transformer.py:169-172 with MultiheadAttention.forward (:110-115)."""
    a = _linear(p, prefix + ".attn.c_qkv", _ln(p, prefix + ".ln_1", x))
    a = _linear(p, prefix + ".attn.c_proj", attention_qkvpacked(a, heads))
    x = x + a
    x = x + _mlp(p, prefix + ".mlp", _ln(p, prefix + ".ln_2", x))
    return x


def num_layers(p: Params) -> int:
    n = 0
    while f"backbone.resblocks.{n}.ln_1.weight" in p:
        n += 1
    return n


def denoiser_forward(p: Params, coords: torch.Tensor, feats: torch.Tensor, t: torch.Tensor,
                     heads: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """transformer.py:246-274.  coords [B,3,N], feats [B,F,N], t [B] -> eps_coords, eps_feats."""
    width = p["ln_pre.weight"].shape[0]
    cdim = coords.shape[1]
    x = torch.cat((coords, feats), dim=1).permute(0, 2, 1)       # [B,N,C]
    temb = _mlp(p, "time_embed", timestep_embedding(t, width))   # [B,W]
    h = _linear(p, "input_proj", x)
    h = torch.cat((temb[:, None, :], h), dim=1)                   # time token first
    h = _ln(p, "ln_pre", h)
    for i in range(num_layers(p)):
        h = resblock(p, f"backbone.resblocks.{i}", h, heads)
    h = _ln(p, "ln_post", h)[:, 1:]
    out = _linear(p, "output_proj", h).permute(0, 2, 1)
    return out[:, :cdim], out[:, cdim:]


def init_params(coords_dim: int, feats_dim: int, width: int, layers: int, heads: int,
                seed: int = 0, init_scale: float = 0.25, output_std: float = 0.02) -> Params:
    """Synthetic weights following the reference init (transformer.py:27-30,190,229):
    block / time_embed Linears ~ N(0, (init_scale/sqrt(W))^2), zero bias; LayerNorm = (1, 0);
    input_proj = PyTorch default; output_proj ~ N(0, output_std^2) instead of the reference's
    zeros (transformer.py:242-244) so that gradients are not identically zero."""
    g = torch.Generator().manual_seed(seed)
    std = init_scale * math.sqrt(1.0 / width)
    C = coords_dim + feats_dim
    p: Params = {}

    def lin(name, out_f, in_f, s):
        p[name + ".weight"] = torch.randn(out_f, in_f, generator=g) * s
        p[name + ".bias"] = torch.zeros(out_f)

    def ln(name):
        p[name + ".weight"] = torch.ones(width)
        p[name + ".bias"] = torch.zeros(width)

    lin("time_embed.c_fc", 4 * width, width, std)
    lin("time_embed.c_proj", width, 4 * width, std)
    ln("ln_pre")
    for i in range(layers):
        pre = f"backbone.resblocks.{i}"
        lin(pre + ".attn.c_qkv", 3 * width, width, std)
        lin(pre + ".attn.c_proj", width, width, std)
        ln(pre + ".ln_1")
        ln(pre + ".ln_2")
        lin(pre + ".mlp.c_fc", 4 * width, width, std)
        lin(pre + ".mlp.c_proj", width, 4 * width, std)
    ln("ln_post")
    bound = 1.0 / math.sqrt(C)
    p["input_proj.weight"] = (torch.rand(width, C, generator=g) * 2 - 1) * bound
    p["input_proj.bias"] = (torch.rand(width, generator=g) * 2 - 1) * bound
    lin("output_proj", C, width, output_std)
    return p
