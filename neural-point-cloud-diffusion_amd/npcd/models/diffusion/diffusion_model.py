"""DiffusionModel = DDPM process + transformer denoiser + the two data normalisers
(reference npcd/models/diffusion/diffusion_model.py)."""
import contextlib

import torch
import torch.nn as nn

from .gaussian_diffusion import GaussianDiffusion
from .transformer import NPCDTransformer


class _Normalization(nn.Module):
    """Shared state of the two normalisers: buffers min/max/shift/scale with the reference's shapes
    (reference :16-19, :53-56).  training: data -> model space; eval: model -> data space."""

    def __init__(self, dim, scale_per_axis=False, clip_per_axis=False):
        super().__init__()
        self.dim, self.scale_per_axis, self.clip_per_axis = dim, scale_per_axis, clip_per_axis
        self.register_buffer("min", torch.zeros(dim if clip_per_axis else 1))
        self.register_buffer("max", torch.zeros(dim if clip_per_axis else 1))
        self.register_buffer("shift", torch.zeros(dim))
        self.register_buffer("scale", torch.ones(dim if scale_per_axis else 1))

    def _flat(self, data):
        data = torch.as_tensor(data, device=self.shift.device).detach().float()
        assert data.shape[0] == self.dim
        return data.reshape(self.dim, -1)

    def _set_clip_range(self, data):
        z = (data - self.shift[:, None]) / self.scale[:, None]
        if self.clip_per_axis:
            self.min.copy_(z.min(dim=1).values)
            self.max.copy_(z.max(dim=1).values)
        else:
            self.min.fill_(z.min())
            self.max.fill_(z.max())

    def forward(self, x):
        shift, scale = self.shift[None, :, None], self.scale[None, :, None]
        return (x - shift) / scale if self.training else x * scale + shift


class UnitGaussianNormalization(_Normalization):
    @torch.no_grad()
    def set_from_all_data(self, data):
        d = self._flat(data)
        self.shift.copy_(d.mean(dim=1))
        self.scale.copy_(d.std(dim=1) if self.scale_per_axis else d.std().reshape(1))
        self._set_clip_range(d)


class MinusOneToOneNormalization(_Normalization):
    @torch.no_grad()
    def set_from_all_data(self, data):
        d = self._flat(data)
        lo, hi = d.min(dim=1).values, d.max(dim=1).values
        self.shift.copy_((lo + hi) / 2.0)
        half = (hi - lo) / 2.0
        self.scale.copy_(half if self.scale_per_axis else half.max().reshape(1))
        self._set_clip_range(d)


class DiffusionModel(nn.Module):
    def __init__(self, coords_dim, feats_dim, num_points, width, layers, heads, use_flash_attn):
        super().__init__()
        self.coords_dim, self.feats_dim, self.num_points = coords_dim, feats_dim, num_points
        self.diffusion_process = GaussianDiffusion()
        self.denoiser = NPCDTransformer(coords_dim=coords_dim, feats_dim=feats_dim, width=width, layers=layers,
                                        heads=heads, use_flash_attn=use_flash_attn)
        self.coords_normalization = UnitGaussianNormalization(dim=coords_dim)
        self.feats_normalization = MinusOneToOneNormalization(dim=feats_dim)

    def compute_loss(self, coords, feats, t=None, coords_noise=None, feats_noise=None, want_pointwise=True):
        """Reference :99-106.  t / noise may be injected (parity tests); by default they are drawn on
        the device like the reference does."""
        coords = self.coords_normalization(coords)
        feats = self.feats_normalization(feats)
        if t is None:
            t = torch.randint(0, self.diffusion_process.num_timesteps, size=(coords.shape[0],), device=coords.device)
        return self.diffusion_process.p_losses(self.denoiser, coords, feats, t, coords_noise, feats_noise, want_pointwise=want_pointwise)

    @torch.no_grad()
    def generate(self, num, batch_size=8, progress=True, dtype=None, use_graph=False):
        """Reference :108-133.  Two options the reference does not have: `dtype` (e.g. torch.bfloat16) runs the denoiser under
        autocast on the MFMA attention kernels -- 6x faster per reverse step than the reference's fp32 sampling, eps within the
        training-time bf16 tolerance; `use_graph` replays each reverse step from a captured HIP graph."""
        assert not self.training, "Model must be in eval mode for generation"
        device = next(self.parameters()).device
        sizes = [batch_size] * (num // batch_size) + ([num % batch_size] if num % batch_size else [])
        # dtype = "fp32_class": the reference's fp32 sampling with the backbone's Linear layers as split-operand bf16 GEMMs (two bf16 halves
        # per operand, three cross products, fp32 accumulation: 3e-6 relative per product; everything else fp32) -- fused.backbone_forward_x2
        fp32_class = isinstance(dtype, str)
        if fp32_class and dtype != "fp32_class":
            raise ValueError("generate(dtype=...): a torch dtype for autocast, 'fp32_class', or None")
        backbone = getattr(self.denoiser, "backbone", None)
        ctx = torch.autocast(device.type, dtype=dtype) if (dtype is not None and not fp32_class) else contextlib.nullcontext()
        prev_mode = getattr(backbone, "fp32_class", False)
        if backbone is not None:
            backbone.fp32_class = fp32_class
        try:
            return self._generate(sizes, device, ctx, progress, use_graph)
        finally:
            if backbone is not None:
                backbone.fp32_class = prev_mode

    def _generate(self, sizes, device, ctx, progress, use_graph):
        coords_out, feats_out = [], []
        for bs in sizes:
            c = torch.randn(bs, self.coords_dim, self.num_points, device=device)
            f = torch.randn(bs, self.feats_dim, self.num_points, device=device)
            with ctx:
                c, f = self.diffusion_process.p_sample_loop(
                    self.denoiser, c, f,
                    coords_clip_range=(self.coords_normalization.min, self.coords_normalization.max),
                    feats_clip_range=(self.feats_normalization.min, self.feats_normalization.max), progress=progress,
                    use_graph=use_graph)
            coords_out += list(self.coords_normalization(c).unbind())
            feats_out += list(self.feats_normalization(f).unbind())
        return coords_out, feats_out
