"""PointNeRF auto-decoder renderer (reference npcd/models/pointnerf/pointnerf.py) on the HIP kernels."""
import torch
import torch.nn as nn

from ...hip.render import HipVoxelGrid
from ...utils import AttrDict
from .embeddings import Embedding, VariationalEmbedding
from .field import Field
from .renderer import VolumeRenderer


def _get_pointnerf_options() -> AttrDict:
    """The hard-coded option tree of the reference (pointnerf.py:134-194)."""
    return AttrDict(
        model=dict(
            kp=dict(num=512, feat_dim=32),
            embedding=dict(type="VariationalEmbedding", kwargs=dict(gpu=True)),
            voxel_grid=dict(voxel_size=(0.04, 0.04, 0.04), voxel_scale=(2, 2, 2), kernel_size=(3, 3, 3),
                            max_points_per_voxel=4, max_occ_voxels_per_example=5000,
                            ranges=(-1.0, -1.0, -1.0, 1.0, 1.0, 1.0)),
            field=dict(network="MLP", nerf=True,
                       kwargs=dict(feat_freqs=0, dir_freqs=8, channel_layers=[256, 256, 256, 256], shape_layers=[256],
                                   activation="LeakyReLU", layer_norm=False, use_dir=False),
                       aggregator=dict(network="MLP",
                                       kwargs=dict(k=8, r=2, max_shading_pts=50, ray_subsamples=128, n_freqs=10, freq_mult=1,
                                                   out_dim=256, layers=[256, 256, 256, 256], activation="LeakyReLU",
                                                   layer_norm=False))),
            renderer=dict(network="VolumeRenderer",
                          kwargs=dict(depth_resolution=128, disparity_space_sampling=False, white_back=True, cube_scale=1.0,
                                      ray_subsamples=112, ray_limits=None))),
        sizes=dict(default_resolution=128))


class PointNeRF(nn.Module):
    def __init__(self, n_obj: int, feats_dim: int, num_points: int, use_view_dir: bool):
        super().__init__()
        opt = _get_pointnerf_options()
        opt.model.field.kwargs.use_dir = use_view_dir
        opt.model.kp.feat_dim = feats_dim
        opt.model.kp.num = num_points
        self.opt = opt
        self.voxel_grid = HipVoxelGrid(**opt.model.voxel_grid)
        emb_cls = {"VariationalEmbedding": VariationalEmbedding, "Embedding": Embedding}[opt.model.embedding.type]
        self.feats = emb_cls(num_points, feats_dim, n_obj, **opt.model.embedding.kwargs)
        self.coords = Embedding(num_points, 3, n_obj, **opt.model.embedding.kwargs)
        self.coords.freeze(True)
        self.field = Field(feats_dim, self.voxel_grid, opt.model.field.aggregator, **opt.model.field.kwargs,
                           nerf=opt.model.field.nerf)
        self.renderer = VolumeRenderer(self.field, **opt.model.renderer.kwargs)

    def train(self, mode=True):
        super().train(mode)
        self.renderer.randomize_depth_samples = mode
        return self

    @torch.no_grad()
    def set_all_coords(self, coords):
        self.coords.get_emb().weight.copy_(coords.reshape(coords.shape[0], -1))

    def get_all_coords(self):
        w = self.coords.get_emb().weight
        return w.reshape(w.shape[0], self.opt.model.kp.num, 3)

    def get_all_feats(self):
        w = self.feats.get_emb().weight
        F_ = self.opt.model.kp.feat_dim
        if self.opt.model.embedding.type == "VariationalEmbedding":
            return w.reshape(w.shape[0], self.opt.model.kp.num, 2 * F_)[:, :, :F_]
        return w.reshape(w.shape[0], self.opt.model.kp.num, F_)

    def _set_pointset(self, coords):
        # every cloud holds all of its kp.num points (the reference passes that count per example, pointnerf.py:67): the HIP grid
        # takes "no counts" as exactly that, and can then recognise an unchanged cloud from one view to the next
        if coords.shape[1] != self.opt.model.kp.num:
            counts = torch.full((coords.shape[0],), self.opt.model.kp.num, device=coords.device, dtype=torch.int)
            self.voxel_grid.set_pointset(coords.detach(), counts)
        else:
            self.voxel_grid.set_pointset(coords.detach())

    def forward(self, obj_idx, intrinsics, extrinsics, sample_rays: bool, rng=None):
        """reference pointnerf.py:56-105 -> (pred AttrDict, aux dict).  `rng` (tests) replays given random draws:
        eps (feature reparametrisation), ray_perm, jitter, valid_perm (npcd.models.pointnerf.train_path)."""
        rng = rng or {}
        feats = self.feats(idx=obj_idx, eps=rng["eps"]) if "eps" in rng else self.feats(idx=obj_idx)
        coords = self.coords(idx=obj_idx)
        self._set_pointset(coords)
        if hasattr(self.feats, "get_mean_log_var_std"):
            mean, log_var, std = self.feats.get_mean_log_var_std(idx=obj_idx)
            aux = {"coords": coords, "feats": mean, "feats_mean": mean, "feats_log_var": log_var, "feats_std": std}
        else:
            aux = {"coords": coords, "feats": feats}
        pred = self.renderer(coords, feats, extrinsics, intrinsics, resolution=self.opt.sizes.default_resolution,
                             sample=sample_rays, return_channels=True, rng=rng)
        return pred, aux

    def render(self, coords, feats, extrinsics, intrinsics, resolution=128, max_shading_points=None, sample_rays=False, mlp_dtype=None):
        """reference pointnerf.py:107-131.  mlp_dtype (not in the reference): None = the fused fp16-operand shading kernels with their
        range guard (out["shading_status"]); torch.float32 = the reference's numerics class for the field MLPs (the evaluation
        scripts run them in fp32): fp32-class matrix-core per-pair layers + fp32 heads, VolumeRenderer.shade_dtype."""
        agg = self.field.aggregator
        prev, prev_dtype = agg.max_shading_pts, self.renderer.shade_dtype
        if max_shading_points is not None:
            agg.max_shading_pts = max_shading_points
        if mlp_dtype is not None:
            if mlp_dtype != torch.float32 or not self.field.fp32_class_ok():
                raise ValueError("render(mlp_dtype=...): torch.float32 on the published field architecture, or None")
            self.renderer.shade_dtype = mlp_dtype
        try:
            self._set_pointset(coords)
            return self.renderer(coords, feats, extrinsics, intrinsics, resolution, sample_rays)
        finally:
            agg.max_shading_pts, self.renderer.shade_dtype = prev, prev_dtype
