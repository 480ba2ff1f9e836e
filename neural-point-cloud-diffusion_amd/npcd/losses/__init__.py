"""Stage-1 (PointNeRF autodecoder) losses with the reference's module surface (npcd/losses/*): every loss is called as
`loss(sample, pred, aux, iteration) -> (total, sub_losses, pointwise_losses)`.

    ImageReconstructionLoss   mean squared error between rendered channels and the ground-truth pixels of the rays that
                              were rendered (image_reconstruction_loss.py:28-40, utils/util.py:188-196)
    NeuralPointCloudKLLoss    KL(N(mean, exp(log_var)) || N(0, 1)) of the variational feature embedding, per point
                              (neural_point_cloud_kl_loss.py:29-44)
    NeuralPointCloudTVLoss    inverse-distance weighted L1 variation of the features over each point's k nearest
                              neighbours (neural_point_cloud_tv_loss.py:29-83); the neighbour lists come from the HIP
                              voxel-grid query through `model.pointnerf.field.aggregator.query_keypoints`
    PointNeRFLoss             their weighted sum (pointnerf_loss.py:10-52; train_pointnerf.py:56-59 uses 1 / 1e-7 / 3.5e-7)
"""
import torch
import torch.nn as nn


def subsample_gt(gt_map: torch.Tensor, ray_idx):
    """images [..., C, H, W] -> per-ray targets [..., R, C], gathered at ray_idx [..., n, 1] when given."""
    samples = gt_map.flatten(-2, -1).transpose(-1, -2)
    if ray_idx is not None:
        samples = samples.expand(*ray_idx.shape[:-2], *samples.shape[-2:])
        samples = samples.gather(dim=-2, index=ray_idx.expand(*ray_idx.shape[:-1], samples.shape[-1]))
    return samples


class _Loss(nn.Module):
    def __init__(self, model=None, weight=1, verbose=False):
        super().__init__()
        self.model, self.weight, self.verbose = [model], weight, verbose        # the model is not a sub-module of its loss

    @property
    def name(self):
        return type(self).__name__


class ImageReconstructionLoss(_Loss):
    def forward(self, sample, pred, aux, iteration):
        gt = subsample_gt(sample["images"], pred.get("ray_idx", None))
        loss = ((pred["channels"].contiguous() - gt) ** 2).mean() * self.weight
        return loss, {"pointnerf_reconstruction": loss}, {}


class NeuralPointCloudKLLoss(_Loss):
    def forward(self, sample, pred, aux, iteration):
        mean, log_var = aux["feats_mean"], aux["feats_log_var"]
        kld = -0.5 * torch.sum(1 + log_var - mean.pow(2) - log_var.exp(), dim=-1) * self.weight      # [B, N]
        total = kld.mean()
        return total, {"00_neural_point_cloud_kl": total}, {"00_neural_point_cloud_kl": kld}


class NeuralPointCloudTVLoss(_Loss):
    def forward(self, sample, pred, aux, iteration):
        feats, coords = aux["feats"], aux["coords"].detach()
        B, N = coords.shape[:2]
        dev = coords.device
        agg = self.model[0].pointnerf.field.aggregator
        # each point queries its own neighbourhood: one "ray" per point with a single sample
        nb, _, found = agg.query_keypoints(coords.view(B, 1, N, 1, 3), coords)
        k = nb.shape[-1]
        found = found[..., 0, :].reshape(B, N, 1)
        own = torch.arange(N, device=dev)[None, :, None] + (torch.arange(B, device=dev) * N)[:, None, None]
        full = torch.full((B, N, k), -1, dtype=torch.long, device=dev)
        full[..., :1] = own                              # a point the grid lost keeps itself as its only neighbour
        full.masked_scatter_(found.expand(B, N, k), nb)
        identity = full == own
        enough = (full >= 0).sum(dim=-1, keepdim=True) > 1
        full = torch.where(identity & enough, torch.full_like(full, -1), full).view(B * N, k)
        valid = full >= 0
        owner = torch.arange(B * N, device=dev)[:, None].expand(B * N, k)[valid]
        cflat, fflat = coords.reshape(B * N, 3), feats.reshape(B * N, -1)
        w = 1.0 / (torch.linalg.norm(cflat[full[valid]] - cflat[owner], dim=-1) + 1e-5)
        dist = torch.linalg.norm(fflat[full[valid]] - fflat[owner], ord=1, dim=-1)
        tv = torch.zeros(B * N, device=dev, dtype=dist.dtype).index_add_(0, owner, w * dist).view(B, N) * self.weight
        total = tv.mean()
        return total, {"00_neural_point_cloud_tv": total}, {"00_neural_point_cloud_tv": tv}


class PointNeRFLoss(nn.Module):
    def __init__(self, model, image_reconstruction_loss_weight=1, neural_point_cloud_kl_loss_weight=1,
                 neural_point_cloud_tv_loss_weight=1, verbose=False):
        super().__init__()
        self.image_reconstruction_loss = ImageReconstructionLoss(model, image_reconstruction_loss_weight, verbose)
        self.neural_point_cloud_kl_loss = NeuralPointCloudKLLoss(model, neural_point_cloud_kl_loss_weight, verbose)
        self.neural_point_cloud_tv_loss = NeuralPointCloudTVLoss(model, neural_point_cloud_tv_loss_weight, verbose)

    @property
    def name(self):
        return type(self).__name__

    def forward(self, sample, pred, aux, iteration):
        rec, _, _ = self.image_reconstruction_loss(sample, pred, aux, iteration)
        kl, _, _ = self.neural_point_cloud_kl_loss(sample, pred, aux, iteration)
        tv, _, _ = self.neural_point_cloud_tv_loss(sample, pred, aux, iteration)
        sub = {"00_image_reconstruction_loss": rec, "01_neural_point_cloud_kl": kl, "02_neural_point_cloud_tv": tv}
        return rec + kl + tv, sub, {}
