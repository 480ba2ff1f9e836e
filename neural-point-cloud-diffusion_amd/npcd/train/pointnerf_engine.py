"""Stage-1 training step (PointNeRF autodecoder): one iteration of the reference loop npcd/train/pointnerf_training.py:
129-147 -- zero_grad, PointNeRF.forward(sample_rays=True) on a batch of objects and views, PointNeRFLoss, backward, Adam
(lr 1e-3 over model.pointnerf.parameters(), :101-105; the point coordinates are frozen).

The forward renders a few hundred random rays per view (npcd.models.pointnerf.train_path): ray generation and the
neighbour queries (rendering and the TV loss) are HIP kernels, the differentiable shading runs on torch operators.
"""
import torch

from ..losses import PointNeRFLoss


class PointNeRFTrainer:
    def __init__(self, model, loss=None, lr: float = 1e-3):
        """model: NPCD (uses model.pointnerf); loss: a PointNeRFLoss (default weights of train_pointnerf.py:56-59)."""
        self.model = model
        self.loss = loss if loss is not None else PointNeRFLoss(model, 1, 1e-7, 3.5e-7)
        self.optimizer = torch.optim.Adam([p for p in model.pointnerf.parameters() if p.requires_grad], lr=lr)
        self.iteration = 0

    def step(self, sample, rng=None):
        """sample: dict(images [B,T,3,H,W], intrinsics [B,T,3,3], extrinsics [B,T,4,4], obj_idx [B]) on the GPU."""
        self.model.pointnerf.train()
        self.optimizer.zero_grad(set_to_none=True)
        pred, aux = self.model.pointnerf(sample["obj_idx"], sample["intrinsics"], sample["extrinsics"], sample_rays=True, rng=rng)
        loss, sub, _ = self.loss(sample=sample, pred=pred, aux=aux, iteration=self.iteration)
        loss.backward()
        self.optimizer.step()
        self.iteration += 1
        return loss.detach(), {k: v.detach() for k, v in sub.items()}
