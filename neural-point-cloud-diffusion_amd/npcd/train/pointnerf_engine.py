"""Stage-1 training step (PointNeRF autodecoder): one iteration of the reference loop npcd/train/pointnerf_training.py:
129-147 -- zero_grad, PointNeRF.forward(sample_rays=True) on a batch of objects and views, PointNeRFLoss, backward, Adam
(lr 1e-3 over model.pointnerf.parameters(), :101-105; the point coordinates are frozen).

The forward renders a few hundred random rays per view (npcd.models.pointnerf.train_path): ray generation and the
neighbour queries (rendering and the TV loss) are HIP kernels, the differentiable shading runs on torch operators.
"""
import torch

from ..losses import PointNeRFLoss


class PointNeRFTrainer:
    def __init__(self, model, loss=None, lr: float = 1e-3, mlp_dtype=None):
        """model: NPCD (uses model.pointnerf); loss: a PointNeRFLoss (default weights of train_pointnerf.py:56-59).
        mlp_dtype: None (or torch.float32, or "library") = the reference's numerics (train_pointnerf.py runs without autocast): TRUE
        fp32 operands and accumulation, every Linear layer on fp32 library GEMMs;
        "fp32_class" = explicit opt-in, faster: every operand as two bf16 halves on the matrix cores (~1e-5 relative per product
        where fp32 has ~6e-8: csrc/pairs_mlp.hip precision 1, csrc/points_x2.hip) -- the per-pair layers forward + backward and,
        from 4,096 shading points on, the point-level layers' forward;
        torch.bfloat16 = bf16 operands throughout (opt-in, narrower still)."""
        self.model = model
        model.pointnerf.field.train_mlp_dtype = mlp_dtype
        self.loss = loss if loss is not None else PointNeRFLoss(model, 1, 1e-7, 3.5e-7)
        # every parameter is handed to Adam like the reference does (:102): frozen ones never get a gradient or a state, but
        # keep their index, so optimizer state dictionaries are interchangeable with the reference's
        self.optimizer = torch.optim.Adam(model.pointnerf.parameters(), lr=lr)
        self.scheduler = torch.optim.lr_scheduler.StepLR(self.optimizer, step_size=1, gamma=1.0)       # constant (:105)
        self.iteration = 0

    def describe(self) -> str:
        """What runs where in this trainer's step (bench.py prints it next to the timing).  Derived from the predicates the forward
        itself uses (train_path.fused_pair_mlp_precision / point_layers_fused), so the text cannot drift from the code path."""
        from ..hip import render as hr
        from ..models.pointnerf.train_path import fused_pair_mlp_precision, point_layers_fused
        field = self.model.pointnerf.field
        dt = getattr(field, "train_mlp_dtype", None)
        base = ("HIP kernels with hand-written backward: ray generation, both neighbour queries, ray march; ")
        prec = fused_pair_mlp_precision(field, dt)
        if prec == hr.PAIR_MLP_BF16:
            return base + ("per-pair aggregator MLP (gather, positional encoding, 4 layers, weighted mean) forward + backward on the matrix "
                           "cores with bf16 operands (csrc/pairs_mlp.hip); point-level layers and losses: torch under bf16 autocast")
        if prec == hr.PAIR_MLP_X2:
            points = ("point-level layers: forward as ONE fp32-class fused launch (csrc/points_x2.hip, activations saved in fp32) from 4,096 "
                      "shading points on, backward on fp32 library GEMMs (row-split weight gradients)" if point_layers_fused(field, dt)
                      else "point-level layers: fp32 library GEMMs (row-split weight gradients)")
            return base + ("per-pair aggregator MLP (gather, positional encoding, 4 layers, weighted mean) forward + backward on the matrix "
                           "cores in the fp32-class mode (two bf16 halves per operand, three products, fp32 accumulation: csrc/pairs_mlp.hip); "
                           + points + "; losses: torch fp32")
        return base + ("pair inputs / aggregation (csrc/pairs.hip); Linear layers: library GEMMs (row-split weight gradients) in "
                       + ("fp32" if dt is None or isinstance(dt, str) or dt == torch.float32 else str(dt)) + " under torch autograd")

    def step(self, sample, rng=None):
        """sample: dict(images [B,T,3,H,W], intrinsics [B,T,3,3], extrinsics [B,T,4,4], obj_idx [B]) on the GPU."""
        self.model.pointnerf.train()
        self.optimizer.zero_grad(set_to_none=True)
        pred, aux = self.model.pointnerf(sample["obj_idx"], sample["intrinsics"], sample["extrinsics"], sample_rays=True, rng=rng)
        loss, sub, _ = self.loss(sample=sample, pred=pred, aux=aux, iteration=self.iteration)
        loss.backward()
        self.optimizer.step()
        self.scheduler.step()
        self.iteration += 1
        return loss.detach(), {k: v.detach() for k, v in sub.items()}

    # ---- train state in the reference's layout (utils/checkpoint_utils.py:196-236, pointnerf_training.py:180-187,212-215) ----
    def state_dict(self):
        return {"model_state_dict": self.model.state_dict(), "optimizer_state_dict": self.optimizer.state_dict(),
                "scheduler_state_dict": self.scheduler.state_dict()}

    def load_state_dict(self, ckpt):
        self.model.load_state_dict(ckpt["model_state_dict"])
        self.optimizer.load_state_dict(ckpt["optimizer_state_dict"])
        self.scheduler.load_state_dict(ckpt["scheduler_state_dict"])
        self.iteration = int(self.scheduler.last_epoch)

    def save(self, base_path: str, base_name: str = "pointnerf_training", max_to_keep=None) -> str:
        import os
        from .checkpoint import checkpoint_name, list_checkpoints
        os.makedirs(base_path, exist_ok=True)
        path = os.path.join(base_path, checkpoint_name(base_name, self.iteration))
        torch.save(self.state_dict(), path)
        if max_to_keep is not None:
            files = list_checkpoints(base_path, base_name)
            for _, old in files[:max(0, len(files) - max_to_keep)]:
                os.remove(old)
        return path

    def resume_latest(self, base_path: str, base_name: str = "pointnerf_training"):
        from .checkpoint import list_checkpoints
        found = list_checkpoints(base_path, base_name)
        if not found:
            return None
        self.load_state_dict(torch.load(found[-1][1], map_location=next(self.model.parameters()).device, weights_only=False))
        return found[-1][1]
