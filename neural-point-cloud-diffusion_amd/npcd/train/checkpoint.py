"""Train-state checkpoints of the denoiser training in the reference's on-disk layout.

The reference writes (utils/checkpoint_utils.py:196-236, called from train/diffusion_training.py:212-218,252) one
`torch.save`d dictionary per checkpoint, `<base_name>-iter-<9 digits>.pt`:

    model_state_dict                         state_dict of model.diffusion
    optimizer_state_dict                     torch.optim.AdamW.state_dict(): per-parameter step / exp_avg / exp_avg_sq
    scheduler_state_dict                     StepLR(step_size=1, gamma=1.0).state_dict()   (constant learning rate)
    ema_<params>_model_state_dict            state_dict of the EMA copy of the WHOLE model (utils/ema.py:74-85)
    ema_<params>_scheduler_state_dict        EmaScheduler.__dict__ (utils/ema.py:29-31)

with <params> = "power1_0min0_9999max0_9999buffers0" for the published configuration (configs/npcd_srncars.yaml:25).
This build keeps parameters, Adam moments and the EMA in flat buffers (engine.FlatBuffers); the functions here slice
them into that layout and back, so a run can be resumed from a reference checkpoint and the reference can resume from
one written here.  The optimizer / scheduler dictionaries are produced by real torch objects of the installed torch
version, i.e. they are whatever `load_state_dict` of that version expects.
"""
import os
import re
from typing import Optional

import torch


def ema_param_string(power: float, min_value: float, max_value: float, on_buffers: bool) -> str:
    """utils/ema.py:52-55"""
    return f"power{float(power)}min{float(min_value)}max{float(max_value)}buffers{int(on_buffers)}".replace(".", "_")


def _moments(trainer):
    """(exp_avg, exp_avg_sq) as flat fp32 tensors (zeros before the first step)."""
    if trainer.native:
        return trainer.exp_avg, trainer.exp_avg_sq
    st = trainer.optimizer.state.get(trainer.master, {})
    if "exp_avg" in st:
        return st["exp_avg"], st["exp_avg_sq"]
    z = torch.zeros_like(trainer.flat.flat)
    return z, z.clone()


def trainer_state_dict(trainer, full_model: Optional[torch.nn.Module] = None, ema_prefix: str = "diffusion.") -> dict:
    """The reference's train-state dictionary for a DiffusionTrainer.  `full_model` (the NPCD module that owns
    trainer.model as `.diffusion`) makes the EMA entry cover the whole model like the reference's; without it the EMA entry
    holds the diffusion model's keys only.

    COLLECTIVE when the optimizer is sharded over ranks: EVERY rank must call it (the moment / EMA shards are all-gathered;
    a rank-0-only call blocks in the first gather like any unmatched collective -- nothing in here can detect that)."""
    trainer.gather_state()                              # Adam moments AND the EMA: gathered once, here
    model, flat = trainer.model, trainer.flat
    m, v = _moments(trainer)
    params = list(model.parameters())
    opt = torch.optim.AdamW(params, lr=trainer.lr, weight_decay=trainer.weight_decay, betas=trainer.betas, eps=trainer.eps)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=1, gamma=1.0)
    if trainer.iteration > 0:
        off = {id(p): o for p, o in zip(flat.params, flat.offsets)}
        for p in params:
            if id(p) in off:
                o = off[id(p)]
                opt.state[p] = {"step": torch.tensor(float(trainer.iteration)),
                                "exp_avg": m[o:o + p.numel()].view_as(p).clone(),
                                "exp_avg_sq": v[o:o + p.numel()].view_as(p).clone()}
    sched.last_epoch = trainer.finished_iterations
    sched._step_count = trainer.finished_iterations + 1
    ckpt = {"model_state_dict": model.state_dict(), "optimizer_state_dict": opt.state_dict(),
            "scheduler_state_dict": sched.state_dict()}
    if trainer.ema is not None:
        name = ema_param_string(1.0, trainer.ema_decay, trainer.ema_decay, False)
        ema_sd = trainer.ema_state_dict(gathered=True)
        if full_model is not None:
            whole = {k: (val.clone() if torch.is_tensor(val) else val) for k, val in full_model.state_dict().items()}
            for k, val in ema_sd.items():
                assert ema_prefix + k in whole, ema_prefix + k
                whole[ema_prefix + k] = val
            ema_sd = whole
        ckpt[f"ema_{name}_model_state_dict"] = ema_sd
        ckpt[f"ema_{name}_scheduler_state_dict"] = dict(inv_gamma=1.0, power=1.0, min_value=trainer.ema_decay, max_value=trainer.ema_decay,
                                                        start_at=0, last_epoch=trainer.finished_iterations)
    return ckpt


@torch.no_grad()
def load_trainer_state(trainer, ckpt: dict, ema_prefix: str = "diffusion.") -> None:
    """Inverse of trainer_state_dict; accepts checkpoints written by the reference (EMA entry over the whole model)."""
    model, flat = trainer.model, trainer.flat
    model.load_state_dict(ckpt["model_state_dict"])          # parameters are views of the flat buffer: this fills it
    params = list(model.parameters())
    off = {id(p): o for p, o in zip(flat.params, flat.offsets)}
    state = ckpt["optimizer_state_dict"]["state"]
    group = ckpt["optimizer_state_dict"]["param_groups"][0]
    trainer.lr, trainer.weight_decay = float(group["lr"]), float(group["weight_decay"])
    trainer.betas, trainer.eps = tuple(group["betas"]), float(group["eps"])
    m = torch.zeros_like(flat.flat)
    v = torch.zeros_like(flat.flat)
    steps = set()
    for i, p in enumerate(params):
        if i in state and id(p) in off:
            o = off[id(p)]
            m[o:o + p.numel()].copy_(state[i]["exp_avg"].reshape(-1))
            v[o:o + p.numel()].copy_(state[i]["exp_avg_sq"].reshape(-1))
            steps.add(int(state[i]["step"]))
    assert len(steps) <= 1, f"parameters with different step counts: {sorted(steps)}"
    trainer.iteration = steps.pop() if steps else 0
    trainer.finished_iterations = int(ckpt.get("scheduler_state_dict", {}).get("last_epoch", trainer.iteration))
    if trainer.native:
        trainer.exp_avg.copy_(m)
        trainer.exp_avg_sq.copy_(v)
        trainer._ew.cast_f32_bf16(flat.flat, trainer.shadow)
        trainer._shadow_written()
    else:
        trainer.optimizer.param_groups[0].update(lr=trainer.lr, weight_decay=trainer.weight_decay, betas=trainer.betas, eps=trainer.eps)
        if trainer.iteration > 0:
            step = torch.tensor(float(trainer.iteration), device=flat.flat.device) if flat.flat.is_cuda else torch.tensor(float(trainer.iteration))
            trainer.optimizer.state[trainer.master] = {"step": step, "exp_avg": m, "exp_avg_sq": v}
        else:
            trainer.optimizer.state.pop(trainer.master, None)
    if trainer.ema is not None:
        keys = [k for k in ckpt if re.fullmatch(r"ema_.*_model_state_dict", k)]
        assert keys, "checkpoint holds no EMA model"
        ema_sd = ckpt[keys[0]]
        names = {id(p): n for n, p in model.named_parameters()}
        for p, o in zip(flat.params, flat.offsets):
            n = names[id(p)]
            src = ema_sd[n] if n in ema_sd else ema_sd[ema_prefix + n]
            trainer.ema[o:o + p.numel()].copy_(src.reshape(-1))


def checkpoint_name(base_name: str, iteration: int) -> str:
    """utils/checkpoint_utils.py:203-208"""
    return f"{base_name}-iter-{iteration:09d}.pt"


def list_checkpoints(base_path: str, base_name: str = "diffusion_training"):
    """[(iteration, path)] sorted by iteration (utils/checkpoint_utils.py:94-108,239-246)."""
    out = []
    if os.path.isdir(base_path):
        for f in os.listdir(base_path):
            mt = re.fullmatch(re.escape(base_name) + r"-iter-(\d{9})\.pt", f)
            if mt:
                out.append((int(mt.group(1)), os.path.join(base_path, f)))
    return sorted(out)


def save_train_state(trainer, base_path: str, base_name: str = "diffusion_training", max_to_keep: Optional[int] = None,
                     full_model: Optional[torch.nn.Module] = None) -> str:
    """Write `<base_name>-iter-<finished_iterations>.pt`; with max_to_keep the oldest files are removed (:226-234).
    EVERY rank MUST call this when the optimizer is sharded (the shards are gathered collectively, see
    trainer_state_dict); rank 0 writes."""
    ckpt = trainer_state_dict(trainer, full_model)
    path = os.path.join(base_path, checkpoint_name(base_name, trainer.finished_iterations))
    if trainer.reducer.rank == 0:
        os.makedirs(base_path, exist_ok=True)
        torch.save(ckpt, path)
        if max_to_keep is not None:
            files = list_checkpoints(base_path, base_name)
            for _, old in files[:max(0, len(files) - max_to_keep)]:
                os.remove(old)
    return path


def resume_latest(trainer, base_path: str, base_name: str = "diffusion_training") -> Optional[str]:
    """Load the newest checkpoint under base_path if there is one (train/diffusion_training.py:231-243)."""
    found = list_checkpoints(base_path, base_name)
    if not found:
        return None
    path = found[-1][1]
    load_trainer_state(trainer, torch.load(path, map_location=trainer.flat.flat.device, weights_only=False))
    return path
