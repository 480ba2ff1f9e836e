"""Per-object embedding tables (reference npcd/models/pointnerf/embeddings/*.py, utils/flex_embedding.py).

Checkpoint format kept: the table does not appear as an ordinary state_dict key; it travels as
``<name>._extra_state = {"emb": {"weight": Parameter[n_obj, n_kp*dim]}}`` and a shape mismatch on
load re-initialises silently with a warning (flex_embedding.py:9-26)."""
import warnings

import torch
import torch.nn as nn


class FlexEmbedding(nn.Embedding):
    def get_extra_state(self):
        return {"weight": self.weight}

    def set_extra_state(self, state):
        if state is None:
            return
        if "weight" in state and state["weight"].shape == self.weight.shape:
            with torch.no_grad():
                self.weight.copy_(state["weight"])
        else:
            warnings.warn("Found unequal shapes of embeddings in module and state_dict. "
                          "Continue with re-initialized embedding.")

    def state_dict(self, *args, **kwargs):                 # table is carried by the owner's extra state
        return args[0] if args else kwargs["destination"]

    def _load_from_state_dict(self, *args, **kwargs):
        return


class Embedding(nn.Module):
    """idx [B] -> [B, n_kp, out_dim] (embedding.py:29-43)."""
    channels_per_point = 1      # multiplier on out_dim for the table width

    def __init__(self, n_kp: int, out_dim: int, n_obj: int, gpu: bool = True):
        super().__init__()
        self.n_kp, self.out_dim, self.n_obj, self.gpu = n_kp, out_dim, n_obj, gpu
        self.emb = FlexEmbedding(n_obj, n_kp * out_dim * self.channels_per_point)
        nn.init.zeros_(self.emb.weight)

    def get_emb(self):
        return self.emb

    def _rows(self, idx):
        return self.emb(idx).view(-1, self.n_kp, self.out_dim * self.channels_per_point)

    def forward(self, idx):
        return self._rows(idx)

    def get_extra_state(self):
        return {"emb": self.emb.get_extra_state()}

    def set_extra_state(self, state):
        if state is not None and "emb" in state:
            self.emb.set_extra_state(state["emb"])

    def freeze(self, emb: bool = False):
        if emb:
            for p in self.emb.parameters():
                p.requires_grad = False
            self.emb.eval()


class VariationalEmbedding(Embedding):
    """Table rows are [mean | log_var]; train mode samples mean + exp(log_var/2) * eps, eval mode
    returns the mean (variational_embedding.py:36-58)."""
    channels_per_point = 2

    def __init__(self, n_kp, out_dim, n_obj, gpu=True):
        super().__init__(n_kp, out_dim, n_obj, gpu)
        self.sample_embedding = True

    def train(self, mode=True):
        super().train(mode)
        self.sample_embedding = mode
        return self

    def forward(self, idx, eps=None):
        """`eps` replays a given standard-normal draw (tests); by default it is drawn here (reparametrisation trick)."""
        rows = self._rows(idx)
        mean = rows[:, :, :self.out_dim]
        if not self.sample_embedding:
            return mean
        std = torch.exp(0.5 * rows[:, :, self.out_dim:])
        return mean + std * (torch.randn_like(std) if eps is None else eps.to(std))

    def get_mean_log_var_std(self, idx):
        rows = self._rows(idx)
        mean, log_var = rows[:, :, :self.out_dim], rows[:, :, self.out_dim:]
        return mean, log_var, torch.exp(0.5 * log_var)
