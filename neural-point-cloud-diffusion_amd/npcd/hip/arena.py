"""Step arena: the buffers a fused denoiser step allocates, kept and handed out again in the same order on the next step.

A rank's step at per-GPU batch 8 makes ~740 `torch.empty` calls (activations saved for the backward, temporaries, partial-sum rows):
2.2 us of host time each, 1.6 ms of a step whose host side is as long as its GPU side (docs/experiments.md R6.5).  The sequence of
(shape, dtype) requests of a step is the same every step, so the fused backbone node records it once: inside an active arena
`empty()` returns the i-th buffer of the previous step when shape, dtype and device match (otherwise it allocates and replaces the
slot: a changed batch size heals itself).  Nothing else changes: same kernels, same order, same bits.

Safety rules (enforced here, relied upon by npcd.models.diffusion.fused):
  * only tensors that live INSIDE one forward + backward of the fused node come from the arena; what leaves the node (its output, the
    gradient it returns) is allocated outside (`paused()`);
  * an arena serves ONE step at a time: `begin()` fails (returns False, the caller runs that step on plain allocations) while the
    previous step's buffers may still be referenced, i.e. until its backward has run or its graph was dropped;
  * the arena is per engine and per thread-of-use; buffers never move between devices.
Outside an active arena `empty()` is `torch.empty`.
"""
import contextlib
import threading

import torch

_tls = threading.local()


def _active():
    return getattr(_tls, "arena", None)


def empty(shape, dtype, device):
    a = _active()
    if a is None:
        return torch.empty(shape, dtype=dtype, device=device)
    return a.take(tuple(shape) if not isinstance(shape, int) else (shape,), dtype, device)


def empty_like(t):
    a = _active()
    if a is None:
        return torch.empty_like(t)
    return a.take(tuple(t.shape), t.dtype, t.device)


class StepArena:
    def __init__(self):
        self.slots = []
        self.pos = 0
        self.busy = False          # a step's buffers are handed out and its backward has not finished
        self.token = 0             # identifies the step that holds the arena (begin() -> token, end(token))
        self.hits = self.misses = 0

    def take(self, shape, dtype, device):
        i = self.pos
        self.pos = i + 1
        if i < len(self.slots):
            t = self.slots[i]
            if t.shape == shape and t.dtype == dtype and t.device == device:
                self.hits += 1
                return t
            t = self.slots[i] = torch.empty(shape, dtype=dtype, device=device)
        else:
            t = torch.empty(shape, dtype=dtype, device=device)
            self.slots.append(t)
        self.misses += 1
        return t

    def begin(self) -> int:
        """Start a step: its token (> 0), or 0 while another step's buffers are out (the caller then does without the arena for
        this step)."""
        if self.busy:
            return 0
        self.busy, self.pos = True, 0
        self.token += 1
        return self.token

    def end(self, token: int):
        """The step `token` is over -- its backward has run, or its graph was dropped: the buffers may be handed out again.  Late or
        repeated calls (a guard object collected after a newer step began) do nothing."""
        if not self.busy or token != self.token:
            return
        del self.slots[self.pos:]          # (a shorter step than the recorded one: drop the tail instead of keeping stale buffers)
        self.busy = False

    def release(self):
        self.slots, self.pos, self.busy = [], 0, False

    @contextlib.contextmanager
    def active(self):
        prev = _active()
        _tls.arena = self
        try:
            yield self
        finally:
            _tls.arena = prev


class StepGuard:
    """Held by a step's autograd context: ends the step when the context goes away without a backward (graph dropped)."""

    def __init__(self, arena: StepArena, token: int):
        self.arena, self.token = arena, token

    def __del__(self):
        try:
            self.arena.end(self.token)
        except Exception:      # noqa: BLE001  (interpreter shutdown)
            pass


@contextlib.contextmanager
def paused():
    """Allocations inside come from torch.empty (and do not advance the arena): for tensors that outlive the step."""
    prev = _active()
    _tls.arena = None
    try:
        yield
    finally:
        _tls.arena = prev
