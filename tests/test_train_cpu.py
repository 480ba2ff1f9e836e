"""CPU tests of the training-engine host logic: flat buffers, gradient reducer (world_size 2, gloo),
optimizer / EMA bookkeeping.  The model used here is a small torch module -- the engine is agnostic
of what produces the gradients (the denoiser itself needs the GPU)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from npcd.train import FlatBuffers, GradReducer


class Toy(nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.a = nn.Linear(7, 5)
        self.b = nn.Linear(5, 3)
        self.unused = nn.Linear(2, 2)

    def forward(self, x):
        return self.b(torch.tanh(self.a(x)))


def test_flat_buffers_alias_parameters():
    m = Toy()
    ref = {k: v.clone() for k, v in m.state_dict().items()}
    fb = FlatBuffers(m)
    for k, v in m.state_dict().items():
        assert torch.equal(v, ref[k])
    assert all(off % 4 == 0 for off in fb.offsets)
    m(torch.randn(4, 7)).sum().backward()
    assert float(fb.grad.abs().sum()) > 0
    for p, off in zip(fb.params, fb.offsets):
        assert p.grad.data_ptr() == fb.grad.data_ptr() + 4 * off
        assert p.data_ptr() == fb.flat.data_ptr() + 4 * off
    fb.zero_grad()
    assert float(fb.grad.abs().sum()) == 0
    with torch.no_grad():
        fb.flat.add_(1.0)
    assert torch.allclose(m.a.weight, ref["a.weight"] + 1.0)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    return port


def _worker(rank, world, port, bucket_bytes, out, comm_dtype=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m = Toy()
        fb = FlatBuffers(m)
        red = GradReducer(fb, bucket_bytes=bucket_bytes, comm_dtype=comm_dtype)
        torch.manual_seed(100 + rank)
        x = torch.randn(6, 7)
        fb.zero_grad(); red.start_step()
        m(x).pow(2).mean().backward()
        red.finish()
        out[rank] = (fb.grad.clone(), x, len(red.buckets), red.wire_bytes)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("bucket_bytes", [64, 1 << 20])
def test_grad_reducer_world2_gloo(bucket_bytes):
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), bucket_bytes, out), nprocs=world, join=True)
    g0, x0, nb, _ = out[0]
    g1, x1, _, _ = out[1]
    assert torch.equal(g0, g1), "ranks disagree after all-reduce"
    if bucket_bytes == 64:
        assert nb > 1
    # reference: mean of the two ranks' gradients == gradient of the mean loss over both shards
    m = Toy()
    fb = FlatBuffers(m)
    (0.5 * m(x0).pow(2).mean() + 0.5 * m(x1).pow(2).mean()).backward()
    assert torch.allclose(g0, fb.grad, atol=1e-6)
    off = fb.offsets[[id(p) for p in fb.params].index(id(m.unused.weight))]
    assert float(g0[off:off + 4].abs().sum()) == 0.0          # unused parameters stay zero, no hang


def test_grad_reducer_bf16_buckets_world2_gloo():
    """comm_dtype=bfloat16: the buckets travel as bf16 (half the bytes on the wire), ranks still agree bit for bit, and the averaged
    gradient equals the fp32 average to bf16 rounding (one rounding per rank's bucket + one of the sum: 3 x 2^-9 relative)."""
    world = 2
    mgr = mp.Manager()
    o16, o32 = mgr.dict(), mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), 64, o16, torch.bfloat16), nprocs=world, join=True)
    mp.spawn(_worker, args=(world, _free_port(), 64, o32, None), nprocs=world, join=True)
    assert torch.equal(o16[0][0], o16[1][0]), "ranks disagree after the bf16 all-reduce"
    assert o16[0][3] * 2 == o32[0][3] and o32[0][3] == o32[0][0].numel() * 4
    ref = o32[0][0]
    assert float((o16[0][0] - ref).abs().max()) <= 3 * 2.0 ** -8 * float(ref.abs().max())
    assert not torch.equal(o16[0][0], ref)


def test_trainer_bookkeeping_cpu():
    """AdamW over the flat buffer == AdamW over the individual tensors; EMA = lerp(ema, p, 1-decay)."""
    from npcd.train import DiffusionTrainer

    class FakeDiffusion(Toy):
        def compute_loss(self, coords, feats, t=None, coords_noise=None, feats_noise=None):
            l = self(coords).pow(2).mean()
            return l, {"00_coords_loss": l}, {}

    torch.manual_seed(1)
    x = torch.randn(8, 7)
    a, b = FakeDiffusion(), FakeDiffusion()
    tr = DiffusionTrainer(a, lr=1e-2, weight_decay=0.01, ema_decay=0.9, dtype=None)
    opt = torch.optim.AdamW(b.parameters(), lr=1e-2, weight_decay=0.01)
    ema = [p.detach().clone() for p in b.parameters()]
    for _ in range(3):
        tr.step(x, None)
        opt.zero_grad()
        b.compute_loss(x, None)[0].backward()
        for p in b.parameters():                      # the flat AdamW also decays parameters without grad
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        opt.step()
        for e, p in zip(ema, b.parameters()):
            e.lerp_(p.detach(), 0.1)
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.allclose(pa, pb, atol=1e-6)
    sd = tr.ema_state_dict()
    for (n, _), e in zip(b.named_parameters(), ema):
        assert torch.allclose(sd[n], e, atol=1e-6)
    c = tr.comm_stats()                               # single process: nothing on the wire
    assert c["mode"] == "none" and c["world"] == 1 and c["gradient_bytes_handed_to_collectives"] == 0
    tr.close()                                        # idempotent, also on the torch-optimizer path
    tr.close()
