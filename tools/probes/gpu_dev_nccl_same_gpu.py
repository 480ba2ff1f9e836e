"""Dev probe: can two ranks share one GPU under RCCL (to exercise the nccl code path on a 1-GPU box)?"""
import os, torch, torch.distributed as dist
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", rank=rank, world_size=world)
    x = torch.full((1024,), float(rank + 1), device="cuda")
    dist.all_reduce(x, op=dist.ReduceOp.AVG)
    torch.cuda.synchronize()
    print("rank", rank, "allreduce avg ->", float(x[0]), flush=True)
    y = torch.empty(512, device="cuda"); inp = torch.arange(1024, device="cuda", dtype=torch.float32)
    dist.reduce_scatter_tensor(y, inp, op=dist.ReduceOp.SUM); torch.cuda.synchronize()
    print("rank", rank, "reduce_scatter ok", float(y[0]), flush=True)
except Exception as e:
    print("rank", rank, "FAILED:", type(e).__name__, str(e)[:300], flush=True)
