import torch, time
dev="cuda"
T,K,N=8208,3072,3072
x=torch.randn(T,K,device=dev).bfloat16(); w=torch.randn(N,K,device=dev).bfloat16(); b=torch.randn(N,device=dev)
def t(fn,reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/reps*1e3
print("mm out_dtype", t(lambda: torch.mm(x,w.t(),out_dtype=torch.float32)))
print("mm + add_", t(lambda: torch.mm(x,w.t(),out_dtype=torch.float32).add_(b)))
try:
    y=torch.addmm(b, x, w.t(), out_dtype=torch.float32)
    ref=torch.mm(x,w.t(),out_dtype=torch.float32)+b
    print("addmm out_dtype ok", float((y-ref).abs().max()), t(lambda: torch.addmm(b, x, w.t(), out_dtype=torch.float32)))
except Exception as e:
    print("addmm out_dtype:", type(e).__name__, str(e)[:200])
try:
    out=torch.empty(T,N,device=dev)
    bb=b.expand(T,N)
    y=torch.addmm(bb, x, w.t(), out_dtype=torch.float32)
    print("addmm expanded ok")
except Exception as e:
    print("addmm expanded:", type(e).__name__, str(e)[:200])
