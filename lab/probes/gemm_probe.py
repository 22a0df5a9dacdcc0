import torch, time
dev=torch.device('cuda:0')
def t(fn,n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e3
for (M,K,N) in [(8192,512,512),(8192,512,1536),(8192,1536,512)]:
    a=torch.randn(M,K,device=dev).bfloat16(); b=torch.randn(N,K,device=dev).bfloat16()
    print(f"fwd  y[{M}x{N}] = x[{M}x{K}] w^T: torch.matmul {t(lambda: a@b.t()):6.1f} us")
    dy=torch.randn(M,N,device=dev).bfloat16()
    print(f"wgrad dw[{N}x{K}] = dy^T x       : torch.matmul {t(lambda: dy.t()@a):6.1f} us")
