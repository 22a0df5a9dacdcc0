"""The six GEMMs of an attention block (model/nn.py:45-47; B = 128 images x 64 tokens, 512 channels) through the vendor library
(torch.matmul = hipBLASLt), for comparison with this package's 1x1 kernels in bench.py's by_kernel table."""
import torch, time
dev="cuda:0"
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e3
M=8192
for name,(m,k,n) in {"fwd qkv":(M,512,1536),"fwd proj":(M,512,512),"dgrad qkv":(M,1536,512),"dgrad proj":(M,512,512)}.items():
    a=torch.randn(m,k,device=dev).bfloat16(); w=torch.randn(n,k,device=dev).bfloat16()
    us=t(lambda: torch.matmul(a,w.t()))
    print(f"{name:12s} [{m}x{k}]x[{k}x{n}] {us:7.1f} us  {2*m*k*n/us/1e6:7.1f} TFLOP/s")
for name,(co,ci) in {"wgrad qkv":(1536,512),"wgrad proj":(512,512)}.items():
    dy=torch.randn(M,co,device=dev).bfloat16(); x=torch.randn(M,ci,device=dev).bfloat16()
    us=t(lambda: torch.matmul(dy.t(),x))
    us32=t(lambda: torch.matmul(dy.t().float(),x.float()))
    print(f"{name:12s} [{co}x{M}]x[{M}x{ci}] {us:7.1f} us  {2*M*co*ci/us/1e6:7.1f} TFLOP/s (bf16 out); fp32 operands {us32:.1f} us")
