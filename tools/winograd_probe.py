"""Winograd F(2x2, 3x3) for the dominant conv (3x3 stride 1, 128 -> 128 @128x128, B = 128) as a measured decision (profiles/r03_winograd.md).

Three numbers, all on the box this runs on:
  1. the rounding error of the Winograd form with 16-bit MFMA operands (input / weight transforms in fp32, rounded to bf16 / fp16 as the
     matrix core would be fed, fp32 accumulation, output transform in fp32) next to the direct kernel's, both against an fp64 direct
     convolution of the same 16-bit inputs;
  2. the time of the 16 channel-GEMMs alone (hipBLASLt through torch.bmm, transformed operands already in HBM): the matrix-core part of
     an UNFUSED Winograd pipeline -- a lower bound that ignores both transforms;
  3. the HBM bytes an unfused pipeline moves around those GEMMs, against the direct kernel's measured time.
Nothing here is product code: plain torch ops + ops.conv for the direct kernel.
"""
import math, os, sys, json
sys.path.insert(0, os.getcwd())
import torch
import torch.nn.functional as F
from climate2weather_amd import ops

dev = torch.device("cuda:0")
torch.backends.cuda.matmul.allow_tf32 = False
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64, device=dev)
G = torch.tensor([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=torch.float64, device=dev)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64, device=dev)


def winograd_conv(x, w, T):
    """x: (B, C, H, W), w: (K, C, 3, 3) already rounded to T; returns fp32 (B, K, H, W).  Transforms in fp32, operands of the 16
    channel-GEMMs rounded to T (what a 16-bit MFMA would read), fp32 accumulation."""
    B, C, H, W = x.shape
    K = w.shape[0]
    xp = F.pad(x.float(), (1, 1, 1, 1))
    tiles = xp.unfold(2, 4, 2).unfold(3, 4, 2)                       # (B, C, H/2, W/2, 4, 4)
    V = torch.einsum("ij,bchwjk,lk->bchwil", BT.float(), tiles, BT.float())  # B^T d B
    U = torch.einsum("ij,kcjl,ml->kcim", G.float(), w.float(), G.float())    # G g G^T  (K, C, 4, 4)
    V16, U16 = V.to(T).float(), U.to(T).float()
    M = torch.einsum("kcim,bchwim->bkhwim", U16, V16)                 # 16 GEMMs over channels, fp32 accumulate
    Y = torch.einsum("ij,bkhwjl,ml->bkhwim", AT.float(), M, AT.float())      # (B, K, H/2, W/2, 2, 2)
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(B, K, H, W)


res = {}
# ---- 1. error, on a slice small enough for an fp64 reference: B = 2, 128 ch, 128 x 128
torch.manual_seed(0)
Bs, C, H = 2, 128, 128
for name, T, dt in (("bf16", torch.bfloat16, ops.DTYPE_BF16), ("fp16", torch.float16, ops.DTYPE_F16)):
    x = torch.randn(Bs, C, H, H, device=dev).to(T)
    w = (torch.randn(C, C, 3, 3, device=dev) / math.sqrt(9 * C)).to(T)
    ref = F.conv2d(x.double(), w.double(), padding=1)
    scale = ref.abs().max().item()
    yw = winograd_conv(x, w, T)
    # the direct HIP kernel on the same operands (fp32 accumulate, output rounded to T -- Winograd's output would be rounded the same way)
    xn = x.permute(0, 2, 3, 1).reshape(-1, C).contiguous()
    wn = w.permute(0, 2, 3, 1).reshape(C, 9, C).contiguous()
    y = torch.empty_like(xn)
    g = dict(B=Bs, Hin=H, Win=H, Cin=C, Hout=H, Wout=H, Cout=C, ldy=C, wrows=C, mode=ops.CONV_S1)
    ops.conv(xn, wn, None, y, g, dt)
    yd = y.view(Bs, H, H, C).permute(0, 3, 1, 2).double()
    res[name] = dict(direct_kernel_err_of_scale=(yd - ref).abs().max().item() / scale,
                     winograd_err_of_scale_before_output_rounding=(yw.double() - ref).abs().max().item() / scale,
                     winograd_err_of_scale_after_output_rounding=(yw.to(T).double() - ref).abs().max().item() / scale,
                     rms_direct=((yd - ref) ** 2).mean().sqrt().item() / scale, rms_winograd=((yw.to(T).double() - ref) ** 2).mean().sqrt().item() / scale)

# ---- 2. the 16 channel-GEMMs of the bench's launch: B = 128 -> 128 x 64 x 64 = 524288 tiles, [tiles x 128] x [128 x 128] each
ntile = 128 * 64 * 64
V = torch.randn(16, ntile, 128, device=dev).to(torch.bfloat16)
U = torch.randn(16, 128, 128, device=dev).to(torch.bfloat16)
for _ in range(3):
    M = torch.bmm(V, U)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    M = torch.bmm(V, U)
e1.record()
torch.cuda.synchronize()
t_gemm = e0.elapsed_time(e1) / 10
flop_w = 2.0 * 16 * ntile * 128 * 128
res["gemms_16x"] = dict(ms=t_gemm, tflops=flop_w / t_gemm / 1e9, winograd_gflop=flop_w / 1e9, direct_gflop=2.0 * 128 * 128 * 128 * 128 * 9 * 128 / 1e9,
                        bytes_in_out=V.numel() * 2 + M.numel() * 2)
# ---- 3. the direct kernel on this box, same launch (bias only)
x = torch.randn(128 * 128 * 128, 128, device=dev).to(torch.bfloat16)
w = (torch.randn(128, 9, 128, device=dev) / 34).to(torch.bfloat16)
bias = torch.randn(128, device=dev)
y = torch.empty_like(x)
g = dict(B=128, Hin=128, Win=128, Cin=128, Hout=128, Wout=128, Cout=128, ldy=128, wrows=128, mode=ops.CONV_S1)
for _ in range(3):
    ops.conv(x, w, bias, y, g, ops.DTYPE_BF16)
e0.record()
for _ in range(10):
    ops.conv(x, w, bias, y, g, ops.DTYPE_BF16)
e1.record()
torch.cuda.synchronize()
res["direct_kernel_ms"] = e0.elapsed_time(e1) / 10
print(json.dumps(res, indent=1))
