"""Run-to-run determinism of the training step's forward + backward on the two streams it really uses (forward / input gradients on
one, weight gradients + their reductions on the other): the same batch, (t, eps) and weights ROUNDS times; every gradient tensor whose
summation order is fixed (the conv / linear WEIGHT gradients: split-K partial sums reduced in a fixed order) must come out bit-identical
every time.  Bias, LayerNorm-modulation and loss sums use fp32 atomics and are reported separately (they may differ in the last bits).
A race between workgroups -- like the weight-ring one of round 3 -- shows up here as a weight gradient that changes between rounds."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from climate2weather_amd.score import ScoreUNet
from climate2weather_amd.training import Trainer

ROUNDS = int(os.environ.get("ROUNDS", "40"))
PREC = os.environ.get("PRECISION", "bf16")
B, C, H = 128, 65, 128
CFG = dict(embedding_dim=512, hidden_blocks=[3] * 5, hidden_channels=[128, 128, 256, 384, 512], kernel_size=3, padding_mode="zeros", attention_levels=[4])
torch.manual_seed(0)
net = ScoreUNet(channels=C, spatial=2, activation=torch.nn.SiLU, **CFG).cuda()
gen = torch.Generator().manual_seed(128)
x = (torch.randn(B, C, H, H, generator=gen) * 0.5 + 0.5).cuda()
t = torch.rand(B, generator=gen).cuda()
eps = torch.randn(B, C, H, H, generator=gen).cuda()
tr = Trainer(net, precision=PREC, ema_rates=())
named = dict(net.named_parameters())
# conv kernels (4-D): split-K partial sums reduced in a fixed order from deterministic operands.  The Linear weights (time MLP, the
# blocks' modulation projections) take the atomically-summed modulation gradients as their operand, so they inherit those last bits.
fixed = [n for n, p in named.items() if p.dim() == 4]
loose = [n for n, p in named.items() if p.dim() != 4]
ref, bad_fixed, bad_loose, losses = None, 0, 0, []
for r in range(ROUNDS):
    tr.eng.flat_grad.zero_()
    loss = tr._forward_backward(x, t, eps, sync=False)
    torch.cuda.synchronize()
    losses.append(loss.item())
    cur = {n: named[n].grad.detach().clone() for n in named}
    if ref is None:
        ref = cur
        continue
    diff = [n for n in fixed if not torch.equal(cur[n], ref[n])]
    if diff:
        bad_fixed += 1
        n = diff[0]
        d = (cur[n].float() - ref[n].float()).abs()
        print(f"round {r}: {len(diff)} weight gradients differ, first {n}: {int((d > 0).sum())} values, max |d| {d.max().item():.3e} of {ref[n].float().abs().max().item():.3e}", flush=True)
    bad_loose += any(not torch.equal(cur[n], ref[n]) for n in loose)
print(f"{PREC}: {len(fixed)} conv kernels, rounds with a changed conv WEIGHT gradient: {bad_fixed} of {ROUNDS - 1}; rounds with a changed bias / vector / Linear gradient (fp32 atomics upstream): {bad_loose}; "
      f"loss min {min(losses):.7f} max {max(losses):.7f}")
