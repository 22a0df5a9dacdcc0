"""End-to-end GPU parity of the HIP ScoreUNet against (a) the golden vectors generated from the imported reference
and (b) the CPU oracle on fresh seeded inputs.  fp32 mode: <= 1e-4 relative (BASELINE.json north_star);
bf16 mode: <= 3e-2 of the output scale (stated tolerance for the throughput mode); fp16 mode (the reference's autocast type): <= 5e-3
(three more significand bits than bf16).  fp16 gradients are taken of a scaled loss, as under the reference's GradScaler."""
import json
import os

import numpy as np
import pytest
import torch

from climate2weather_amd.score import ScoreUNet
from oracle import diffusion as od
from oracle import unet as ou

pytestmark = pytest.mark.gpu

TINY = dict(embedding_dim=64, hidden_channels=[32, 64], hidden_blocks=[1, 1], attention_levels=[1], kernel_size=3,
            padding_mode="zeros")
DEFAULT = dict(embedding_dim=512, hidden_blocks=[3] * 5, hidden_channels=[128, 128, 256, 384, 512], kernel_size=3,
               padding_mode="zeros", attention_levels=[4])


def _golden(golden_dir, name):
    return {k: v for k, v in np.load(os.path.join(golden_dir, name), allow_pickle=False).items()}


def _rel(a, b):
    return (a.float().cpu() - b.float().cpu()).abs().max().item() / max(b.float().abs().max().item(), 1e-12)


def test_library_is_the_hip_build():
    from climate2weather_amd import _lib
    assert _lib.load().c2w_target() == b"gfx950"
    assert torch.cuda.is_available()


def test_tiny_net_fp32_forward_and_grads_vs_golden(golden_dir):
    g = _golden(golden_dir, "tiny_net.npz")
    torch.manual_seed(3)
    net = ScoreUNet(channels=6, spatial=2, activation=torch.nn.SiLU, **TINY).cuda()
    net.precision = "fp32"
    for k, v in net.state_dict().items():
        assert np.array_equal(v.cpu().numpy(), g["sd." + k]), k
    x, t, eps, xt = (torch.from_numpy(g[k]).cuda() for k in ("x", "t", "eps", "xt"))
    with torch.no_grad():
        y = net(xt, t)
        y32 = net(torch.from_numpy(g["x32"]).cuda(), torch.tensor(0.3))
    assert _rel(y, torch.from_numpy(g["y"])) <= 1e-4
    assert _rel(y32, torch.from_numpy(g["y32"])) <= 1e-4
    loss = od.loss(net, x, t, eps).mean()
    assert loss.item() == pytest.approx(float(g["loss"]), rel=1e-4)
    loss.backward()
    torch.cuda.synchronize()
    worst = 0.0
    for n, p in net.named_parameters():
        ref = torch.from_numpy(g["grad." + n])
        assert p.grad is not None, n
        r = _rel(p.grad, ref)
        worst = max(worst, r)
        assert r <= 2e-4, (n, r)  # wgrad sums are order-dependent (atomics): 2e-4 of the tensor's scale
    print("worst relative grad error", worst)


@pytest.mark.parametrize("precision,tol", [("fp32", 2e-4), ("bf16", 3e-2)])
def test_forcing_projection_vs_reference(golden_dir, precision, tol):
    """forcing_dim > 0 (model/score.py:49-51,65-66) on the GPU: weights from the seed (map_forcing is created first), output, loss and
    gradients against the imported reference's (tests/golden/tiny_net_forcing.npz); 5 forcing features ride a K chunk padded to 32."""
    g = _golden(golden_dir, "tiny_net_forcing.npz")
    cfg = dict(TINY, hidden_channels=[64, 128]) if precision != "fp32" else TINY
    torch.manual_seed(3)
    net = ScoreUNet(channels=6, spatial=2, activation=torch.nn.SiLU, forcing_dim=5, **cfg).cuda()
    net.precision = precision
    x, t, eps, forcing = (torch.from_numpy(g[k]).cuda() for k in ("x", "t", "eps", "forcing"))
    if precision == "fp32":
        assert sum(v.double().abs().sum().item() for v in net.state_dict().values()) == pytest.approx(float(g["sd_abs_sum"]), rel=1e-12)
        y = net(x, t, forcing=forcing)
        assert _rel(y, torch.from_numpy(g["y"])) <= 1e-4
        loss = ((y - eps) ** 2).mean()
        assert loss.item() == pytest.approx(float(g["loss"]), rel=1e-4)
        loss.backward()
        named = dict(net.named_parameters())
        for n, ref in zip([str(v) for v in g["names"]], g["grad_norm"]):
            assert named[n].grad.double().norm().item() == pytest.approx(float(ref), rel=5e-4, abs=1e-9), n
        for n in ("map_forcing.weight", "map_forcing.bias", "map_layer1.weight", "map_layer0.bias", "unet.heads.0.weight"):
            assert _rel(named[n].grad, torch.from_numpy(g["grad." + n])) <= tol, n
        with torch.no_grad():
            assert _rel(net(x[:1], torch.tensor(0.3), forcing=forcing[:1]), torch.from_numpy(g["y_scalar_t"])) <= 1e-4
        return
    # 16-bit: a wider tiny net (whole 64-channel K chunks) against the CPU oracle on the same weights
    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    y = net(x, t, forcing=forcing)
    yo = ou.score_unet_forward(sd, x.cpu(), t.cpu(), cfg["hidden_blocks"], cfg["attention_levels"], forcing=forcing.cpu())
    assert _rel(y, yo.detach()) <= tol
    ((y - eps) ** 2).mean().backward()
    go = torch.autograd.grad(((yo - eps.cpu()) ** 2).mean(), [sd["map_forcing.weight"], sd["map_layer1.weight"]])
    named = dict(net.named_parameters())
    assert _rel(named["map_forcing.weight"].grad, go[0]) <= 2 * tol and _rel(named["map_layer1.weight"].grad, go[1]) <= 2 * tol


def test_tiny_net_input_gradient_fp32(golden_dir):
    g = _golden(golden_dir, "tiny_net.npz")
    sd = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd.")}
    torch.manual_seed(3)
    net = ScoreUNet(channels=6, spatial=2, activation=torch.nn.SiLU, **TINY).cuda()
    net.precision = "fp32"
    x = torch.from_numpy(g["x32"])[:2]
    w = torch.randn(2, 6, 32, 32, generator=torch.Generator().manual_seed(0))
    xo = x.clone().requires_grad_(True)
    (gxo,) = torch.autograd.grad((ou.score_unet_forward(sd, xo, torch.tensor(0.3), [1, 1], [1]) * w).sum(), xo)
    xg = x.cuda().requires_grad_(True)
    (gx,) = torch.autograd.grad((net(xg, torch.tensor(0.3)) * w.cuda()).sum(), xg)
    assert _rel(gx, gxo) <= 2e-4
    J = torch.func.jacrev(lambda xx: (net(xx, torch.tensor(0.3)) * w.cuda()).sum(), chunk_size=1)(x.cuda())
    assert _rel(J, gxo) <= 2e-4


@pytest.mark.parametrize("per_sample_t", [True, False])
def test_small_net_bf16_vs_oracle(per_sample_t):
    cfg = dict(embedding_dim=128, hidden_channels=[64, 128, 128], hidden_blocks=[2, 1, 1], attention_levels=[2], kernel_size=3,
               padding_mode="zeros")
    torch.manual_seed(5)
    net = ScoreUNet(channels=13, spatial=2, activation=torch.nn.SiLU, **cfg)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    net = net.cuda()
    g = torch.Generator().manual_seed(9)
    x = torch.randn(3, 13, 32, 32, generator=g)
    t = torch.rand(3, generator=g) if per_sample_t else torch.tensor(0.6)
    eps = torch.randn(3, 13, 32, 32, generator=g)
    sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    yo = ou.score_unet_forward(sdo, x, t, cfg["hidden_blocks"], cfg["attention_levels"])
    lo = ((yo - eps) ** 2).mean()
    go = torch.autograd.grad(lo, list(sdo.values()))
    for prec, tol_y, tol_g, S in (("fp32", 1e-4, 3e-4, 1.0), ("bf16", 3e-2, 8e-2, 1.0), ("fp16", 5e-3, 1.5e-2, 4096.0)):
        net.precision = prec
        net.zero_grad(set_to_none=True)
        y = net(x.cuda(), t.cuda())
        assert _rel(y, yo.detach()) <= tol_y, (prec, _rel(y, yo.detach()))
        l = ((y - eps.cuda()) ** 2).mean()
        (l * S).backward()  # fp16: scaled loss (GradScaler), else the 1/N gradients underflow half precision
        named = dict(net.named_parameters())
        for (k, _), gr in zip(sdo.items(), go):
            r = _rel(named[k].grad / S, gr)
            assert r <= tol_g, (prec, k, r)


def test_autocast_selects_bf16_and_matches():
    torch.manual_seed(5)
    cfg = dict(embedding_dim=128, hidden_channels=[64, 128], hidden_blocks=[1, 1], attention_levels=[], kernel_size=3)
    net = ScoreUNet(channels=4, spatial=2, activation=torch.nn.SiLU, **cfg).cuda().eval()
    x = torch.randn(2, 4, 16, 16, device="cuda")
    t = torch.tensor([0.2, 0.9], device="cuda")
    with torch.no_grad():
        y32 = net(x, t)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y16 = net(x, t)
            assert net.compute_dtype() == 1
        with torch.autocast("cuda", dtype=torch.float16):  # the reference's Fabric precision="16-mixed" (train.py:98)
            yh = net(x, t)
            assert net.compute_dtype() == 2
        assert net.compute_dtype() == 0
    assert _rel(y16, y32) <= 3e-2 and not torch.equal(y16, y32)
    assert _rel(yh, y32) <= 5e-3 and not torch.equal(yh, y32) and not torch.equal(yh, y16)


def test_full_size_net_fp32_vs_reference_fingerprint(golden_dir):
    """Default configs/sda_unet.yml network, C=52, 128x128: output of the imported reference captured as a strided slice."""
    fp = json.load(open(os.path.join(golden_dir, "full_net_fingerprint.json")))
    ref = torch.from_numpy(np.load(os.path.join(golden_dir, "full_net_slice.npz"))["y_slice"])
    torch.manual_seed(0)
    net = ScoreUNet(channels=52, spatial=2, activation=torch.nn.SiLU, **DEFAULT).cuda().eval()
    net.precision = "fp32"
    x = torch.randn(1, 52, 128, 128, generator=torch.Generator().manual_seed(1234))
    with torch.no_grad():
        y = net(x.cuda(), torch.tensor([0.3], device="cuda")).cpu()
    assert y.double().mean().item() == pytest.approx(fp["y_mean"], abs=1e-5)
    assert y.double().std().item() == pytest.approx(fp["y_std"], rel=1e-4)
    assert _rel(y[:, :, ::16, ::16], ref) <= 1e-4
    net.precision = "bf16"
    with torch.no_grad():
        yb = net(x.cuda(), torch.tensor([0.3], device="cuda")).cpu()
    assert _rel(yb[:, :, ::16, ::16], ref) <= 3e-2
    net.precision = "fp16"
    with torch.no_grad():
        yh = net(x.cuda(), torch.tensor([0.3], device="cuda")).cpu()
    assert _rel(yh[:, :, ::16, ::16], ref) <= 5e-3


def _rl2_cos(a, b):
    """(relative L2 error |a - b|_2 / |b|_2, cosine) of two tensors in float64: unlike max|a - b| / max|b| these see an error that
    sits in a tensor's small-magnitude entries."""
    a, b = a.double().flatten().cpu(), b.double().flatten().cpu()
    nb = b.norm().item()
    return (a - b).norm().item() / max(nb, 1e-300), (a @ b).item() / max(a.norm().item() * nb, 1e-300)


# (relative L2 <=, cosine >=) per gradient tensor the fixtures hold (strided slices of 44 weight gradients incl. every modulation
# projection, all 114 bias gradients in full).  Round 6; bounds = about twice the worst case observed on MI355X
# (profiles/r06_full_grad_parity.txt)
# observed (4 identical runs, profiles/r06_full_grad_parity.txt): fp32 1.5e-6 / 1 - 1.1e-12; bf16 1.27e-2 / 1 - 7.3e-5; fp16 1.57e-3 / 1 - 1.1e-6
FULL_GRAD_L2COS = {"fp32": (1e-5, 1.0 - 1e-10), "bf16": (2e-2, 1.0 - 1.5e-4), "fp16": (3e-3, 1.0 - 2.2e-6)}

# (loss rel, output slice, per-tensor |g|_2 and sum|g| rel, gradient slices) -- all "of the scale": max|a - b| / max|b| for tensors
# Observed on MI355X (profiles/r02_full_grad_parity.txt): fp32 2e-7 / 2e-6 / 6e-7 / 2e-6; bf16 4e-5 / 1e-2 / 3e-3 / 1.2e-2; fp16 3e-6 / 1.4e-3 / 4e-4 / 2e-3
FULL_GRAD_TOL = {"fp32": (2e-6, 1e-4, 2e-5, 2e-5), "bf16": (3e-4, 3e-2, 1e-2, 4e-2), "fp16": (3e-5, 5e-3, 2e-3, 8e-3)}


@pytest.mark.parametrize("C", [52, 65])
def test_full_size_backward_vs_reference_gradients(golden_dir, C):
    """The backward the bench times, pinned to the imported reference (tests/golden/make_golden.py::full_net_gradients): default
    network, B = 2, C = 52 and C = 65 (the 65 -> 128 padding of the edge convs), injected (t, eps): loss, a strided output slice,
    the L2 norm and absolute sum of ALL 228 parameter gradients, strided slices of 16 representative weight gradients and their bias
    gradients in full.  fp32 mode <= 2e-4 of scale; bf16 and fp16 against the SAME reference numbers (not against this repo's fp32
    path).  The observed worst cases are written to gpurun_out/full_grad_parity.txt (README quotes them)."""
    g = _golden(golden_dir, f"full_net_grads_c{C}.npz")
    torch.manual_seed(0)
    net = ScoreUNet(channels=C, spatial=2, activation=torch.nn.SiLU, **DEFAULT).cuda()
    gen = torch.Generator().manual_seed(int(g["seed"]))
    x = torch.randn(2, C, 128, 128, generator=gen) * 0.5 + 0.5
    t = torch.rand(2, 1, 1, 1, generator=gen)
    eps = torch.randn(2, C, 128, 128, generator=gen)
    assert x.double().sum().item() == pytest.approx(float(g["x_checksum"]), rel=1e-12) and np.array_equal(t.numpy(), g["t"])
    xt = od.perturb(x, t, eps).cuda()
    epsd = eps.cuda()
    names = [str(n) for n in g["names"]]
    assert names == [n for n, _ in net.named_parameters()]
    report = []
    for mode, (tl, ty, tn, ts) in FULL_GRAD_TOL.items():
        net.precision = mode
        for p in net.parameters():
            p.grad = None
        S = 65536.0 if mode == "fp16" else 1.0  # GradScaler's initial scale
        y = net(xt, t.reshape(-1).cuda())
        loss = ((y - epsd) ** 2).mean()
        (loss * S).backward()
        torch.cuda.synchronize()
        grads = {n: p.grad.detach().double().cpu() / S for n, p in net.named_parameters()}
        e_loss = abs(loss.item() - float(g["loss"])) / float(g["loss"])
        e_y = _rel(y.detach()[:, :, ::16, ::16], torch.from_numpy(g["y_slice"]))
        e_norm = max((abs(grads[n].norm().item() - ref) / ref, n) for n, ref in zip(names, g["norm"]))
        e_abs = max((abs(grads[n].abs().sum().item() - ref) / ref, n) for n, ref in zip(names, g["abs_sum"]))
        e_slice, e_l2, e_cos, n_cmp = (0.0, ""), (0.0, ""), (2.0, ""), 0
        for k in g:
            if k.startswith("slice."):
                n = k[len("slice."):]
                s0, s1 = (int(v) for v in g["step." + n])
                got, ref = grads[n][::s0, ::s1], torch.from_numpy(g[k])
            elif k.startswith("full."):
                n = k[len("full."):]
                got, ref = grads[n], torch.from_numpy(g[k])
            else:
                continue
            e_slice = max(e_slice, (_rel(got, ref), n))
            rl2, cos = _rl2_cos(got, ref)
            e_l2, e_cos, n_cmp = max(e_l2, (rl2, n)), min(e_cos, (cos, n)), n_cmp + 1
        y_l2, y_cos = _rl2_cos(y.detach()[:, :, ::16, ::16], torch.from_numpy(g["y_slice"]))
        report.append(f"C={C} {mode}: loss {e_loss:.2e}  y {e_y:.2e}  |g|_2 {e_norm[0]:.2e} ({e_norm[1]})  sum|g| {e_abs[0]:.2e} ({e_abs[1]})  "
                      f"slices {e_slice[0]:.2e} ({e_slice[1]})  | over {n_cmp} tensors: rel-L2 {e_l2[0]:.2e} ({e_l2[1]})  "
                      f"1-cos {1.0 - e_cos[0]:.2e} ({e_cos[1]})  | y rel-L2 {y_l2:.2e} 1-cos {1.0 - y_cos:.2e}")
        if not (e_norm[0] <= tn and e_abs[0] <= tn and e_slice[0] <= ts and e_l2[0] <= FULL_GRAD_L2COS[mode][0]):
            # keep the evidence of a mismatch (one run in six of round 6's first GPU call failed here in fp16 and never again): the worst tensors
            os.makedirs("gpurun_out", exist_ok=True)
            torch.save({n: grads[n] for n in {e_norm[1], e_abs[1], e_slice[1], e_l2[1]} if n}, os.path.join("gpurun_out", f"parity_fail_c{C}_{mode}.pt"))
        assert e_loss <= tl and e_y <= ty, report[-1]
        assert e_norm[0] <= tn and e_abs[0] <= tn, report[-1]
        assert e_slice[0] <= ts, report[-1]
        assert n_cmp >= 150, n_cmp  # the round-6 fixtures: every bias gradient, every modulation projection
        assert e_l2[0] <= FULL_GRAD_L2COS[mode][0] and e_cos[0] >= FULL_GRAD_L2COS[mode][1], report[-1]
        assert y_l2 <= FULL_GRAD_L2COS[mode][0] and y_cos >= FULL_GRAD_L2COS[mode][1], report[-1]
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", "full_grad_parity.txt"), "a") as f:
        f.write("\n".join(report) + "\n")


def test_full_size_bf16_large_batch_matches_fp32_path():
    """B = 64 at C = 65 -- the bench's channel count -- as a SELF-comparison of the 16-bit modes with this repo's fp32 path (which is
    pinned to the reference by the fingerprint and gradient tests above).  At B = 64 the 128^2 and 64^2 levels and the 128-channel up-
    convs dispatch to the 16x16-tile kernel with the fused LayerNorm epilogues, and since round 6 (threshold 512 workgroups) so does the 32^2 level (8 x 64 = 512
    workgroups); the 16^2 level stays on the 8x16 tiles, 8x8 images are paired per tile.  The B = 128 dispatch of every level, against independent expected
    values (the PyTorch restatement per kernel, the CPU oracle for the whole step), is tests/test_gpu_bench_dispatch.py.
    Forward and parameter gradients of the modes on the same inputs: bf16 tolerance 3e-2 / 6e-2 of the scale, fp16 5e-3 / 1.5e-2."""
    B, C = 64, 65
    torch.manual_seed(0)
    net = ScoreUNet(channels=C, spatial=2, activation=torch.nn.SiLU, **DEFAULT).cuda()
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(B, C, 128, 128, generator=g) * 0.5 + 0.5).cuda()
    t = torch.rand(B, generator=g).cuda()
    eps = torch.randn(B, C, 128, 128, generator=g).cuda()
    outs, grads = {}, {}
    for mode in ("fp32", "bf16", "fp16"):
        net.precision = mode
        for p in net.parameters():
            p.grad = None
        S = 65536.0 if mode == "fp16" else 1.0  # GradScaler's initial scale
        y = net(x, t)
        loss = ((y - eps) ** 2).mean()
        (loss * S).backward()
        torch.cuda.synchronize()
        outs[mode] = y.detach().float().clone()
        grads[mode] = {n: p.grad.detach().clone() / S for n, p in net.named_parameters()}
    assert _rel(outs["bf16"], outs["fp32"]) <= 3e-2
    worst = max((_rel(grads["bf16"][n], grads["fp32"][n]), n) for n in grads["fp32"] if n.endswith("weight"))
    assert worst[0] <= 6e-2, worst  # gradients pass through ~100 bf16 layers: twice the forward tolerance
    assert _rel(outs["fp16"], outs["fp32"]) <= 5e-3
    worst = max((_rel(grads["fp16"][n], grads["fp32"][n]), n) for n in grads["fp32"] if n.endswith("weight"))
    assert worst[0] <= 1.5e-2, worst


@pytest.mark.parametrize("B,H,W,channels,cfg", [
    (1, 16, 32, 5, dict(embedding_dim=64, hidden_channels=[64, 128], hidden_blocks=[1, 2], attention_levels=[1])),
    (3, 48, 16, 9, dict(embedding_dim=96, hidden_channels=[128, 64, 192], hidden_blocks=[1, 1, 1], attention_levels=[])),
    (5, 8, 8, 65, dict(embedding_dim=64, hidden_channels=[64, 64], hidden_blocks=[2, 1], attention_levels=[0, 1])),
    (2, 32, 64, 13, dict(embedding_dim=64, hidden_channels=[128, 128, 128, 128], hidden_blocks=[1, 1, 1, 1], attention_levels=[3])),
])
def test_ragged_configurations_fp32_vs_oracle(B, H, W, channels, cfg):
    """Non-square images, odd batches, channel counts that need padding, attention at several levels and token counts (incl.
    T = 64 on a non-default width and T != 64), levels wider than their neighbours: forward and all parameter gradients of
    the fp32 mode against the CPU oracle (<= 1e-4 / 3e-4), bf16 forward within 3e-2."""
    cfg = dict(cfg, kernel_size=3, padding_mode="zeros")
    torch.manual_seed(B * 100 + H)
    net = ScoreUNet(channels=channels, spatial=2, activation=torch.nn.SiLU, **cfg)
    sdo = {k: v.clone().requires_grad_(True) for k, v in net.state_dict().items()}
    net = net.cuda()
    g = torch.Generator().manual_seed(H * W)
    x = torch.randn(B, channels, H, W, generator=g)
    t = torch.rand(B, generator=g)
    eps = torch.randn(B, channels, H, W, generator=g)
    yo = ou.score_unet_forward(sdo, x, t, cfg["hidden_blocks"], cfg["attention_levels"])
    go = torch.autograd.grad(((yo - eps) ** 2).mean(), list(sdo.values()))
    net.precision = "fp32"
    y = net(x.cuda(), t.cuda())
    assert _rel(y, yo.detach()) <= 1e-4
    ((y - eps.cuda()) ** 2).mean().backward()
    named = dict(net.named_parameters())
    for (k, _), gr in zip(sdo.items(), go):
        assert _rel(named[k].grad, gr) <= 3e-4, k
    net.precision = "bf16"
    with torch.no_grad():
        assert _rel(net(x.cuda(), t.cuda()), yo.detach()) <= 3e-2


def test_deep_variant_fp16_forward_vs_oracle_and_graph_replayed_score():
    """BASELINE.json configs[4]: the deep spatio-temporal variant -- 5 variables x 16 frames = 80 channels, 256x256 windows, fp16 MFMA,
    hipGraph-captured sampler step -- at its own size.  (a) forward of the default network at 80 ch x 256^2, B = 2 (the 16x16 bottleneck
    attends over T = 256 tokens: attn_mfma_fwd_blocks_kernel, online softmax over key blocks) against the CPU oracle: fp32 mode
    <= 1e-4, fp16 <= 5e-3, bf16 <= 3e-2 of scale.  (b) the sampler's score evaluation at k = 7 (window 15 -> 75 channels: the
    reference's windows are odd) over 256^2 frames, replayed from a hipGraph: bit-equal to the eager launches."""
    from climate2weather_amd.pipelines import SDAPipeline
    from climate2weather_amd.score_fn import BatchedScoreFunction
    torch.manual_seed(0)
    net = ScoreUNet(channels=80, spatial=2, activation=torch.nn.SiLU, **DEFAULT)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    net = net.cuda().eval()
    g = torch.Generator().manual_seed(80)
    x = torch.randn(2, 80, 256, 256, generator=g) * 0.5 + 0.5
    t = torch.rand(2, generator=g)
    with torch.no_grad():
        yo = ou.score_unet_forward(sd, x, t, DEFAULT["hidden_blocks"], DEFAULT["attention_levels"])
        for mode, tol in (("fp32", 1e-4), ("fp16", 5e-3), ("bf16", 3e-2)):
            net.precision = mode
            y = net(x.cuda(), t.cuda())
            assert _rel(y, yo) <= tol, (mode, _rel(y, yo))
    del net
    k, F, L = 7, 5, 19
    torch.manual_seed(1)
    net = ScoreUNet(channels=F * (2 * k + 1), spatial=2, activation=torch.nn.SiLU, **DEFAULT).cuda().eval()
    net.precision = "fp16"
    pipe = SDAPipeline()
    dev = torch.device("cuda", 0)
    noise = torch.randn(L, F, 256, 256, generator=torch.Generator().manual_seed(2))
    outs = []
    for graphs in (False, True):
        sf = BatchedScoreFunction(net, markov_order=k, batch_size=3, device=dev, noise_process=pipe)  # 5 windows: batches of 3 + 2
        sf.use_graphs = graphs
        outs.append(pipe.sample(sf, noise, steps=3, corrections=0, device=dev, show_progressbar=False))
        if graphs:
            assert len(sf._graphs) == 1 and len(next(iter(sf._graphs.values()))["graphs"]) == 2
    assert torch.isfinite(outs[0]).all() and torch.equal(outs[0], outs[1])


def test_relu_network_vs_reference_golden(golden_dir):
    """The reference UNet's own default activation (torch.nn.ReLU, model/nn.py:118) on the HIP path: fp32 forward / loss / every
    gradient against the imported reference (tests/golden/tiny_net_relu.npz), 16-bit forward within the mode tolerances."""
    g = _golden(golden_dir, "tiny_net.npz")
    r = _golden(golden_dir, "tiny_net_relu.npz")
    torch.manual_seed(3)
    net = ScoreUNet(channels=6, spatial=2, activation=torch.nn.ReLU, **TINY).cuda()
    net.precision = "fp32"
    x, t, eps = (torch.from_numpy(g[k]).cuda() for k in ("x", "t", "eps"))
    with torch.no_grad():
        y32 = net(torch.from_numpy(g["x32"]).cuda(), torch.tensor(0.3).cuda())
    assert _rel(y32, torch.from_numpy(r["y32"])) <= 1e-4
    loss = od.loss(net, x, t, eps).mean()
    assert loss.item() == pytest.approx(float(r["loss"]), rel=1e-4)
    loss.backward()
    for n, p in net.named_parameters():
        assert _rel(p.grad, torch.from_numpy(r["grad." + n])) <= 2e-4, n
    torch.manual_seed(3)
    big = ScoreUNet(channels=6, spatial=2, activation=torch.nn.ReLU, embedding_dim=64, hidden_channels=[64, 128], hidden_blocks=[1, 1],
                    attention_levels=[1], kernel_size=3, padding_mode="zeros")
    sd = {k: v.detach().clone() for k, v in big.state_dict().items()}
    big = big.cuda()
    xb = torch.randn(3, 6, 32, 32, generator=torch.Generator().manual_seed(1))
    tb = torch.tensor([0.2, 0.5, 0.9])
    yo = ou.score_unet_forward(sd, xb, tb, [1, 1], [1], act=torch.nn.functional.relu)
    for mode, tol in (("fp32", 1e-4), ("fp16", 5e-3), ("bf16", 3e-2)):
        big.precision = mode
        with torch.no_grad():
            assert _rel(big(xb.cuda(), tb.cuda()), yo) <= tol, mode
