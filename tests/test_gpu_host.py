"""GPU parity of the host-level paths that sit around the network: the training step (training_loop.py:369-391), the
sliding-window score functions and the predictor/corrector sampler (src/thor/score.py, src/thor/pipelines.py), and the
sampler's fused HIP update kernels (csrc/sampler.hip), against the golden vectors of the imported reference and the
CPU oracle.  fp32 mode, tolerances as stated per test."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import emu_ops
from climate2weather_amd import ops
from climate2weather_amd.ops import DTYPE_BF16, DTYPE_F16, DTYPE_F32
from climate2weather_amd.pipelines import SDAPipeline
from climate2weather_amd.score import ScoreUNet
from climate2weather_amd.score_fn import BatchedScoreFunction, DefaultScoreFunction, PoolStrideOperator
from climate2weather_amd.training import Trainer
from oracle import diffusion as od
from oracle import host as oh
from oracle import unet as ou

pytestmark = pytest.mark.gpu

TINY = dict(embedding_dim=64, hidden_channels=[32, 64], hidden_blocks=[1, 1], attention_levels=[1], kernel_size=3,
            padding_mode="zeros")
DEFAULT = dict(embedding_dim=512, hidden_blocks=[3] * 5, hidden_channels=[128, 128, 256, 384, 512], kernel_size=3,
               padding_mode="zeros", attention_levels=[4])
TD = {DTYPE_F32: torch.float32, DTYPE_BF16: torch.bfloat16, DTYPE_F16: torch.float16}


def _golden(golden_dir, name):
    return {k: v for k, v in np.load(os.path.join(golden_dir, name), allow_pickle=False).items()}


def _tiny(seed=3):
    torch.manual_seed(seed)
    net = ScoreUNet(channels=6, spatial=2, activation=torch.nn.SiLU, **TINY).cuda()
    net.precision = "fp32"
    return net


def _assert_adam_close(v, ref, grad, lr, name):
    big = grad.abs() > 1e-5  # see tests/test_host_emulated.py: g/(|g|+eps) is ill-conditioned where |g| ~ eps
    assert torch.allclose(v[big], ref[big], atol=3e-6), name
    assert (v - ref).abs().max().item() <= 2.0 * lr, name


# ------------------------------------------------------------------------------------------------ training step (a13)
def test_training_step_fp32_vs_golden(golden_dir):
    g = _golden(golden_dir, "tiny_net.npz")
    net = _tiny()
    tr = Trainer(net, lr=1e-3, precision="fp32", ema_rates=[0.9, 0.999])
    x, t, eps = (torch.from_numpy(g[k]).cuda() for k in ("x", "t", "eps"))
    loss = tr.step(x, t=t.reshape(-1), eps=eps)
    assert float(loss) == pytest.approx(float(g["loss"]), rel=1e-4)
    sd = {k: v.cpu() for k, v in net.state_dict().items()}
    for k in [str(n) for n in g["param_order"]]:
        p, gr = torch.from_numpy(g["sd." + k]), torch.from_numpy(g["grad." + k])
        exp = oh.adamw_step(p, gr, torch.zeros_like(p), torch.zeros_like(p), 1, 1e-3)[0]
        _assert_adam_close(sd[k], exp, gr, 1e-3, k)
    for rate, esd in tr.ema_state_dicts():
        for k, v in esd.items():
            ref = oh.ema_update(torch.from_numpy(g["sd." + k]), sd[k], rate)
            assert torch.allclose(v.cpu(), ref, atol=1e-6), k


def test_config0_ten_training_steps_follow_the_oracle():
    """BASELINE.json configs[0]: default net, 1 variable x window 13 = 13 channels, 32x32, batch 2, 10 optimizer steps.
    The HIP trainer and the CPU oracle (autograd + oracle.host.adamw_step) start from the same weights and see the same
    (x, t, eps) every step; fp32; per-step losses agree to 1e-3 relative (ten steps of accumulated fp32 reordering),
    and the weights after step 10 (each moved ~10 * lr = 1e-3) to 1e-4 max / 5e-6 mean absolute -- Adam's g/(sqrt(v)+eps)
    turns gradient noise on near-zero gradients into O(lr) differences, hence the two-level bound."""
    torch.manual_seed(0)
    net = ScoreUNet(channels=13, spatial=2, activation=torch.nn.SiLU, **DEFAULT)
    ref = ou.OracleScoreUNet(net.state_dict(), DEFAULT["hidden_blocks"], DEFAULT["attention_levels"])
    net = net.cuda()
    net.precision = "fp32"
    lr, steps = 1e-4, 10
    tr = Trainer(net, lr=lr, precision="fp32", ema_rates=[0.9999])
    m = [torch.zeros_like(p) for p in ref.params]
    v = [torch.zeros_like(p) for p in ref.params]
    gen = torch.Generator().manual_seed(7)
    for s in range(1, steps + 1):
        x = torch.randn(2, 13, 32, 32, generator=gen) * 0.5 + 0.5
        t = torch.rand(2, generator=gen)
        eps = torch.randn(2, 13, 32, 32, generator=gen)
        got = float(tr.step(x.cuda(), t=t.cuda(), eps=eps.cuda()))
        for p in ref.parameters():
            p.grad = None
        loss = od.loss(ref, x, t.view(-1, 1, 1, 1), eps).mean()
        loss.backward()
        with torch.no_grad():
            for i, p in enumerate(ref.params):
                newp, m[i], v[i] = oh.adamw_step(p.detach(), p.grad, m[i], v[i], s, lr)
                p.copy_(newp)
        assert got == pytest.approx(loss.item(), rel=1e-3), s
    sd = net.state_dict()
    for n, p in ref.sd().items():
        d = (sd[n].cpu() - p.detach()).abs()
        assert d.max().item() <= 1e-4 and d.mean().item() <= 5e-6, (n, d.max().item(), d.mean().item())


# ------------------------------------------------------------------------------------------------ sampler (a10 - a12)
@pytest.mark.parametrize("name,corrections,cond,exact,op", [
    ("uncond_c0", 0, False, False, None), ("uncond_c1", 1, False, False, None),
    ("cond_c0", 0, True, False, "pool"), ("cond_c0", 0, True, False, "generic"), ("cond_c1_exact", 1, True, True, "pool")])
@pytest.mark.parametrize("fused", [True, False])
def test_sampler_trajectories_vs_golden(golden_dir, name, corrections, cond, exact, op, fused):
    """Trajectory end points of the imported reference (tiny net, L=9, F=2, k=1, 32x32, 4 steps, recorded corrector draws).
    fused=True: state stays in HBM, csrc/sampler.hip update kernels; fused=False: the reference's torch update rule with
    the state on the host and window batches going through the HIP network."""
    s = _golden(golden_dir, "sampler.npz")
    net = _tiny().eval()
    pipe = SDAPipeline()
    dev = torch.device("cuda", 0)
    sf = BatchedScoreFunction(net, markov_order=1, batch_size=4, device=dev, noise_process=pipe)
    if cond:
        A = PoolStrideOperator(8, 2) if op == "pool" else (lambda z: F.avg_pool2d(z[::2], 8))
        sf.condition_on(A=A, y=torch.from_numpy(s["y_obs"]), std=torch.from_numpy(s["std"]), gamma=float(s["gamma"]),
                        exact_grad=exact)
    sf.device_resident = fused
    zs = [torch.from_numpy(z) for z in s[name + ".z"]] if corrections else None
    xs = pipe.sample(sf, torch.from_numpy(s[name + ".noise"]), steps=4, corrections=corrections, tau=0.5,
                     device=dev if fused else torch.device("cpu"), show_progressbar=False, z_draws=zs)
    ref = torch.from_numpy(s[name + ".x"])
    assert (xs.cpu() - ref).abs().max().item() <= 3e-4 * ref.abs().max().item()


@pytest.mark.parametrize("name,corrections,exact,op", [("cond_c0_gvec", 0, False, "pool"), ("cond_c0_gvec", 0, False, "generic"),
                                                       ("cond_c1_gvec_exact", 1, True, "pool")])
@pytest.mark.parametrize("fused", [True, False])
def test_sampler_trajectories_with_per_variable_gamma(golden_dir, name, corrections, exact, op, fused):
    """condition_on(gamma=<(1, F, 1, 1) tensor>), the form exp/downscaling.py:228-233 builds for a list-valued likelihood_gamma
    (src/thor/score.py:55 broadcasts it), with the tensor handed over on the HOST as the reference driver does: fused guidance
    kernel (per-variable gamma vector), generic-operator autograd route, exact gradient -- against the imported reference's
    trajectories (tests/golden/sampler_gamma.npz)."""
    s, sg = _golden(golden_dir, "sampler.npz"), _golden(golden_dir, "sampler_gamma.npz")
    net = _tiny().eval()
    pipe = SDAPipeline()
    dev = torch.device("cuda", 0)
    sf = BatchedScoreFunction(net, markov_order=1, batch_size=4, device=dev, noise_process=pipe)
    A = PoolStrideOperator(8, 2) if op == "pool" else (lambda z: F.avg_pool2d(z[::2], 8))
    gamma = torch.from_numpy(sg["gamma"])
    assert gamma.shape == (1, 2, 1, 1) and not gamma.is_cuda
    sf.condition_on(A=A, y=torch.from_numpy(s["y_obs"]), std=torch.from_numpy(s["std"]), gamma=gamma, exact_grad=exact)
    assert (sf._fused_guidance is not None) == (op == "pool" and not exact)
    sf.device_resident = fused
    zs = [torch.from_numpy(z) for z in sg[name + ".z"]] if corrections else None
    xs = pipe.sample(sf, torch.from_numpy(s["cond_c0.noise"]), steps=4, corrections=corrections, tau=0.5,
                     device=dev if fused else torch.device("cpu"), show_progressbar=False, z_draws=zs)
    ref = torch.from_numpy(sg[name + ".x"])
    assert (xs.cpu() - ref).abs().max().item() <= 3e-4 * ref.abs().max().item()
    # and it is not the scalar-gamma trajectory
    assert (ref - torch.from_numpy(s["cond_c0.x"])).abs().max().item() > 1e-2 * ref.abs().max().item()


def test_score_functions_vs_golden(golden_dir):
    s = _golden(golden_dir, "sampler.npz")
    net = _tiny().eval()
    pipe = SDAPipeline()
    x = torch.from_numpy(s["score_x"])
    for sf in (DefaultScoreFunction(net, markov_order=1, noise_process=pipe),
               BatchedScoreFunction(net, markov_order=1, batch_size=4, device=torch.device("cuda", 0), noise_process=pipe)):
        with torch.no_grad():
            y = sf(x.cuda(), torch.tensor(0.7))
        assert torch.allclose(y.cpu(), torch.from_numpy(s["score_y"]), atol=2e-5)


def test_nan_in_state_raises_on_the_device_path():
    net = _tiny().eval()
    pipe = SDAPipeline()
    sf = BatchedScoreFunction(net, markov_order=1, batch_size=4, device=torch.device("cuda", 0), noise_process=pipe)
    noise = torch.randn(5, 2, 16, 16)
    noise[0, 0, 0, 0] = float("nan")
    with pytest.raises(ValueError, match="NaN detected"):
        pipe.sample(sf, noise, steps=2, show_progressbar=False)
    # ... and it raises where the reference does (src/thor/pipelines.py:90-91: in the step that made the NaN), one step late, not at the
    # end of the trajectory: the flag is published into pinned host memory behind every step and read one step later, no stream drained
    calls = []

    class Counting:
        device_resident, device = True, sf.device

        def __call__(self, x, t):
            calls.append(float(t))
            return sf(x, t)
    for corrections in (0, 1):
        calls.clear()
        with pytest.raises(ValueError, match="NaN detected"):
            pipe.sample(Counting(), noise, steps=64, corrections=corrections, show_progressbar=False)
        assert len(calls) <= 2 * (1 + corrections), f"raised after {len(calls)} score evaluations of a 64-step run"
    # a clean run publishes 64 flags and raises nothing
    calls.clear()
    out = pipe.sample(Counting(), torch.randn(5, 2, 16, 16), steps=64, show_progressbar=False)
    assert len(calls) == 64 and bool(torch.isfinite(out).all())


# ------------------------------------------------------------------------------------------------ csrc/sampler.hip kernels
@pytest.mark.parametrize("dtype", [DTYPE_F32, DTYPE_BF16, DTYPE_F16])
@pytest.mark.parametrize("L,Fv,k,H,i0,nw", [(9, 2, 1, 16, 0, 7), (9, 2, 1, 16, 3, 4), (20, 4, 6, 32, 0, 8), (20, 4, 6, 32, 5, 3),
                                            (14, 5, 6, 8, 0, 2), (3, 1, 1, 8, 0, 1)])
def test_window_gather_scatter_vs_emulation(dtype, L, Fv, k, H, i0, nw):
    """unfold (src/thor/score.py:122-133) straight into padded NHWC rows, and fold (:135-154) straight out of them."""
    w = 2 * k + 1
    HW = H * H
    ldc = (w * Fv + 7) // 8 * 8
    nwin = L - w + 1
    gen = torch.Generator().manual_seed(L * 100 + i0)
    x = torch.randn(L, Fv, H, H, generator=gen)
    y_ref = torch.full((nw * HW, ldc), 7.0, dtype=TD[dtype])
    emu_ops.window_gather(x, y_ref, nw, Fv, HW, k, i0, ldc, dtype)
    y = torch.full((nw * HW, ldc), 7.0, dtype=TD[dtype], device="cuda")
    ops.window_gather(x.cuda(), y, nw, Fv, HW, k, i0, ldc, dtype)
    assert torch.equal(y.cpu(), y_ref)
    net_out = torch.randn(nw * HW, ldc, generator=gen).to(TD[dtype])
    e_ref = torch.full((L, Fv, H, H), -3.0)
    emu_ops.window_scatter(net_out, e_ref, nw, Fv, HW, k, i0, nwin, ldc, dtype)
    e = torch.full((L, Fv, H, H), -3.0, device="cuda")
    ops.window_scatter(net_out.cuda(), e, nw, Fv, HW, k, i0, nwin, ldc, dtype)
    assert torch.equal(e.cpu(), e_ref)


def test_unfold_fold_round_trip_at_full_size():
    """Size-independent property at the sampler's shipped shape (L=49, F=4, k=6, 128x128): scattering the gathered
    windows back (fold o unfold) reproduces the trajectory exactly, whatever the window batching."""
    L, Fv, k, H = 49, 4, 6, 128
    w, HW = 2 * k + 1, H * H
    ldc = (w * Fv + 7) // 8 * 8
    nwin = L - w + 1
    x = torch.randn(L, Fv, H, H, device="cuda")
    out = torch.zeros_like(x)
    for i0 in range(0, nwin, 16):
        nw = min(16, nwin - i0)
        y = torch.empty(nw * HW, ldc, device="cuda")
        ops.window_gather(x, y, nw, Fv, HW, k, i0, ldc, DTYPE_F32)
        ops.window_scatter(y, out, nw, Fv, HW, k, i0, nwin, ldc, DTYPE_F32)
    assert torch.equal(out, x)


@pytest.mark.parametrize("n", [1, 1000, 9 * 2 * 32 * 32, 49 * 4 * 128 * 128 + 3])
def test_predict_correct_sumsq_vs_emulation(n):
    gen = torch.Generator().manual_seed(n)
    x, eps, z = (torch.randn(n, generator=gen) for _ in range(3))
    a, b, tau, sg = 1.0123, -0.0456, 0.5, 0.8
    xr = x.clone()
    emu_ops.sampler_predict(xr, eps, None, n, a, b)
    xg, flag = x.cuda(), torch.zeros(1, dtype=torch.int32, device="cuda")
    ops.sampler_predict(xg, eps.cuda(), flag, n, a, b)
    assert torch.allclose(xg.cpu(), xr, rtol=1e-6, atol=1e-6) and int(flag) == 0
    ss_ref = torch.zeros(1, dtype=torch.float64)
    ss_ref += eps.double().square().sum()
    ss = torch.zeros(1, device="cuda")
    ops.sumsq(eps.cuda(), ss, n)
    assert float(ss) == pytest.approx(float(ss_ref), rel=1e-5)
    ssr = ss.cpu().clone()
    emu_ops.sampler_correct(xr, eps, z, ssr, None, n, tau, sg)
    ops.sampler_correct(xg, eps.cuda(), z.cuda(), ss, flag, n, tau, sg)
    assert torch.allclose(xg.cpu(), xr, rtol=1e-5, atol=1e-5) and int(flag) == 0
    # non-finite state raises the flag (src/thor/pipelines.py:90-91)
    xg[n // 2] = float("inf")
    ops.sampler_predict(xg, eps.cuda(), flag, n, a, b)
    assert int(flag) != 0


@pytest.mark.parametrize("L,Fv,H,s_step,t_step", [(9, 2, 32, 8, 2), (13, 4, 128, 16, 6), (12, 4, 64, 16, 6), (5, 1, 16, 4, 1)])
def test_guidance_kernel_vs_emulation_and_autograd(L, Fv, H, s_step, t_step):
    """exact_grad=False guidance (src/thor/score.py:24-42) for A = AvgPool2d(s) o [::t] (exp/downscaling.py:129-132):
    eps <- eps - sigma * d/dx log p(y|x), closed form in one kernel; checked against the emulation and against
    torch.autograd of the log-likelihood itself."""
    gen = torch.Generator().manual_seed(L)
    nobs = (L + t_step - 1) // t_step
    x = torch.randn(L, Fv, H, H, generator=gen)
    eps = torch.randn(L, Fv, H, H, generator=gen)
    yobs = torch.randn(nobs, Fv, H // s_step, H // s_step, generator=gen)
    std = torch.rand(Fv, generator=gen) * 0.5 + 0.2
    mu, sigma, gamma = 0.6, 0.8, 1e-2
    e_ref = eps.clone()
    emu_ops.guidance(x, e_ref, yobs, std, nobs, Fv, H, H, s_step, t_step, mu, sigma, gamma)
    e = eps.cuda()
    ops.guidance(x.cuda(), e, yobs.cuda(), std.cuda(), nobs, Fv, H, H, s_step, t_step, mu, sigma, gamma)
    assert torch.allclose(e.cpu(), e_ref, rtol=1e-5, atol=1e-5)
    xa = x.clone().requires_grad_(True)
    x0 = (xa - sigma * eps) / mu
    err = yobs - F.avg_pool2d(x0[::t_step], s_step)
    var = std.view(1, Fv, 1, 1) ** 2 + gamma * (sigma / mu) ** 2
    logp = -(err ** 2 / var).sum() / 2
    (J,) = torch.autograd.grad(logp, xa)
    assert torch.allclose(e.cpu(), eps - sigma * J, rtol=1e-4, atol=1e-5)
    # one gamma per variable (exp/downscaling.py:228-233): c2w_guidance_per_variable
    gvec = torch.rand(Fv, generator=gen) * 0.2 + 1e-3
    e2 = eps.cuda()
    ops.guidance(x.cuda(), e2, yobs.cuda(), std.cuda(), nobs, Fv, H, H, s_step, t_step, mu, sigma, gvec.cuda())
    var = std.view(1, Fv, 1, 1) ** 2 + gvec.view(1, Fv, 1, 1) * (sigma / mu) ** 2
    xa = x.clone().requires_grad_(True)
    logp = -((yobs - F.avg_pool2d(((xa - sigma * eps) / mu)[::t_step], s_step)) ** 2 / var).sum() / 2
    (J,) = torch.autograd.grad(logp, xa)
    assert torch.allclose(e2.cpu(), eps - sigma * J, rtol=1e-4, atol=1e-5)
    e3 = eps.cuda()  # a constant vector == the scalar entry point, bit for bit
    ops.guidance(x.cuda(), e3, yobs.cuda(), std.cuda(), nobs, Fv, H, H, s_step, t_step, mu, sigma, torch.full((Fv,), gamma).cuda())
    assert torch.equal(e3, e)


# ------------------------------------------------------------------------------------------------ time-sharded sampler (f1)
def test_time_sharded_sampler_single_rank_and_local_guidance(golden_dir):
    """One rank: the sharded driver reproduces the golden trajectories on the HIP kernels.  Two-rank halo exchange runs in
    the CPU (gloo) suite; here the per-rank guidance slice (global observation index / first-observed-frame offset) is
    checked against the guidance of the whole trajectory."""
    from climate2weather_amd.sharded import TimeShardedScoreFunction, sample_time_sharded
    s = _golden(golden_dir, "sampler.npz")
    net = _tiny().eval()
    pipe = SDAPipeline()
    dev = torch.device("cuda", 0)
    for name, corrections, cond in [("uncond_c1", 1, False), ("cond_c0", 0, True)]:
        sf = TimeShardedScoreFunction(net, markov_order=1, length=9, batch_size=4, device=dev, noise_process=pipe, rank=0, world=1)
        if cond:
            sf.condition_on(A=PoolStrideOperator(8, 2), y=torch.from_numpy(s["y_obs"]), std=torch.from_numpy(s["std"]), gamma=float(s["gamma"]))
        zs = [torch.from_numpy(z) for z in s[name + ".z"]] if corrections else None
        x = sample_time_sharded(pipe, sf, torch.from_numpy(s[name + ".noise"]), steps=4, corrections=corrections, tau=0.5, z_draws=zs)
        ref = torch.from_numpy(s[name + ".x"])
        assert (x.cpu() - ref).abs().max().item() <= 3e-4 * ref.abs().max().item(), name
    # guidance on the frames [5, 9) of rank 1 of 2 == the same frames of the whole-trajectory guidance
    L, Fv, H = 9, 2, 32
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(L, Fv, H, H, generator=gen).cuda()
    eps = torch.randn(L, Fv, H, H, generator=gen).cuda()
    y = torch.from_numpy(s["y_obs"]).cuda()
    std = torch.from_numpy(s["std"]).reshape(-1).cuda()
    full = eps.clone()
    mu, sigma = pipe._mu_sigma_f(0.7)
    ops.guidance(x, full, y, std, y.shape[0], Fv, H, H, 8, 2, mu, sigma, float(s["gamma"]))
    sf1 = TimeShardedScoreFunction(net, markov_order=1, length=L, batch_size=4, device=dev, noise_process=pipe, rank=1, world=2)
    sf1.condition_on(A=PoolStrideOperator(8, 2), y=y, std=torch.from_numpy(s["std"]).cuda(), gamma=float(s["gamma"]))  # (1, F, 1, 1)
    with pytest.raises(NotImplementedError):  # a 1-D (F,) std would broadcast over the LAST axis in the reference: not per variable
        sf1.condition_on(A=PoolStrideOperator(8, 2), y=y, std=std, gamma=float(s["gamma"]))
    lo, hi = sf1.bounds[1]
    part = eps[lo:hi].clone()
    sf1._apply_guidance(x[lo:hi].contiguous(), part, 0.7)
    assert torch.equal(part, full[lo:hi])


# ------------------------------------------------------------------------------------------------ f4: operator + normalisation
@pytest.mark.parametrize("L,Fv,H,s_step,t_step", [(9, 2, 32, 8, 2), (49, 4, 128, 16, 6), (13, 4, 64, 16, 6), (5, 1, 16, 4, 1)])
def test_measurement_operator_kernel(L, Fv, H, s_step, t_step):
    gen = torch.Generator().manual_seed(L)
    x = torch.randn(L, Fv, H, H, generator=gen)
    y = PoolStrideOperator(s_step, t_step)(x.cuda())  # HIP path (no grad, fp32, cuda)
    assert torch.allclose(y.cpu(), oh.measure(x, s_step, t_step), rtol=1e-5, atol=1e-6)


def test_quantile_normalizer_kernel_and_round_trip_at_full_size():
    from climate2weather_amd.normalize import QuantileNormalizer
    q = {0.0: [-5.0, 0.0, 1.0, -2.0], 0.01: [-4.0, 0.5, 1.5, -1.5], 0.05: [2.0, 1.0, 2.0, -1.0], 0.25: [4.0, 2.0, 3.0, 0.0],
         0.5: [6.0, 3.0, 4.0, 1.0], 0.75: [9.0, 5.0, 6.0, 2.5], 0.95: [12.0, 9.0, 8.0, 4.0], 0.99: [16.0, 10.5, 9.0, 5.0],
         1.0: [20.0, 12.0, 11.0, 7.0]}
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(49, 4, 128, 128, generator=gen) * 3 + 4
    xg = x.cuda()
    for mode in ("minmax", "robust", "robust95", "quant95", "quant99"):
        qn = QuantileNormalizer(q, mode)
        y = qn.normalize(xg)
        assert torch.allclose(y[:5].cpu(), oh.normalize(x[:5], q, mode), rtol=1e-5, atol=1e-5), mode
        assert torch.allclose(qn.unnormalize(xg)[:5].cpu(), oh.unnormalize(x[:5], q, mode), rtol=1e-5, atol=1e-5), mode
        assert torch.allclose(qn.unnormalize(y), xg, rtol=1e-5, atol=1e-4), mode  # size-independent round trip, full sampler shape


def test_graph_replayed_score_function_equals_eager(golden_dir):
    """hipGraph capture of the score evaluation (window gather -> network -> fold), replayed every sampler step: identical
    trajectories to the eager launches and to the golden end point."""
    s = _golden(golden_dir, "sampler.npz")
    net = _tiny().eval()
    pipe = SDAPipeline()
    dev = torch.device("cuda", 0)
    outs = []
    for graphs in (False, True):
        sf = BatchedScoreFunction(net, markov_order=1, batch_size=4, device=dev, noise_process=pipe)  # 7 windows: batches of 4 + 3
        sf.condition_on(A=PoolStrideOperator(8, 2), y=torch.from_numpy(s["y_obs"]), std=torch.from_numpy(s["std"]), gamma=float(s["gamma"]),
                        exact_grad=False)
        sf.use_graphs = graphs
        outs.append(pipe.sample(sf, torch.from_numpy(s["cond_c0.noise"]), steps=4, corrections=0, tau=0.5, device=dev, show_progressbar=False))
        if graphs:
            assert len(sf._graphs) == 1 and len(next(iter(sf._graphs.values()))["graphs"]) == 2
            # a second trajectory (another tensor, another address: the next ensemble member) replays the SAME captures
            first = next(iter(sf._graphs.values()))["graphs"]
            keep = torch.empty(1 << 20, device=dev)  # shifts the allocator so that the new trajectory cannot land on the old address
            again = pipe.sample(sf, torch.from_numpy(s["cond_c0.noise"]).clone(), steps=4, corrections=0, tau=0.5, device=dev, show_progressbar=False)
            assert next(iter(sf._graphs.values()))["graphs"] is first and len(sf._graphs) == 1
            assert torch.equal(again, outs[1])
            del keep
    assert torch.equal(outs[0], outs[1])
    ref = torch.from_numpy(s["cond_c0.x"])
    assert (outs[1].cpu() - ref).abs().max().item() <= 3e-4 * ref.abs().max().item()


@pytest.mark.parametrize("prec", ["bf16", "fp16"])
def test_graph_replayed_score_function_sees_weights_written_through_the_parameters(prec):
    """Round-2 advisor finding: replay never re-enters Engine.forward(), so the graph cache has to look at the Parameters' version
    counters itself.  Capture in a 16-bit mode (the captured launches read the 16-bit weight SHADOW), then change the weights the way
    callers do -- load_state_dict, an in-place optimizer-style update, an EMA copy_ -- and evaluate again with graphs on: each time
    the result must equal a fresh eager evaluation of the new weights (stale captures would return the old weights' score)."""
    dev = torch.device("cuda", 0)
    torch.manual_seed(11)
    cfg = dict(embedding_dim=64, hidden_channels=[64, 128], hidden_blocks=[1, 1], attention_levels=[1], kernel_size=3, padding_mode="zeros")
    net = ScoreUNet(channels=6, spatial=2, activation=torch.nn.SiLU, **cfg).to(dev).eval()
    net.precision = prec
    pipe = SDAPipeline()
    x = torch.randn(9, 2, 32, 32, device=dev)
    t = torch.tensor(0.6)

    def both():
        outs = []
        for graphs in (True, False):
            sf_ = sf if graphs else BatchedScoreFunction(net, markov_order=1, batch_size=4, device=dev, noise_process=pipe)
            with torch.no_grad():
                outs.append(sf_(x, t).clone())
        return outs

    sf = BatchedScoreFunction(net, markov_order=1, batch_size=4, device=dev, noise_process=pipe)
    sf.use_graphs = True
    g0, e0 = both()
    assert torch.equal(g0, e0) and len(sf._graphs) == 1
    # 1. load_state_dict
    sd = {k: v + 0.05 * torch.randn_like(v) for k, v in net.state_dict().items()}
    net.load_state_dict(sd)
    g1, e1 = both()
    assert torch.equal(g1, e1) and not torch.equal(g1, g0)
    # 2. in-place update through the Parameter objects (what torch.optim does)
    with torch.no_grad():
        for p in net.parameters():
            p.add_(0.02 * torch.randn_like(p))
    g2, e2 = both()
    assert torch.equal(g2, e2) and not torch.equal(g2, g1)
    # 3. copy_ into the parameters (EMA -> net)
    with torch.no_grad():
        for p in net.parameters():
            p.copy_(p * 0.9)
    g3, e3 = both()
    assert torch.equal(g3, e3) and not torch.equal(g3, g2)
    # unchanged weights: the captures are reused
    first = next(iter(sf._graphs.values()))["graphs"]
    g4, _ = both()
    assert next(iter(sf._graphs.values()))["graphs"] is first and torch.equal(g4, g3)


def test_bf16_and_fp16_training_track_fp32_training():
    """60 optimizer steps on a small network and a fixed synthetic batch stream: the bf16 throughput mode and the fp16 mode (the
    reference's own "16-mixed", under the device-resident dynamic loss scale) must learn like the fp32 parity mode (same data,
    noise and times injected): losses fall, and the curves stay close to the fp32 one -- fp16 within 2 % (observed +0.2 %, the same
    from run to run), bf16 within 10 %: 60 steps of lr 2e-3 amplify the order of the fp32 atomics behind 8-bit significands, and
    the tail of the bf16 curve lands anywhere between -0.1 % and +4.6 % of the fp32 one from run to run
    (lab/probes/flake_training_curves.py; a 5 % bound failed once in ten full-suite runs)."""
    cfg = dict(embedding_dim=64, hidden_channels=[64, 128], hidden_blocks=[1, 1], attention_levels=[1], kernel_size=3, padding_mode="zeros")
    curves = {}
    for prec in ("fp32", "bf16", "fp16"):
        torch.manual_seed(11)
        net = ScoreUNet(channels=6, spatial=2, activation=torch.nn.SiLU, **cfg).cuda()
        tr = Trainer(net, lr=2e-3, precision=prec, ema_rates=[0.999])
        gen = torch.Generator().manual_seed(3)
        base = torch.randn(8, 6, 32, 32, generator=gen) * 0.5 + 0.5
        losses = []
        for s in range(60):
            x = (base + 0.05 * torch.randn(8, 6, 32, 32, generator=gen)).cuda()
            t = torch.rand(8, generator=gen).cuda()
            eps = torch.randn(8, 6, 32, 32, generator=gen).cuda()
            losses.append(float(tr.step(x, t=t, eps=eps)))
        curves[prec] = losses
        if prec == "fp16":  # at most a few early steps were skipped while the scale settled; it never collapsed
            assert tr.optimizer_steps_taken() >= 50 and tr.loss_scale() >= 1.0, (tr.optimizer_steps_taken(), tr.loss_scale())
    first = sum(curves["fp32"][:5]) / 5
    for prec in curves:
        last = sum(curves[prec][-10:]) / 10
        assert last < 0.6 * first, (prec, first, last)
    a = torch.tensor(curves["fp32"][-20:]).mean().item()
    for prec, bound in (("bf16", 0.10), ("fp16", 0.02)):
        b = torch.tensor(curves[prec][-20:]).mean().item()
        _MARGINS[f"curve_{prec}"] = abs(a - b) / a
        assert abs(a - b) <= bound * a, (prec, a, b)


_MARGINS = {}  # measured distances the tests above were GIVEN ROOM for; the margin tests below hold them against the tight bounds


@pytest.mark.xfail(strict=False, reason="margin watch, not a gate: the bf16 curve's tail lands -0.1 ... +4.6 % off the fp32 one from run to run "
                                        "(fp32 atomics order amplified over 60 steps; lab/probes/flake_training_curves.py); the gate is 10 %")
def test_margin_bf16_training_curve_within_5_percent():
    """The bound test_bf16_and_fp16_training_track_fp32_training had before round 4 widened it (5 % -> 10 %), kept in the suite so that a
    drift of the margin shows up as XFAIL counts in the record instead of a silently wider tolerance."""
    if "curve_bf16" not in _MARGINS:
        pytest.skip("the curve test did not run")
    assert _MARGINS["curve_bf16"] <= 0.05, _MARGINS["curve_bf16"]


def test_training_step_with_regenerated_noise_equals_the_step_on_the_written_noise_tensor():
    """Trainer(fused_noise=True) (the default) never materialises eps: the input conversion and the loss regenerate the Philox stream of
    the step's seed.  The same step on the tensor c2w_philox_normal writes for that seed gives the same parameters, EMA and loss."""
    cfg = dict(embedding_dim=64, hidden_channels=[64, 128], hidden_blocks=[1, 1], attention_levels=[1], kernel_size=3, padding_mode="zeros")
    nets = []
    for _ in range(2):
        torch.manual_seed(21)
        nets.append(ScoreUNet(channels=6, spatial=2, activation=torch.nn.SiLU, **cfg).cuda())
    a = Trainer(nets[0], lr=1e-3, precision="bf16", ema_rates=[0.99])
    b = Trainer(nets[1], lr=1e-3, precision="bf16", ema_rates=[0.99], fused_noise=False)
    g = torch.Generator().manual_seed(4)
    x = (torch.randn(4, 6, 32, 32, generator=g) * 0.5 + 0.5).cuda()
    t = torch.rand(4, generator=g).cuda()
    la = a.step(x, t=t)
    assert a.last_noise_seed is not None and b.last_noise_seed is None
    eps = torch.empty_like(x)
    ops.philox_normal(eps, eps.numel(), a.last_noise_seed)
    lb = b.step(x, t=t, eps=eps)
    assert abs(float(la) - float(lb)) <= 1e-5 * abs(float(lb))
    # Same arithmetic on both sides; what may differ is the summation ORDER of fp32 atomics (bias / modulation gradients), and Adam's
    # first step is lr * g / (|g| + eps): where |g| ~ eps a last-bit difference moves the parameter by up to 2 lr.  So: tight where
    # the gradient is well above eps, bounded by 2 lr everywhere (the rule of _assert_adam_close).
    big = b.eng.flat_grad.abs() > 1e-5
    d = (a.eng.flat - b.eng.flat).abs()
    assert d[big].max().item() <= 2e-6 and d.max().item() <= 2.0 * 1e-3
    assert (a.eng.flat_grad - b.eng.flat_grad).abs().max().item() <= 1e-5 * b.eng.flat_grad.abs().max().item()
    de = (a.ema_flats[0] - b.ema_flats[0]).abs()
    assert de[big].max().item() <= 2e-6 and de.max().item() <= 2.0 * 1e-3 * 0.01 + 1e-6
    s1 = a.last_noise_seed
    a.step(x, t=t)
    assert a.last_noise_seed != s1  # a fresh stream every step


def test_two_stream_backward_equals_single_stream(monkeypatch):
    """engine.grad_stream: weight gradients (+ their split-K reductions) on a second HIP stream, ordered by events, operands
    record_stream'ed.  Gradients of four steps (lr = 0: fixed weights, fresh data) at a size where the kernels run long enough to
    overlap (C2W_WGRAD_STREAM=1), against the same steps with everything on one stream (the default): equal up to the order of the
    modulation-gradient atomics (a race would show as garbage in some layer's gradient)."""
    cfg = dict(embedding_dim=128, hidden_channels=[128, 128, 256], hidden_blocks=[2, 1, 1], attention_levels=[2], kernel_size=3, padding_mode="zeros")
    grads, losses = [], []
    for single in ("0", "1"):  # "0" is the default since round 5 (grouped launches fill the chip on one stream); "1": rounds 1-4
        monkeypatch.setenv("C2W_WGRAD_STREAM", single)
        torch.manual_seed(8)
        net = ScoreUNet(channels=13, spatial=2, activation=torch.nn.SiLU, **cfg).cuda()
        tr = Trainer(net, lr=0.0, weight_decay=0.0, precision="bf16", ema_rates=[], fused_noise=False)
        assert (tr.eng.grad_stream() is None) == (single == "0")
        g = torch.Generator().manual_seed(2)
        gs, ls = [], []
        for _ in range(4):
            x = (torch.randn(16, 13, 64, 64, generator=g) * 0.5 + 0.5).cuda()
            t, eps = torch.rand(16, generator=g).cuda(), torch.randn(16, 13, 64, 64, generator=g).cuda()
            ls.append(float(tr.step(x, t=t, eps=eps)))
            gs.append(tr.eng.flat_grad.clone())
        grads.append(gs)
        losses.append(ls)
    assert max(abs(p - q) for p, q in zip(*losses)) <= 1e-5  # the loss sum itself is reduced with atomics
    for a, b in zip(*grads):
        assert torch.isfinite(a).all() and (a - b).abs().max().item() <= 2e-5 * a.abs().max().item()


def test_window_batches_on_several_streams_equal_one_stream(monkeypatch):
    """A score evaluation whose window batches alternate between HIP streams (score_fn.num_streams) is the same arithmetic as on one
    stream: bit-identical eps for 3 co-sampled members in 5 batches, repeated to give a race the chance to show."""
    torch.manual_seed(3)
    net = ScoreUNet(channels=6, spatial=2, activation=torch.nn.SiLU, embedding_dim=64, hidden_channels=[64, 128], hidden_blocks=[1, 1],
                    attention_levels=[1], kernel_size=3, padding_mode="zeros").cuda().eval()
    net.precision = "bf16"
    sf = BatchedScoreFunction(net, markov_order=1, batch_size=8, device=torch.device("cuda", 0), noise_process=SDAPipeline())
    x = torch.randn(3, 14, 2, 64, 64, device="cuda")  # 3 members x 12 windows = 36 windows -> 5 balanced batches
    t = torch.tensor(0.4)
    with torch.no_grad():
        sf.num_streams = 1
        ref = sf.score_fn(x, t).clone()
        sf.num_streams = 4
        for _ in range(6):
            out = sf.score_fn(x, t)
            torch.cuda.synchronize()
            assert torch.equal(out, ref)
    assert len(sf._side_streams(5)) == 4


@pytest.mark.parametrize("prec,tol", [("bf16", 2e-2), ("fp16", 3e-3)])
def test_output_convolution_folds_inside_the_engine_on_the_gpu(prec, tol):
    """engine.py::_fold_output with the HIP kernel (c2w_conv_center): the sampler's network calls compute only the frames fold() keeps
    (src/thor/score.py:76-88) and write them into the trajectory; eps equals the full output rows + window_scatter to the rounding of
    the compute type, for one trajectory and for co-sampled members, and bit for bit between two runs."""
    torch.manual_seed(3)
    dev = torch.device("cuda", 0)
    net = ScoreUNet(channels=6, spatial=2, activation=torch.nn.SiLU, embedding_dim=64, hidden_channels=[64, 128], hidden_blocks=[1, 1],
                    attention_levels=[1], kernel_size=3, padding_mode="zeros").to(dev).eval()
    net.precision = prec
    eng = net._get_engine()
    sf = BatchedScoreFunction(net, markov_order=1, batch_size=5, device=dev, noise_process=SDAPipeline())
    calls, real = [], ops.conv_center

    def spy(*a, **kw):
        calls.append(a[4])
        return real(*a, **kw)
    ops.conv_center = spy
    try:
        with torch.no_grad():
            for shape in ((14, 2, 32, 32), (3, 9, 2, 32, 32)):
                x = torch.randn(*shape, device=dev)
                eng.use_center_conv = True
                a = sf.score_fn(x, 0.4).clone()
                n = sum(calls)
                again = sf.score_fn(x, 0.4).clone()
                eng.use_center_conv = False
                b = sf.score_fn(x, 0.4).clone()
                torch.cuda.synchronize()
                assert n == (shape[-4] - 2) * (shape[0] if len(shape) == 5 else 1) and sum(calls) == 2 * n
                del calls[:]
                assert torch.equal(a, again)
                assert (a - b).abs().max().item() <= tol * b.abs().max().item()
    finally:
        ops.conv_center = real
        eng.use_center_conv = True


def test_independent_stream_is_on_another_hardware_queue():
    """streams.py: HIP hands hardware queues to new streams in rotation and two streams on one queue run in order -- after a few
    streams exist, a freshly created one may sit on the caller's queue and the two-stream backward would silently serialise.
    ``independent_stream`` returns one whose work, handed over through an event, runs next to what the current stream goes on with
    (round 6: the probe times that pattern; the one-launch probe of rounds 4-5 passed on pairs that serialise), wherever the rotation stands."""
    from climate2weather_amd.streams import independent_stream, overtakes
    dev = torch.device("cuda", 0)
    plain = [torch.cuda.Stream(device=dev) for _ in range(12)]
    verdicts = [overtakes(s) for s in plain]
    torch.cuda.synchronize()
    for shift in range(4):  # whatever slot of the rotation comes next
        keep = [torch.cuda.Stream(device=dev) for _ in range(shift)]
        s = independent_stream(dev)
        assert overtakes(s), (shift, verdicts)
        del keep
    torch.cuda.synchronize()
    if all(verdicts):
        pytest.skip("no plain stream shared the default stream's queue on this runtime: the premise could not be shown (the helper still works)")
    assert 1 <= verdicts.count(False) <= 8, verdicts  # some slots of the rotation, not a broken probe
    # the engine's side stream (the gradient stream of the two-stream mode) is such a stream
    torch.manual_seed(0)
    net = ScoreUNet(channels=6, spatial=2, activation=torch.nn.SiLU, embedding_dim=64, hidden_channels=[64, 128], hidden_blocks=[1, 1],
                    attention_levels=[1], kernel_size=3, padding_mode="zeros").to(dev)
    eng = net._get_engine()
    assert (eng.grad_stream() is None) == (os.environ.get("C2W_WGRAD_STREAM", "0") != "1")  # one stream by default (round 5)
    assert overtakes(eng.side_stream())
    torch.cuda.synchronize()


def test_window_batch_floor_gives_the_same_score_in_fewer_launch_sequences():
    """score_fn.py::window_batch_floor (the product default; tests/conftest.py sets 0 for the rest of the suite): ``batch_size`` windows
    per network call become AT LEAST the floor -- same eps to fp32 round-off (the launch size picks the kernels, hence the summation
    order), one network call instead of six."""
    torch.manual_seed(3)
    dev = torch.device("cuda", 0)
    net = ScoreUNet(channels=6, spatial=2, activation=torch.nn.SiLU, embedding_dim=64, hidden_channels=[64, 128], hidden_blocks=[1, 1],
                    attention_levels=[1], kernel_size=3, padding_mode="zeros").to(dev).eval()
    net.precision = "fp32"
    sf = BatchedScoreFunction(net, markov_order=1, batch_size=2, device=dev, noise_process=SDAPipeline())
    assert sf.window_batch_floor == 0 and type(sf).window_batch_floor == 256
    x = torch.randn(14, 2, 64, 64, device=dev)  # 12 windows
    t = torch.tensor(0.4)
    eng = net._get_engine()
    calls, real = [], eng.forward

    def spy(*a, **kw):
        calls.append(kw["shape"][0])
        return real(*a, **kw)
    eng.forward = spy
    try:
        with torch.no_grad():
            exact = sf.score_fn(x, t).clone()
            assert calls == [2] * 6
            del calls[:]
            sf.window_batch_floor = type(sf).window_batch_floor
            one = sf.score_fn(x, t).clone()
            assert calls == [12]  # 256 windows of 128x128 = 1024 windows of 64x64: the whole trajectory
    finally:
        del eng.forward
    torch.cuda.synchronize()
    assert (one - exact).abs().max().item() <= 2e-5 * exact.abs().max().item()


def test_ensemble_driver_on_device():
    """a14 on the GPU: members of one rank, conditioned with the experiment's operator, state resident in HBM, bf16 network."""
    from climate2weather_amd.sampling import run_ensemble
    torch.manual_seed(1)
    net = ScoreUNet(channels=6, spatial=2, activation=torch.nn.SiLU, embedding_dim=64, hidden_channels=[64, 128], hidden_blocks=[1, 1],
                    attention_levels=[1], kernel_size=3, padding_mode="zeros").cuda().eval()  # bf16 needs channel counts in 64s
    A = PoolStrideOperator(8, 2)
    truth = torch.rand(9, 2, 32, 32)
    y = A(truth)
    kw = dict(length=9, n_vars=2, height=32, width=32, markov_order=1, num_samples=4, steps=8, batch_size=4, seed=3, A=A, y=y,
              std=torch.tensor([0.1, 0.1]).view(1, 2, 1, 1), gamma=1e-2, device=torch.device("cuda", 0), precision="bf16")
    out = run_ensemble(net, world=2, rank=1, **kw)
    assert [i for i, _ in out] == [2, 3]
    for _, x in out:
        assert x.is_cuda and x.shape == (9, 2, 32, 32) and torch.isfinite(x).all()
    again = run_ensemble(net, world=2, rank=1, **kw)
    assert all(torch.equal(a[1], b[1]) for a, b in zip(out, again))
    # co-sampled members (device-resident predictor + fused guidance, windows of both members in shared batches): the same
    # members as one by one; fp32 so that a different kernel choice at another batch size cannot show (<= 1e-4), with a corrector
    one = run_ensemble(net, world=1, rank=0, **dict(kw, precision="fp32", corrections=0))
    co = run_ensemble(net, world=1, rank=0, members_per_batch=3, **dict(kw, precision="fp32", corrections=0, batch_size=5))
    assert [i for i, _ in co] == [0, 1, 2, 3]
    for (_, a_), (_, b_) in zip(one, co):
        assert (a_ - b_).abs().max().item() <= 1e-4 * a_.abs().max().item()
    # ... and in bf16, the DEFAULT grouping (members_per_batch=None co-samples up to the window floor): equal to rounding -- the kernel a
    # layer runs on depends on the number of windows in a batch, and 8 sampler steps carry the difference (observed 1-2e-2 of the scale)
    one16 = run_ensemble(net, world=1, rank=0, members_per_batch=1, **dict(kw, corrections=0))
    co16 = run_ensemble(net, world=1, rank=0, **dict(kw, corrections=0))
    for (i_, a_), (j_, b_) in zip(one16, co16):
        assert i_ == j_ and (a_ - b_).abs().max().item() <= 6e-2 * a_.abs().max().item()
    pipe = SDAPipeline()
    sf = BatchedScoreFunction(net, markov_order=1, batch_size=6, device=torch.device("cuda", 0), noise_process=pipe)
    sf.condition_on(A=A, y=y, std=kw["std"], gamma=1e-2, exact_grad=False)
    g = torch.Generator().manual_seed(5)
    noise = torch.randn(2, 9, 2, 32, 32, generator=g).cuda()
    zs = [torch.randn(2, 9, 2, 32, 32, generator=g).cuda() for _ in range(3)]
    net.precision = "fp32"
    both = pipe.sample(sf, noise, steps=3, corrections=1, tau=0.4, show_progressbar=False, z_draws=zs)
    for m in range(2):
        alone = pipe.sample(sf, noise[m], steps=3, corrections=1, tau=0.4, show_progressbar=False, z_draws=[z[m] for z in zs])
        assert (both[m] - alone).abs().max().item() <= 1e-4 * alone.abs().max().item()


@pytest.mark.parametrize("corrections,cond", [(0, False), (1, False), (0, True)])
def test_ensemble_members_are_the_reference_members_on_the_gpu(corrections, cond):
    """a14 parity on the HIP path: seed s, rank r, member i is the member of the reference's loop (exp/downscaling.py:96-103,248-265;
    oracle/host.py::ensemble_members over the CPU oracle network / score function / sampler).  2 ranks x 2 members, fp32, <= 3e-4."""
    from climate2weather_amd.sampling import run_ensemble
    net = _tiny().eval()
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    kw = dict(length=5, n_vars=2, height=16, width=16, markov_order=1, num_samples=4, steps=3, batch_size=2, seed=7, corrections=corrections,
              tau=0.5)
    cnd = {}
    if cond:
        A = PoolStrideOperator(8, 2)
        truth = torch.rand(5, 2, 16, 16, generator=torch.Generator().manual_seed(9))
        cnd = dict(A=A, y=oh.measure(truth, 8, 2), std=torch.tensor([0.5, 0.3]).view(1, 2, 1, 1), gamma=1e-2)
    fwd = lambda a, b: ou.score_unet_forward(sd, a, b, hidden_blocks=[1, 1], attention_levels=[1])
    score = od.GuidedScore(fwd, 1, A=(lambda z: oh.measure(z, 8, 2)) if cond else None, y=cnd.get("y"), std=cnd.get("std"), gamma=1e-2,
                           exact_grad=False, batch_size=2)
    for rank in (0, 1):
        mine = run_ensemble(net, world=2, rank=rank, device=torch.device("cuda", 0), precision="fp32", exact_grad=False, **kw, **cnd)
        ref = oh.ensemble_members(lambda noise, zs: od.sample(score, noise, steps=3, corrections=corrections, tau=0.5, z_draws=zs),
                                  seed=7, rank=rank, world=2, num_samples=4, shape=(5, 2, 16, 16), steps=3, corrections=corrections)
        assert [i for i, _ in mine] == [i for i, _ in ref] == [2 * rank, 2 * rank + 1]
        for (_, a), (_, b) in zip(mine, ref):
            assert a.is_cuda
            assert (a.cpu() - b).abs().max().item() <= 3e-4 * b.abs().max().item()


def test_reference_style_training_loop_on_the_module_api(golden_dir, tmp_path):
    """The reference's loop with only the class names changed: module -> .cuda() -> StandardEMA deep copies ->
    torch.optim.AdamW(net.parameters()) -> [pipeline.loss(net, x).mean().backward(); optimizer.step(); ema.update()] -> snapshot.
    Everything that captures parameters is created BEFORE the first forward, like training_loop.py:116-131.  fp32: first step
    against the golden gradients / the oracle's AdamW; then autocast steps keep the loss finite and move the EMA."""
    from climate2weather_amd.ema import StandardEMA
    from climate2weather_amd.snapshot import load_network_snapshot, save_network_snapshot
    g = _golden(golden_dir, "tiny_net.npz")
    torch.manual_seed(3)
    net = ScoreUNet(channels=6, spatial=2, activation=torch.nn.SiLU, **TINY).cuda()
    ema = StandardEMA(net, rates=[0.9])
    opt = torch.optim.AdamW(net.parameters(), lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-3)
    pipe = SDAPipeline()
    x, t, eps = (torch.from_numpy(g[k]).cuda() for k in ("x", "t", "eps"))
    opt.zero_grad(set_to_none=True)
    loss = od.loss(net, x, t, eps).mean()
    loss.backward()
    assert loss.item() == pytest.approx(float(g["loss"]), rel=1e-4)
    opt.step()
    ema.update()
    for n, p in net.named_parameters():
        p0, gr = torch.from_numpy(g["sd." + n]), torch.from_numpy(g["grad." + n])
        exp = oh.adamw_step(p0, gr, torch.zeros_like(p0), torch.zeros_like(p0), 1, 1e-3)[0]
        _assert_adam_close(p.detach().cpu(), exp, gr, 1e-3, n)
        e = dict(ema.emas[0].named_parameters())[n].detach().cpu()
        assert torch.allclose(e, 0.9 * p0 + 0.1 * p.detach().cpu(), atol=1e-6), n
    # the same loop under autocast (fabric precision "bf16-mixed") on a network whose channel counts suit the bf16 kernels
    torch.manual_seed(4)
    net = ScoreUNet(channels=6, spatial=2, activation=torch.nn.SiLU, embedding_dim=64, hidden_channels=[64, 128], hidden_blocks=[1, 1],
                    attention_levels=[1], kernel_size=3, padding_mode="zeros").cuda()
    ema = StandardEMA(net, rates=[0.9])
    opt = torch.optim.AdamW(net.parameters(), lr=1e-3, weight_decay=1e-3)
    w_before = net.unet.heads[0].weight.detach().clone()
    losses = []
    for _ in range(3):
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            l = pipe.loss(net, x).mean()
        l.backward()
        opt.step()
        ema.update()
        losses.append(l.item())
    assert all(np.isfinite(losses)) and not torch.equal(net.unet.heads[0].weight.detach(), w_before)
    assert not torch.equal(ema.emas[0].unet.heads[0].weight, w_before)
    # the 16-bit shadow, padded input-conv operand and input-gradient operands followed torch.optim's / the EMA's writes: the
    # stepped module computes exactly what a FRESH module with the same weights computes (forward, loss, every gradient)
    for mod in (net, ema.emas[0]):
        fresh = ScoreUNet(channels=6, spatial=2, activation=torch.nn.SiLU, embedding_dim=64, hidden_channels=[64, 128], hidden_blocks=[1, 1],
                          attention_levels=[1], kernel_size=3, padding_mode="zeros").cuda()
        fresh.load_state_dict({k: v.detach().clone() for k, v in mod.state_dict().items()})
        mod.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            la = od.loss(mod, x, t, eps).mean()
            lb = od.loss(fresh, x, t, eps).mean()
        la.backward()
        lb.backward()
        assert la.item() == lb.item()
        for (n, a_), (_, b_) in zip(mod.named_parameters(), fresh.named_parameters()):
            assert (a_.grad - b_.grad).abs().max().item() <= 1e-5 * b_.grad.abs().max().item() + 1e-9, n  # split-K summation order only
    p = save_network_snapshot(str(tmp_path / "network-snapshot-0000001-0.900000.pkl"), ema.emas[0], pipe, dict(train=dict(window=3)))
    snap = load_network_snapshot(p, device="cuda")
    with torch.no_grad():
        ya = ema.emas[0](x, t.reshape(-1))
        yb = snap.ema(x, t.reshape(-1))
    assert (ya - yb).abs().max().item() <= 2e-2 * ya.abs().max().item()  # the snapshot stores fp16 weights


# ------------------------------------------------------------------------------------------------ f2: snapshot / checkpoint formats
def test_reference_network_snapshot_runs_on_the_hip_path(golden_dir):
    """the reference-made `network-snapshot-*.pkl` fixture (pickled fp16 reference module, tests/golden/make_snapshot.py)
    -> load_network_snapshot -> HIP forward, against the imported reference's fp32 outputs of the same network (tiny_net.npz).
    Tolerance = the fp16 rounding of the stored weights (training_loop.py:259): 5e-3 of scale (the CPU-emulated test sees the same)."""
    from climate2weather_amd.snapshot import load_network_snapshot
    g = _golden(golden_dir, "tiny_net.npz")
    snap = load_network_snapshot(os.path.join(golden_dir, "ref_snapshot_tiny.pkl"), device="cuda", precision="fp32")
    assert snap.markov_order == 1 and next(snap.ema.parameters()).is_cuda
    for k, v in snap.ema.state_dict().items():
        assert torch.equal(v.cpu(), torch.from_numpy(g["sd." + k]).to(torch.float16).float()), k
    with torch.no_grad():
        y = snap.ema(torch.from_numpy(g["xt"]).cuda(), torch.from_numpy(g["t"]).cuda())
        y32 = snap.ema(torch.from_numpy(g["x32"]).cuda(), torch.tensor(0.3).cuda())
    for a, ref in ((y, g["y"]), (y32, g["y32"])):
        assert (a.cpu() - torch.from_numpy(ref)).abs().max().item() <= 5e-3 * float(np.abs(ref).max())
    # the same weights through the oracle: what is left is the kernels' own error, <= 1e-4
    sd = {k: v.cpu() for k, v in snap.ema.state_dict().items()}
    yo = ou.score_unet_forward(sd, torch.from_numpy(g["x32"]), torch.tensor(0.3), hidden_blocks=[1, 1], attention_levels=[1])
    assert (y32.cpu() - yo).abs().max().item() <= 1e-4 * yo.abs().max().item()
    # and the sampler runs on it, as exp/downscaling.py:110-126,208-214 does with the unpickled module
    sf = BatchedScoreFunction(snap.ema, markov_order=snap.markov_order, batch_size=4, device=torch.device("cuda", 0), noise_process=snap.pipeline)
    x = snap.pipeline.sample(sf, torch.randn(5, 2, 32, 32), steps=2, show_progressbar=False)
    assert x.shape == (5, 2, 32, 32) and torch.isfinite(x).all()


def test_resume_from_a_reference_made_training_state_checkpoint_on_the_gpu(golden_dir):
    from _ckpt_check import resume_from_reference_checkpoint
    resume_from_reference_checkpoint(golden_dir, torch.device("cuda", 0), lambda seed: _tiny(seed=seed))


# ------------------------------------------------------------------------------------------------ f3: on-device dataset feed
@pytest.mark.parametrize("start_idx", [0, 37])
def test_device_window_feed_yields_the_reference_batches(start_idx):
    """the array lives in HBM and a batch is an index gather on the device; the batches are the items
    dataset.py:114-126 builds at the indices dataset.py:23-40 walks (oracle/host.py::infinite_order / window_item), here for rank 1
    of 2, across an epoch boundary, and resumed from `start_idx = cur_ndata` (training_loop.py:164-171)."""
    from climate2weather_amd.data import COSMODataset, DeviceWindowFeed, SyntheticWindowDataset
    arr = np.random.RandomState(1).randn(30, 4, 16, 16).astype(np.float32)
    ds = COSMODataset(arr, num_features=4, spatial_res=16, window=13, flatten=True)
    feed = DeviceWindowFeed(ds, torch.device("cuda", 0), rank=1, num_replicas=2, seed=5, start_idx=start_idx)
    order = oh.infinite_order(len(ds), rank=1, num_replicas=2, seed=5, start_idx=start_idx, count=3 * 7)
    assert len(ds) == 18 and len(set(order)) > 1
    data = torch.from_numpy(arr)
    for b in range(3):
        batch = feed.next_batch(7)
        assert batch.is_cuda and batch.shape == (7, 52, 16, 16) and batch.dtype == torch.float32
        for j in range(7):
            assert torch.equal(batch[j].cpu(), oh.window_item(data, order[7 * b + j], 13)), (b, j)
    # the synthetic stand-in used by bench.py: same item contract
    sds = SyntheticWindowDataset(n_frames=20, n_vars=2, height=16, width=16, window=3, seed=0)
    sfeed = DeviceWindowFeed(sds, torch.device("cuda", 0), rank=0, num_replicas=1, seed=0)
    so = oh.infinite_order(len(sds), 0, 1, 0, 0, 4)
    sb = sfeed.next_batch(4)
    for j in range(4):
        assert torch.equal(sb[j].cpu(), oh.window_item(sds.data, so[j], 3))


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_lazy_window_batches_train_like_gathered_ones(prec):
    """DeviceWindowFeed.next_batch(lazy=True) hands the trainer the window indices only; the input conversion reads the windows in
    place (c2w_windows_to_nhwc_noise) and addresses the noise stream with the dense (B,C,H,W) index, so the step is the SAME step:
    identical loss and identical weights after two optimizer steps, and the lazy batch materialises to the gathered batch
    (dataset.py:114-126 via oracle/host.py::window_item)."""
    from climate2weather_amd.data import DeviceWindowFeed, SyntheticWindowDataset, WindowBatch
    dev = torch.device("cuda", 0)
    cfg = dict(embedding_dim=64, hidden_channels=[64, 128], hidden_blocks=[1, 1], attention_levels=[1], kernel_size=3, padding_mode="zeros")
    ds = SyntheticWindowDataset(n_frames=24, n_vars=2, height=32, width=32, window=3, seed=0)
    outs = []
    for lazy in (False, True):
        torch.manual_seed(5)
        net = ScoreUNet(channels=6, spatial=2, activation=torch.nn.SiLU, **cfg).to(dev)
        tr = Trainer(net, lr=1e-3, precision=prec, ema_rates=[0.999], seed=77)
        feed = DeviceWindowFeed(ds, dev, rank=0, num_replicas=1, seed=3)
        losses = []
        for _ in range(2):
            batch = feed.next_batch(5, lazy=lazy)
            if lazy:
                assert isinstance(batch, WindowBatch) and tuple(batch.shape) == (5, 6, 32, 32)
                order = batch.first.cpu().tolist()
                m = batch.materialize()
                for j, i in enumerate(order):
                    assert torch.equal(m[j].cpu(), oh.window_item(ds.data, i, 3))
            losses.append(float(tr.step(batch)))
        outs.append((losses, tr.eng.flat.clone(), tr.last_noise_seed))
    assert outs[0][2] == outs[1][2]  # same noise seeds drawn
    # the loss and the LayerNorm modulation gradients are summed with fp32 atomics: two runs of the SAME step agree to rounding, not bits
    assert outs[0][0] == pytest.approx(outs[1][0], rel=1e-4), (outs[0][0], outs[1][0])
    dw = (outs[0][1] - outs[1][1]).abs()  # Adam's g / (|g| + eps) is ill-conditioned where |g| ~ eps: bound by the two steps taken,
    assert dw.max().item() <= 2 * 1e-3 and dw.mean().item() <= 1e-5  # and require agreement on average far below one step (1e-3)
    # what the two paths hand the network is identical bit for bit: the in-place conversion vs the conversion of the gathered batch
    batch = DeviceWindowFeed(ds, dev, rank=0, num_replicas=1, seed=9).next_batch(5, lazy=True)
    dt = DTYPE_F32 if prec == "fp32" else DTYPE_BF16
    td = torch.float32 if prec == "fp32" else torch.bfloat16
    musig = torch.rand(5, 2, device=dev)
    ld = 64 if prec != "fp32" else 32
    a = torch.zeros(5 * 32 * 32, ld, dtype=td, device=dev)
    b = torch.zeros_like(a)
    assert ops.windows_to_nhwc_noise(batch.data, batch.offsets(), 123456789, musig, a, 5, 6, 32 * 32, ld, dt)
    assert ops.nchw_to_nhwc_noise(batch.materialize().contiguous(), 123456789, musig, b, 5, 6, 32 * 32, ld, dt)
    torch.cuda.synchronize()
    assert torch.equal(a, b) and float(a.float().abs().sum()) > 0


def test_the_default_network_on_four_concurrent_streams_is_bit_identical_to_one_stream():
    """Soak for races between workgroups that only show up beside other kernels (round 3: a weight fragment read left outstanding
    across the barrier its ring slot is refilled behind -- one forward in ~200 under four streams): the bench-size batch through the
    default network on four HIP streams at once, 25 rounds, every result compared bit for bit with the single-stream one."""
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    net = ScoreUNet(channels=52, spatial=2, activation=torch.nn.SiLU, embedding_dim=512, hidden_blocks=[3] * 5,
                    hidden_channels=[128, 128, 256, 384, 512], kernel_size=3, padding_mode="zeros", attention_levels=[4]).to(dev).eval()
    net.precision = "bf16"
    x = torch.randn(128, 52, 128, 128, device=dev)
    t = torch.tensor(0.7, device=dev)
    streams = [torch.cuda.Stream() for _ in range(4)]
    with torch.no_grad():
        net(x, t)
        ref = net(x, t).clone()
        torch.cuda.synchronize()
        assert ref.isfinite().all()
        for rnd in range(25):
            outs = []
            for st in streams:
                st.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(st):
                    outs.append(net(x, t))
            torch.cuda.synchronize()
            for si, y in enumerate(outs):
                assert torch.equal(y, ref), f"round {rnd}, stream {si}: {int((y != ref).sum())} values differ (Engine.debug_trace localises the layer)"


def test_conditioned_score_evaluation_at_the_shipped_full_length_folds_like_the_reference():
    """exp/configs/001_clim-downscaling/biased_climate_hadgem.yml: num_hours = 8737, batch_size = 128, conditioned (t_step 6, s_step 16,
    the shipped likelihood_std / gamma, exact_grad = False) on the default network (F = 4, k = 6 -> 52 channels), bf16.  One score
    evaluation = 8725 windows in 69 batches over a 2.3 GB trajectory.  Size-independent properties of src/thor/score.py:76-88,143-185:
      * fold: frame i of eps (k <= i < L - k) is the CENTRE frame of window i - k's network output, frames 0..k-1 are the leading frames
        of the first window, frames L-k..L-1 the trailing frames of the last -- checked against the module called directly on the first,
        the last and a few middle window batches (same 128-window batches -> the same launches -> bit-equal);
      * conditioning is frame-local: the conditioned evaluation == guidance applied to the unconditioned one (bit-equal), and frames
        that are not observed (i % 6 != 0) keep the unconditioned score."""
    L, Fv, k, H = 8737, 4, 6, 128
    w = 2 * k + 1
    nwin = L - w + 1
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    net = ScoreUNet(channels=Fv * w, spatial=2, activation=torch.nn.SiLU, **DEFAULT).to(dev).eval()
    net.precision = "bf16"
    pipe = SDAPipeline()
    g = torch.Generator(device=dev).manual_seed(8737)
    x = torch.randn((L, Fv, H, H), device=dev, generator=g)
    t = torch.tensor(0.7)
    sf = BatchedScoreFunction(net, markov_order=k, batch_size=128, device=dev, noise_process=pipe)
    eng = net._get_engine()
    with torch.no_grad():
        # fold against direct module calls on whole batches of the evaluation (first, two middle ones, the ragged last: 8725 = 68 * 128 + 21).
        # With the full output rows scattered (use_center_conv off) every kept frame is bit-equal to the module's; with the centre
        # frames out of c2w_conv_center (the default: only the kept rows are computed, engine.py::_fold_output) they are the same
        # convolution summed in another order: equal to one rounding step of bf16 (the bound below is 1.5 steps: one step is reached exactly, lab/probes/fold_margins.py), and
        # bit-equal for all but 0.015 % of the elements.
        for center in (False, True):
            eng.use_center_conv = center
            eps_u = sf(x, t).clone()
            assert eps_u.shape == x.shape and bool(torch.isfinite(eps_u).all())
            for b0 in (0, 128 * 17, 128 * 40, 128 * 68):
                nb = min(128, nwin - b0)
                wins = torch.stack([x[i:i + w].reshape(w * Fv, H, H) for i in range(b0, b0 + nb)])
                y = net(wins, t.to(dev)).view(nb, w, Fv, H, H)
                got, want = eps_u[b0 + k: b0 + k + nb], y[:, k]
                if not center:
                    assert torch.equal(got, want), f"centre frames of windows {b0}..{b0 + nb - 1}"
                else:
                    d = (got - want).abs()
                    _MARGINS["fold_steps"] = max(_MARGINS.get("fold_steps", 0.0), float((d / (2.0 ** -7 * want.abs().clamp_min(2.0 ** -10))).max()))
                    assert bool((d <= 1.5 * 2.0 ** -7 * want.abs().clamp_min(2.0 ** -10)).all()), f"centre frames of windows {b0}..: max |d| {d.max().item():.3e}"
                    assert float((d != 0).float().mean()) < 0.02
                # the first / last window's other frames: out of the full output rows -- of the whole batch (bit-equal to the module's),
                # or of that one window alone (another launch size, another kernel: equal to one rounding step)
                def same(a, b, what):
                    if not center:
                        assert torch.equal(a, b), what
                    else:
                        assert bool(((a - b).abs() <= 1.5 * 2.0 ** -7 * b.abs().clamp_min(2.0 ** -10)).all()), what
                if b0 == 0:
                    same(eps_u[:k], y[0, :k], "leading frames come from the first window")
                if b0 + nb == nwin:
                    same(eps_u[L - k:], y[-1, k + 1:], "trailing frames come from the last window")
        # conditioning
        A = PoolStrideOperator(16, 6)
        std = torch.tensor([0.1692666615037876, 0.0425178630338289, 0.3268027589410125, 0.3268027589410125]).view(1, Fv, 1, 1)
        gamma = 0.0007196856730011522
        yobs = A(torch.randn((L, Fv, H, H), device=dev, generator=g) * 0.5 + 0.5)
        assert yobs.shape == ((L + 5) // 6, Fv, 8, 8)
        sf.condition_on(A=A, y=yobs, std=std, gamma=gamma, exact_grad=False)
        assert sf._fused_guidance is not None
        eps_c = sf(x, t)
        ref = eps_u.clone()
        mu, sigma = pipe._mu_sigma_f(float(t))
        ops.guidance(x, ref, yobs, std.reshape(Fv).to(dev), yobs.shape[0], Fv, H, H, 16, 6, mu, sigma, gamma)
        ne = eps_c != ref
        assert not bool(ne.any()), (f"{int(ne.sum())} elements differ in frames {ne.flatten(1).any(1).nonzero().flatten().tolist()[:12]}; "
                                    f"max |d| {(eps_c - ref).abs().nan_to_num(1e30).max().item():.3e}; non-finite {int((~torch.isfinite(eps_c)).sum())}")
        observed = torch.zeros(L, dtype=torch.bool, device=dev)
        observed[::6] = True
        assert torch.equal(eps_c[~observed], eps_u[~observed])
        assert not torch.equal(eps_c[observed], eps_u[observed])
        # the guidance term itself against the reference's formula on a few observed frames (fp32 torch ops, src/thor/score.py:48-57)
        for i in (0, 6 * 700, 8736):
            x0 = (x[i] - sigma * eps_u[i]) / mu
            err = yobs[i // 6] - F.avg_pool2d(x0, 16)
            var = std.to(dev)[0] ** 2 + gamma * (sigma / mu) ** 2
            J = F.interpolate((err / var)[None], scale_factor=16, mode="nearest")[0] / 256.0 / mu
            want = eps_u[i] - sigma * J
            assert (eps_c[i] - want).abs().max().item() <= 1e-4 * want.abs().max().item()



@pytest.mark.xfail(strict=False, reason="margin watch, not a gate: the centre conv sums in another order than the full output conv; one bf16 rounding "
                                        "step is reached exactly (lab/probes/fold_margins.py); the gate is 1.5 steps")
def test_margin_fold_within_one_rounding_step():
    """The bound the fold test had before round 4 widened it (1 -> 1.5 bf16 rounding steps), kept as a watched margin."""
    if "fold_steps" not in _MARGINS:
        pytest.skip("the full-length fold test did not run")
    assert _MARGINS["fold_steps"] < 1.0, _MARGINS["fold_steps"]
