"""RCCL ("nccl" backend) runs of the two paths that communicate: the training step's bucketed gradient
all-reduce issued from the gradient stream (training_loop.py:116,373-378) and the time-sharded sampler's halo exchange.

* world_size 1: runs on every GPU box -- a real RCCL communicator, the initial broadcast, every bucket all-reduced from the
  side stream while backward is still running, the optimizer ordered behind the collectives.
* world_size 2: the gloo tests' assertions (tests/test_host_emulated.py) on two GPUs; skipped where fewer than two are visible.
Workers are fresh processes (tests/_ddp_worker.py, tests/_shard_worker.py): one process per GPU, like the product.
"""
import os
import socket

import numpy as np
import pytest
import torch

from oracle import host as oh

pytestmark = pytest.mark.gpu

NGPU = torch.cuda.device_count()
two_gpus = pytest.mark.skipif(NGPU < 2, reason="needs two GPUs (RCCL refuses two ranks on one device)")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _golden(golden_dir, name):
    return {k: v for k, v in np.load(os.path.join(golden_dir, name), allow_pickle=False).items()}


def _check_step(golden_dir, tmp_path, world, bucket_mb):
    import torch.multiprocessing as mp
    from _ddp_worker import run
    mp.spawn(run, args=(world, _free_port(), golden_dir, str(tmp_path), bucket_mb, "nccl"), nprocs=world, join=True)
    g = _golden(golden_dir, "tiny_net.npz")
    outs = [torch.load(tmp_path / f"rank{r}.pt", weights_only=False) for r in range(world)]
    assert all(o["world"] == world and o["backend"] == "nccl" for o in outs)
    if bucket_mb < 1:
        assert outs[0]["nb"] > 4  # several buckets chased the backward
    for k, v in outs[0]["sd"].items():
        p0, gr = torch.from_numpy(g["sd." + k]), torch.from_numpy(g["grad." + k])
        exp = oh.adamw_step(p0, gr, torch.zeros_like(p0), torch.zeros_like(p0), 1, 1e-3)[0]
        big = gr.abs() > 1e-5  # g/(|g|+eps) is ill-conditioned where |g| ~ eps (see tests/test_host_emulated.py)
        assert torch.allclose(v[big], exp[big], atol=3e-6), k
        assert (v - exp).abs().max().item() <= 2e-3, k
        for o in outs[1:]:
            assert torch.equal(v, o["sd"][k]), k  # ranks stay in lock step
    assert sum(o["loss"] for o in outs) / world == pytest.approx(float(g["loss"]), rel=1e-4)
    if world > 1:
        assert outs[0]["draws"]["seed"] != outs[1]["draws"]["seed"]


@pytest.mark.parametrize("bucket_mb", [48.0, 0.05])
def test_training_step_over_rccl_single_rank(golden_dir, tmp_path, bucket_mb):
    _check_step(golden_dir, tmp_path, 1, bucket_mb)


@two_gpus
@pytest.mark.parametrize("bucket_mb", [48.0, 0.05])
def test_training_step_over_rccl_two_ranks_equals_single_process(golden_dir, tmp_path, bucket_mb):
    _check_step(golden_dir, tmp_path, 2, bucket_mb)


@pytest.mark.parametrize("world", [1, pytest.param(2, marks=two_gpus)])
def test_optimizer_chasing_the_backward_over_rccl(golden_dir, tmp_path, world):
    """Trainer.chase_optimizer (C2W_CHASE_OPT=1) through RCCL: all-reduce + fused AdamW + EMA per finished bucket on the gradient
    stream while the backward runs, against the update behind the backward: same weights after two steps."""
    import torch.multiprocessing as mp
    from _ddp_worker import run_chase
    mp.spawn(run_chase, args=(world, _free_port(), golden_dir, str(tmp_path), "nccl"), nprocs=world, join=True)
    outs = [torch.load(tmp_path / f"chase{r}.pt", weights_only=False) for r in range(world)]
    for r in outs:
        assert r[True]["update_calls"] > 2 * 4 and r[False]["update_calls"] == 2
        for k, v in r[False]["sd"].items():
            # same arithmetic per element; the gradients themselves carry fp32-atomics order noise (LayerNorm dm, loss sums), which
            # Adam's g / (|g| + eps) amplifies where |g| ~ eps: at most the two steps' full stride there, nothing on average
            d = (v - r[True]["sd"][k]).abs()
            assert d.max().item() <= 2 * 2e-3 and d.mean().item() <= 2e-6, (k, d.max().item(), d.mean().item())
    for r in outs[1:]:
        for k, v in outs[0][True]["sd"].items():
            assert torch.equal(v, r[True]["sd"][k]), k


@pytest.mark.parametrize("world", [1, pytest.param(2, marks=two_gpus)])
def test_bf16_wire_format_of_the_gradient_all_reduce_over_rccl(golden_dir, tmp_path, world):
    """Trainer(allreduce_dtype="bf16") through RCCL: every bucket cast, all-reduced as bfloat16 from the gradient stream, cast back."""
    import torch.multiprocessing as mp
    from _ddp_worker import run_wire
    from test_host_emulated import _check_wire
    mp.spawn(run_wire, args=(world, _free_port(), golden_dir, str(tmp_path), "nccl"), nprocs=world, join=True)
    _check_wire([torch.load(tmp_path / f"wire{r}.pt", weights_only=False) for r in range(world)])


@pytest.mark.parametrize("world", [1, pytest.param(2, marks=two_gpus)])
def test_full_size_bf16_steps_through_an_rccl_communicator(tmp_path, world):
    """The bench's network and precision through RCCL (one rank on every box, two where two GPUs are visible): 12 buckets of 25 MB
    all-reduced from inside the backward per step, next to the real wgrad_patch / conv_patch launches; the weights follow the same
    steps taken without a process group."""
    import torch.multiprocessing as mp
    from _ddp_worker import run_full_size
    mp.spawn(run_full_size, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    outs = [torch.load(tmp_path / f"full{r}.pt", weights_only=False) for r in range(world)]
    for o in outs:
        d, p = o["dist"], o["plain"]
        # the backward runs on one stream by default (round 5); with the fp32 wire the (asynchronous) collectives are issued from it and
        # waited for behind the backward; the communication stream carries the sequences that wait inside the backward (bf16 wire,
        # chased update: Trainer._comm_stream, round 6); with C2W_WGRAD_STREAM=1 the gradient stream carries everything
        two = os.environ.get("C2W_WGRAD_STREAM") == "1"
        assert d["nb"] == 12 and d["all_reduces"] == 3 * 12 and d["on_side_stream"] == two and p["on_side_stream"] == two
        assert all(np.isfinite(d["losses"])) and all(np.isfinite(p["losses"]))
    if world == 1:  # a one-rank sum is the identity: the communicator must not change the step
        d, p = outs[0]["dist"], outs[0]["plain"]
        assert p["all_reduces"] == 0
        assert d["losses"] == pytest.approx(p["losses"], rel=2e-3)
        # three AdamW steps of lr 1e-4 move a weight by <= 3e-4; bf16 gradients + atomics-order noise may flip g/(|g| + eps) where |g| ~ eps
        assert (d["flat"] - p["flat"]).abs().max().item() <= 6.5e-4
        assert (d["flat"] - p["flat"]).abs().mean().item() <= 2e-5
    else:
        assert torch.equal(outs[0]["dist"]["flat"], outs[1]["dist"]["flat"])  # ranks in lock step


@pytest.mark.parametrize("world", [1, pytest.param(2, marks=two_gpus)])
def test_drop_in_module_under_torch_ddp_over_rccl_with_segmented_gradient_delivery(tmp_path, world):
    """training_loop.py:116,369-391 on the HIP engine: the module wrapped in torch's DistributedDataParallel over RCCL, five class-name
    seams, bf16 autocast, the backward pass as the segmented chain (score.py::_GradSegment).  DDP's reducer copies gradients into its
    buckets on the caller's stream and all-reduces them on its own while the engine's gradient stream is still writing later ranges:
    the chain's events order the two.  One rank: the same steps as the loop without a process group (a one-rank mean is the
    identity); two ranks: lock step."""
    import torch.multiprocessing as mp
    from _ddp_worker import run_module_ddp_gpu
    mp.spawn(run_module_ddp_gpu, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    outs = [torch.load(tmp_path / f"modgpu{r}.pt", weights_only=False) for r in range(world)]
    for o in outs:
        d = o["ddp"]
        assert d["flat_path"] and all(np.isfinite(d["losses"]))
        assert len(set(d["arrivals"])) >= 3 and d["arrivals"][0] < d["launches"], (sorted(set(d["arrivals"])), d["launches"])
    if world == 1:
        d, p = outs[0]["ddp"], outs[0]["plain"]
        assert len(set(p["arrivals"])) == 1  # the single node delivers everything at the end
        assert d["losses"] == pytest.approx(p["losses"], rel=2e-3)
        # three AdamW steps of lr 1e-3: bf16 gradients + atomics-order noise may flip g / (|g| + eps) where |g| ~ eps
        assert (d["flat"] - p["flat"]).abs().max().item() <= 6.5e-3 and (d["flat"] - p["flat"]).abs().mean().item() <= 2e-4
        assert (d["ema"] - p["ema"]).abs().max().item() <= 1e-5
    else:
        assert torch.equal(outs[0]["ddp"]["flat"], outs[1]["ddp"]["flat"])


@two_gpus
def test_time_sharded_sampler_over_rccl_two_ranks(golden_dir, tmp_path):
    import torch.multiprocessing as mp
    from _shard_worker import run
    mp.spawn(run, args=(2, _free_port(), golden_dir, str(tmp_path), "nccl"), nprocs=2, join=True)
    s = _golden(golden_dir, "sampler.npz")
    r0, r1 = (torch.load(tmp_path / f"shard{r}.pt", weights_only=False) for r in (0, 1))
    assert r0["uncond_c0.bounds"] == [(0, 5), (5, 9)]
    sg = _golden(golden_dir, "sampler_gamma.npz")  # per-variable gamma as exp/downscaling.py:228-233 builds it
    for name in ("uncond_c0", "uncond_c1", "cond_c0", "cond_c0_gvec", "cond_c1_exact"):
        ref = torch.from_numpy(sg[name + ".x"] if name.endswith("_gvec") else s[name + ".x"])
        assert torch.equal(r0[name], r1[name]), name
        if name + ".no_overlap" in r0:  # interior windows evaluated while the halos were in flight == halos first
            assert (r0[name] - r0[name + ".no_overlap"]).abs().max().item() <= 1e-5 * ref.abs().max().item(), name
        assert (r0[name] - ref).abs().max().item() <= 3e-4 * ref.abs().max().item(), name


def test_bench_launcher_spawns_one_rank_per_gpu(tmp_path):
    """`python bench.py --gpus N` without torchrun starts N rank processes itself (train.py:93-100: fabric.launch()) and relays
    rank 0's single JSON line; n_gpus is the world size RCCL saw."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n = 2 if NGPU >= 2 else 1
    env = dict(os.environ, C2W_FORCE_DIST="1")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1", "--batch", "8",
                          "--no-cpu-baseline", "--sample-steps", "0", "--no-extras"], capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == n and rec["config"]["parallelism"] == f"dp{n}" and rec["config"]["global_batch"] == 8 * n
    assert rec["world_size_rccl"] == n and rec["value"] > 0


def test_bench_multi_rank_extras_run_through_the_communicator(tmp_path):
    """The extras every rank of `bench.py --gpus N` runs -- by_kernel steps (collectives inside), the sampler legs and the member-sharded
    sampler record with its barrier + MAX all-reduce -- through a real RCCL communicator (one rank per visible GPU, at least one)."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n = 2 if NGPU >= 2 else 1
    extras = tmp_path / "bench_extras.json"
    env = dict(os.environ, C2W_FORCE_DIST="1", C2W_BENCH_EXTRAS=str(extras))
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1", "--batch", "8",
                          "--no-cpu-baseline", "--sample-steps", "1", "--kernel-steps", "1", "--light-extras"],
                         capture_output=True, text=True, env=env, timeout=1200)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and len(lines[0]) < 6144, (len(lines), [len(l) for l in lines])  # ONE compact line: what the driver parses
    rec = json.loads(lines[0])
    # the driver's contract for the one JSON line (metric / value / unit / ... / roofline / cpu_baseline keys present, value = whole job)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline", "cpu_baseline", "step_ms"):
        assert key in rec, key
    assert rec["unit"] == "windows/s" and rec["higher_is_better"] is True and rec["scaling"] == "weak" and rec["vs_baseline"] is None
    assert rec["dtype"] == "bf16" and rec["data"] == "synthetic" and "workload" in rec["config"] and "model" not in rec["config"]
    assert abs(rec["value"] - 8 * n * rec["steps"] / (rec["ms_per_step"] * rec["steps"] / 1e3)) <= 0.01 * rec["value"]
    rf = rec["roofline"]
    assert rf is None or (rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and "traffic" in rf)
    # the legs live in the extras file the line names; the line carries their scalars
    full = json.load(open(extras))
    sm = full["sampler_member_sharded"]
    assert sm["n_gpus"] == n and sm["members_total"] == 8 * n and sm["window_forwards_per_s"] > 0
    assert rec["sampler_member_sharded_window_forwards_per_s"] == sm["window_forwards_per_s"]
    assert full["by_kernel"]["serialised_step_ms"] == rec["serialised_step_ms"] > 0 and "module_api" not in full and "deep_variant" not in full
    assert "by_kernel" not in rec and full["value"] == rec["value"]


def test_bench_launcher_counts_gpus_without_hip_and_refuses_more_ranks_than_gpus():
    """bench.visible_gpus() reads the KFD topology of THIS box (no HIP call); asking for one rank more than there are GPUs is
    refused (rc 2) before any child starts -- or, where the topology is unreadable, fails in the rank that finds no device."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, repo)
    import bench
    seen = bench.visible_gpus()
    assert seen is None or seen == NGPU, (seen, NGPU)
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", str(NGPU + 1), "--steps", "1", "--warmup", "0", "--batch", "4",
                          "--no-cpu-baseline", "--no-extras"], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode != 0 and out.stdout.strip() == ""
    if seen is not None:
        assert out.returncode == 2 and f"only {NGPU} GPU(s) are visible" in out.stderr

