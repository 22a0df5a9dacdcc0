"""bench.py's own rank launcher (`python bench.py --gpus N` without torch.distributed.run: what fabric.launch() does for the
reference, train.py:93-100) on the CPU, with subprocess.Popen replaced by a recorder: per-child RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* wiring, rank 0's stdout relayed, a dead rank turns into a non-zero exit status, more ranks than visible GPUs are refused,
and the parent never asks HIP for the device count."""
import argparse
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402


class FakeProc:
    """What launch() uses of subprocess.Popen: poll / communicate / kill / returncode.  `life`: polls until it exits by itself
    (None: never -- a rank stuck in a collective)."""

    def __init__(self, argv, env, stdout, rc=0, out=b"", life=1):
        self.argv, self.env, self.stdout_arg, self.rc_final, self.out, self.life = argv, env, stdout, rc, out, life
        self.returncode = None
        self.killed = False
        self.polls = 0

    def poll(self):
        if self.returncode is None:
            self.polls += 1
            if self.killed:
                self.returncode = -9
            elif self.life is not None and self.polls > self.life:
                self.returncode = self.rc_final
        return self.returncode

    def communicate(self):
        import time as _t
        while self.returncode is None:
            _t.sleep(0.01)
        return self.out, None

    def kill(self):
        self.killed = True


def _launch(n, rcs=None, hang=None, count=lambda: 8, grace=0.2, straggler_grace=0.3):
    procs = []

    def popen(argv, env=None, stdout=None):
        r = len(procs)
        p = FakeProc(argv, env, stdout, rc=(rcs or {}).get(r, 0), out=b'{"metric": "m", "n_gpus": %d}\n' % n if r == 0 else b"",
                     life=None if r == hang else 2 + r)
        procs.append(p)
        return p
    a = argparse.Namespace(gpus=n)
    return bench.launch(a, popen=popen, count=count, grace=grace, straggler_grace=straggler_grace), procs


def test_launcher_wires_one_rank_per_gpu(capsys, monkeypatch):
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    monkeypatch.setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rc, procs = _launch(4)
    assert rc == 0 and len(procs) == 4
    ports = {p.env["MASTER_PORT"] for p in procs}
    assert len(ports) == 1 and 1024 < int(next(iter(ports))) < 65536
    for r, p in enumerate(procs):
        assert p.argv[0] == sys.executable and p.argv[1] == os.path.abspath(bench.__file__) and p.argv[2:] == ["--gpus", "4", "--steps", "3"]
        assert (p.env["RANK"], p.env["LOCAL_RANK"], p.env["WORLD_SIZE"], p.env["LOCAL_WORLD_SIZE"]) == (str(r), str(r), "4", "4")
        assert p.env["MASTER_ADDR"] == "127.0.0.1" and p.env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
        assert (p.stdout_arg == subprocess.PIPE) == (r == 0)  # only rank 0's stdout is captured: it carries the one JSON line
    assert capsys.readouterr().out == '{"metric": "m", "n_gpus": 4}\n'
    # a child under this environment goes straight to run_rank(): main() only launches when WORLD_SIZE is absent
    assert "WORLD_SIZE" in procs[0].env


def test_launcher_reports_a_dead_rank_and_kills_the_ranks_it_leaves_stuck(monkeypatch):
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    rc, _ = _launch(2, rcs={1: 1})
    assert rc == 1
    rc, _ = _launch(2, rcs={0: -11})  # rank 0 killed by a signal
    assert rc == 11
    # rank 1 dies at once (no device), rank 0 and rank 2 wait for it in the rendezvous forever: killed after the grace period
    rc, procs = _launch(3, rcs={1: 3}, hang=0, grace=0.2)
    assert procs[0].killed and not procs[1].killed and rc == 9
    rc, procs = _launch(2, hang=1, rcs={0: 5}, grace=0.2)
    assert procs[1].killed and rc == 9


def test_launcher_kills_a_rank_that_outlives_a_clean_exit(monkeypatch, capsys):
    """Every rank exits 0 except one that hangs (destroy_process_group / a collective nobody joins any more): no rank FAILED, so the
    failure deadline never arms -- the straggler deadline, armed by the first clean exit, ends the job and reports a failure, and
    rank 0's line is still relayed (round-3 advice: this case used to spin forever)."""
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    rc, procs = _launch(2, rcs={0: 0}, hang=1, straggler_grace=0.3)
    assert procs[1].killed and not procs[0].killed and rc == 9
    assert capsys.readouterr().out == '{"metric": "m", "n_gpus": 2}\n'
    rc, procs = _launch(3, hang=0, straggler_grace=0.3)  # rank 0 itself is the straggler
    assert procs[0].killed and rc == 9


def test_launcher_refuses_more_ranks_than_gpus_and_never_calls_hip(monkeypatch, capsys):
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4"])
    import torch
    monkeypatch.setattr(torch.cuda, "device_count", lambda: pytest.fail("the launcher parent must not ask HIP for the device count"))
    monkeypatch.setattr(torch.cuda, "is_available", lambda: pytest.fail("the launcher parent must not initialise the GPU"))
    rc, procs = _launch(4, count=lambda: 2)
    assert rc == 2 and procs == [] and "only 2 GPU(s) are visible" in capsys.readouterr().err
    rc, procs = _launch(4, count=lambda: None)  # topology unreadable: --gpus is trusted, a rank without a device fails by itself
    assert rc == 0 and len(procs) == 4


def test_visible_gpus_reads_the_kfd_topology(tmp_path, monkeypatch):
    nodes = tmp_path / "nodes"
    for i, simd in enumerate([0, 0, 1024, 1024, 1024]):  # two CPU nodes, three GPUs
        d = nodes / str(i)
        d.mkdir(parents=True)
        (d / "properties").write_text(f"cpu_cores_count {0 if simd else 64}\nsimd_count {simd}\nmem_banks_count 1\n")
    real_listdir, real_open = os.listdir, open
    root = "/sys/class/kfd/kfd/topology/nodes"
    monkeypatch.setattr(bench.os, "listdir", lambda p: real_listdir(str(nodes)) if p == root else real_listdir(p))
    import builtins
    monkeypatch.setattr(builtins, "open", lambda p, *a, **k: real_open(str(p).replace(root, str(nodes)), *a, **k))
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    assert bench.visible_gpus() == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2")
    assert bench.visible_gpus() == 2


def test_visible_gpus_asks_a_child_process_where_sysfs_is_not_readable(monkeypatch):
    """Containers without /sys/class/kfd: a throw-away child counts the devices (it may initialise the runtime; the parent does not)."""
    real_listdir = os.listdir
    root = "/sys/class/kfd/kfd/topology/nodes"

    def listdir(p):
        if p == root:
            raise FileNotFoundError(p)
        return real_listdir(p)
    monkeypatch.setattr(bench.os, "listdir", listdir)
    calls = []

    def run(argv, **kw):
        calls.append(argv)
        return subprocess.CompletedProcess(argv, 0, stdout="some banner\n4\n", stderr="")
    assert bench.visible_gpus(run=run) == 4
    assert calls and calls[0][0] == sys.executable and "device_count" in calls[0][2]
    assert bench.visible_gpus(run=lambda argv, **kw: subprocess.CompletedProcess(argv, 1, stdout="", stderr="boom")) is None


# ---------------------------------------------------------------------------------------------------- the line the driver parses
def _fake_full_result():
    """The full result of a real run (round 4's 24 KB line, profiles/r04_final_bench.json) -- the input that broke the record -- with the
    legs added since; falls back to a synthetic dict of the same shape where that profile is not in the tree."""
    import json
    path = os.path.join(REPO, "profiles", "r04_final_bench.json")
    if os.path.exists(path):
        full = json.load(open(path))
    else:
        full = dict(metric="m", value=2700.0, unit="windows/s", n_gpus=1, steps=20, warmup=5, ms_per_step=47.0, higher_is_better=True, scaling="weak",
                    vs_baseline=None, dtype="bf16", data="synthetic", config=dict(workload="w" * 150),
                    roofline=dict(bound="mfma", kernel="k" * 400, achieved=1050.0, peak=2500.0, unit="TFLOP/s", frac=0.42, traffic=1.7e9,
                                  avg_launch_ms=0.59, flops_per_launch=6.18e11, launches_timed=120, from_profile=dict(note="n" * 2000)),
                    cpu_baseline=dict(value=4.2, unit="windows/s", cores=16, kind="port", sample="s" * 500, cpu_model="cpu", legs={}),
                    by_kernel=dict(kernels=[dict(kernel="k" * 60, ms=1.0)] * 80))
    full.setdefault("module_api", {})["trainer_bf16_c52"] = dict(windows_per_s=2750.0)
    return full


def test_compact_line_is_bounded_and_parses_with_roofline_and_cpu_baseline():
    import json
    full = _fake_full_result()
    assert len(json.dumps(full)) > 8192  # the input really is the kind that overflowed the driver's 8 KB tail
    line = bench.compact_line(full, "gpurun_out/bench_extras.json")
    assert "\n" not in line and len(line) < 6144 == bench.COMPACT_LIMIT
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["value"] == full["value"] and d["ms_per_step"] == full["ms_per_step"] and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and "traffic" in r
    assert r["avg_launch_ms"] > 0 and r["flops_per_launch"] > 0
    c = d["cpu_baseline"]
    assert c["value"] > 0 and c["cores"] >= 1 and c["kind"] == "port" and c["sample"]
    assert d["trainer_bf16_c52_windows_per_s"] == 2750.0 and d["extras_file"] == "gpurun_out/bench_extras.json"
    for k in ("by_kernel", "deep_variant", "module_api", "sampler_configs3"):
        assert k not in d


def test_compact_line_stays_bounded_when_strings_grow():
    import json
    full = _fake_full_result()
    full["roofline"]["kernel"] = "x" * 5000
    full["cpu_baseline"]["sample"] = "y" * 5000
    full["config"]["workload"] = "z" * 300
    line = bench.compact_line(full, None)
    assert len(line) < bench.COMPACT_LIMIT
    d = json.loads(line)
    assert d["roofline"]["frac"] and d["cpu_baseline"]["value"]


def test_emit_writes_extras_to_a_file_and_one_line_to_the_descriptor(tmp_path, monkeypatch):
    import json
    full = _fake_full_result()
    monkeypatch.setenv("C2W_BENCH_EXTRAS", str(tmp_path / "x" / "extras.json"))
    rd, wr = os.pipe()
    bench.emit(wr, full)
    os.close(wr)
    data = b""
    while True:
        chunk = os.read(rd, 65536)
        if not chunk:
            break
        data += chunk
    os.close(rd)
    assert data.count(b"\n") == 1 and len(data) < 6144
    assert json.loads(data)["roofline"]["frac"] == full["roofline"]["frac"]
    assert json.load(open(tmp_path / "x" / "extras.json"))["by_kernel"] == full["by_kernel"]


def test_a_failing_extra_leg_is_reported_and_the_record_survives(capsys):
    import json
    out = dict(value=1.0)

    def boom():
        raise RuntimeError("out of memory in an extra leg")
    assert bench._leg(out, "deep_variant", boom) is None
    assert bench._leg(out, "fine", lambda: 7) == 7
    assert out["errors"] == {"deep_variant": "RuntimeError: out of memory in an extra leg"}
    assert "extra leg 'deep_variant' failed" in capsys.readouterr().err
    full = _fake_full_result()
    full["errors"] = out["errors"]
    full["deep_variant"] = None
    d = json.loads(bench.compact_line(full, "x.json"))
    assert d["extra_legs_failed"] == ["deep_variant"] and d["value"] == full["value"] and d["roofline"]["frac"]
