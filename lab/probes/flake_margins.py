"""Observed values behind the tolerance-based assertions of the round's new GPU tests, several runs in one process."""
import os, sys, tempfile
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
import torch.multiprocessing as mp
from _ddp_worker import run_module_ddp_gpu
import socket


def port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


if __name__ == "__main__":
    for i in range(int(os.environ.get("N", "5"))):
        with tempfile.TemporaryDirectory() as d:
            mp.spawn(run_module_ddp_gpu, args=(1, port(), d), nprocs=1, join=True)
            o = torch.load(os.path.join(d, "modgpu0.pt"), weights_only=False)
        a, b = o["ddp"], o["plain"]
        dl = max(abs(x - y) / abs(y) for x, y in zip(a["losses"], b["losses"]))
        df = (a["flat"] - b["flat"]).abs()
        print(f"run {i}: loss rel diff {dl:.2e} (bound 2e-3)  weights max {df.max().item():.2e} (6.5e-3) mean {df.mean().item():.2e} (2e-4)  "
              f"ema max {(a['ema'] - b['ema']).abs().max().item():.2e} (1e-5)", flush=True)
