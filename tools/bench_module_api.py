"""The reference-shaped loop (bench.py::module_api) on its own: `python tools/bench_module_api.py [--legs bf16_autocast,...] [--steps N]`.
Used under rocprofv3 (kernel-trace stats of the drop-in path) and for A/B runs; bench.py reports the same legs in `module_api`."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--steps", type=int, default=10)
p.add_argument("--warmup", type=int, default=3)
p.add_argument("--batch", type=int, default=128)
p.add_argument("--vars", type=int, default=5)
p.add_argument("--markov-order", type=int, default=6)
p.add_argument("--size", type=int, default=128)
p.add_argument("--legs", default="bf16_autocast,fp16_autocast_gradscaler,trainer_fp16,trainer_bf16")
p.add_argument("--no-item", action="store_true", help="drop the per-step loss.item() of training_loop.py:385 (diagnostic)")
p.add_argument("--lazy", action="store_true", help="hand pipeline.loss the un-gathered WindowBatch (diagnostic)")
a = p.parse_args()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
res = bench.module_api(dev, a, 1.0, legs=tuple(a.legs.split(",")), item=not a.no_item, lazy=a.lazy)
print(json.dumps(res, indent=1))
