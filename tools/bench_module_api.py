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
p.add_argument("--ddp", action="store_true", help="wrap the module in torch's DistributedDataParallel over a one-rank RCCL group (what fabric.setup_module does)")
p.add_argument("--bucket-view", action="store_true", help="with --ddp: gradient_as_bucket_view=True (p.grad become views of the reducer's buckets: no copy-out; "
                                                        "the drop-in AdamW gathers them for its fused step)")
p.add_argument("--segments", type=int, default=None, help="ScoreUNet.grad_segments (default: 8 under more than one rank, else 1)")
a = p.parse_args()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
wrap = None
if a.ddp or a.segments is not None:
    import torch.distributed as dist
    if a.ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)

    def wrap(net):
        if a.segments is not None:
            net.grad_segments = a.segments
        return torch.nn.parallel.DistributedDataParallel(net, device_ids=[0], gradient_as_bucket_view=a.bucket_view) if a.ddp else net
res = bench.module_api(dev, a, 1.0, legs=tuple(a.legs.split(",")), item=not a.no_item, lazy=a.lazy, wrap=wrap)
print(json.dumps(res, indent=1))
