"""Which use of a side stream makes it stop overlapping with the compute stream (tools/probe_stream_overlap_dist.py, stage 8)?"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch, torch.distributed as dist
from climate2weather_amd import streams
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
cur = torch.cuda.current_stream()
US = 2350


def pattern(side, tag, wait=True):
    def body():
        for _ in range(12):
            torch.cuda._sleep(500 * US)
            if wait:
                side.wait_stream(cur)
            with torch.cuda.stream(side):
                torch.cuda._sleep(200 * US)
    body(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); body(); cur.wait_stream(side); e1.record(); torch.cuda.synchronize()
    print(f"{tag}: {e0.elapsed_time(e1):.2f} ms", flush=True)


os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29545")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
g32 = torch.randn(1 << 22, device=dev)
g16 = torch.empty(1 << 22, dtype=torch.bfloat16, device=dev)

S = streams.independent_stream(dev)
pattern(S, "a0 fresh stream")
with torch.cuda.stream(S):
    for _ in range(36):
        g16.copy_(g32); g32.copy_(g16)
torch.cuda.synchronize()
pattern(S, "a1 after 72 cast kernels on it")
for _ in range(36):
    S.wait_stream(cur); torch.cuda._sleep(10 * US); cur.wait_stream(S)
torch.cuda.synchronize()
pattern(S, "a2 after 36 wait_stream round trips")
with torch.cuda.stream(S):
    for _ in range(36):
        w = dist.all_reduce(g32, async_op=True); w.wait()
torch.cuda.synchronize()
pattern(S, "a3 after 36 async all_reduce(fp32) + work.wait() issued from it")
with torch.cuda.stream(S):
    for _ in range(36):
        g16.copy_(g32); w = dist.all_reduce(g16, async_op=True); w.wait(); g32.copy_(g16)
torch.cuda.synchronize()
pattern(S, "a4 after 36 x (cast, all_reduce(bf16), wait, cast back) issued from it")
with torch.cuda.stream(S):
    for _ in range(36):
        v = g16[1000:5000]; v.copy_(g32[1000:5000]); w = dist.all_reduce(v, async_op=True); w.wait(); g32[1000:5000].copy_(v)
torch.cuda.synchronize()
pattern(S, "a5 the same on slices")

from climate2weather_amd.score import ScoreUNet
from climate2weather_amd.training import Trainer
os.environ["C2W_FORCE_DIST"] = "1"
cfg = dict(embedding_dim=64, hidden_channels=[64, 64], hidden_blocks=[1, 1], attention_levels=[1], kernel_size=3, padding_mode="zeros")
net = ScoreUNet(channels=6, spatial=2, activation=torch.nn.SiLU, **cfg).to(dev)
x = torch.randn(4, 6, 32, 32, device=dev)
for wire in ("bf16", None):
    tr = Trainer(net, precision="bf16", ema_rates=(), allreduce_dtype=wire, bucket_mb=0.05)
    tr.eng._wg_stream = None
    os.environ["C2W_COMM_STREAM"] = "1"
    tr.step(x); torch.cuda.synchronize()
    E = tr.eng.side_stream()
    pattern(E, f"b1 engine side stream after ONE trainer step (wire {wire})")
    pattern(E, f"b2 the same stream, pattern WITHOUT wait_stream (wire {wire})", wait=False)
    pattern(S, f"b3 the bisect stream S again (wire {wire})")
    print("   engine stream id", E.cuda_stream, "S id", S.cuda_stream, "priority", E.priority, S.priority, flush=True)
# is it the engine's way of making the stream?  the same call, outside any step
E2 = streams.independent_stream(dev, priority=0)
pattern(E2, "c1 another independent_stream(dev, priority=0) made now")
with torch.cuda.stream(E2):
    net.precision = "bf16"
tr2 = Trainer(net, precision="bf16", ema_rates=(), allreduce_dtype="bf16", bucket_mb=0.05)
tr2.eng._wg_stream = E2
tr2.step(x); torch.cuda.synchronize()
pattern(E2, "c2 that stream after serving as the trainer's communication stream for one step")
dist.destroy_process_group()
