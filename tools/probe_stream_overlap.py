"""What does it take for work on a second HIP stream to run NEXT TO the compute stream on this stack?  Four patterns, 12 rounds each of
~500 us of "backward" on the compute stream and ~200 us of "collective" on a side stream (streams.independent_stream: another hardware
queue), timed end to end with events (serial: ~8.4 ms, overlapped: ~6.2 ms):
  A  spin kernels, no dependency between the streams at all
  B  spin kernels, side.wait_stream(compute) before every side launch (the trainer's per-bucket hand-over)
  C  like B with the hand-over by ONE event recorded once per round and waited for (explicit torch.cuda.Event)
  D  like B with REAL kernels: an 8192^3 bf16 matmul slice on compute (~500 us), a 100-MB copy on the side stream (~40 us x 5)
"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from climate2weather_amd import streams
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
cur = torch.cuda.current_stream()
side = streams.independent_stream(dev)
print("side stream overtakes the compute stream:", streams.overtakes(side), flush=True)
US = 2350


def timed(name, body):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    body()
    cur.wait_stream(side)
    e1.record()
    torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1):.2f} ms", flush=True)


def A():
    for _ in range(12):
        torch.cuda._sleep(500 * US)
        with torch.cuda.stream(side):
            torch.cuda._sleep(200 * US)


def B():
    for _ in range(12):
        torch.cuda._sleep(500 * US)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            torch.cuda._sleep(200 * US)


def C():
    for _ in range(12):
        torch.cuda._sleep(500 * US)
        ev = torch.cuda.Event()
        ev.record(cur)
        side.wait_event(ev)
        with torch.cuda.stream(side):
            torch.cuda._sleep(200 * US)


a = torch.randn(8192, 8192, device=dev).bfloat16()
b = torch.randn(8192, 2048, device=dev).bfloat16()
src = torch.empty(100 << 20, dtype=torch.uint8, device=dev)
dst = torch.empty_like(src)
for _ in range(3):
    torch.matmul(a, b)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(12):
    torch.matmul(a, b); torch.matmul(a, b)
e1.record(); torch.cuda.synchronize()
t_mm = e0.elapsed_time(e1)
e0.record()
for _ in range(12):
    for _ in range(5):
        dst.copy_(src)
e1.record(); torch.cuda.synchronize()
t_cp = e0.elapsed_time(e1)
print(f"D's parts alone: 24 matmuls {t_mm:.2f} ms, 60 copies of 100 MB {t_cp:.2f} ms (serial sum {t_mm + t_cp:.2f} ms)", flush=True)


def D():
    for _ in range(12):
        torch.matmul(a, b); torch.matmul(a, b)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(5):
                dst.copy_(src)


for name, fn in (("A spin kernels, independent streams", A), ("B spin kernels, side.wait_stream(compute) per round", B),
                 ("C spin kernels, explicit event per round", C), ("D matmuls on compute, copies on the side stream, wait_stream per round", D)):
    fn()  # warm
    torch.cuda.synchronize()
    timed(name, fn)
