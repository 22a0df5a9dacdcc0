"""Does a 4-byte D2H copy on a side stream overtake a long kernel on the main stream?  (score.py::_LossScalar)"""
import time, torch
dev = torch.device("cuda", 0)
x = torch.ones((), device=dev) * 3.0
pinned = torch.empty((), dtype=torch.float32, pin_memory=True)
torch.cuda.synchronize()
t0 = time.perf_counter(); torch.cuda._sleep(int(1e8)); torch.cuda.synchronize(); print("sleep(1e8) takes %.3f s" % (time.perf_counter() - t0))
import sys
PRIO = int(sys.argv[1]) if len(sys.argv) > 1 else 0
sides = [torch.cuda.Stream(device=dev, priority=PRIO) for _ in range(12)]
print("priority", PRIO, "range", torch.cuda.Stream.priority_range())
for i, side in enumerate(sides):
    ev = torch.cuda.Event(); ev.record()
    torch.cuda._sleep(int(1e8))
    t0 = time.perf_counter()
    side.wait_event(ev)
    with torch.cuda.stream(side):
        pinned.copy_(x, non_blocking=True)
    side.synchronize()
    took = time.perf_counter() - t0
    busy = not torch.cuda.current_stream().query()
    torch.cuda.synchronize()
    print("side stream %d: copy returned after %.4f s, main still busy: %s, value %.1f" % (i, took, busy, pinned.item()))
# a kernel instead of a copy engine transfer: write into host-mapped memory
for i, side in enumerate(sides[:3]):
    ev = torch.cuda.Event(); ev.record()
    torch.cuda._sleep(int(1e8))
    t0 = time.perf_counter()
    side.wait_event(ev)
    with torch.cuda.stream(side):
        y = x.clone()
    side.synchronize()
    took = time.perf_counter() - t0
    busy = not torch.cuda.current_stream().query()
    torch.cuda.synchronize()
    print("side stream %d: device clone returned after %.4f s, main still busy: %s" % (i, took, busy))
