"""HIP streams that really run next to each other.

A HIP stream is bound to one of a few hardware queues when it is created (four per process and priority level on this stack,
handed out in rotation), and two streams on one queue execute in order: the third, seventh, eleventh ... normal-priority stream
a process creates shares the default stream's queue (lab/probes/early_item_probe.py: a 4-byte copy on such a stream returns
after the 43-ms kernel on the default stream, on every other one after 0.3 ms).  The engine's gradient stream is usually the first
stream of its process and lands elsewhere by luck; a process that has created streams before (a DDP communicator's pool, a
sampler's window-batch streams, a test suite) may not be so lucky and would lose the whole two-stream backward without a sign.
``independent_stream`` asks the hardware instead of the creation order.
"""
from __future__ import annotations

import os
from typing import Optional

import torch

_PROBE_CYCLES = 8_000_000  # ~3.4 ms of torch.cuda._sleep on the stream that must NOT hold the candidate up


def overtakes(side: "torch.cuda.Stream", main: Optional["torch.cuda.Stream"] = None, votes: int = 3) -> bool:
    """True if work handed from ``main`` (default: the current stream) to ``side`` through an event runs NEXT TO what ``main`` goes on
    with (_overtakes_once).  One sample is three ~0.85-ms spin kernels; a host hiccup can flip one, so the verdict is the majority of up
    to ``votes`` samples (two agreeing end it)."""
    yes = no = 0
    need = votes // 2 + 1
    while yes < need and no < need:
        if _overtakes_once(side, main):
            yes += 1
        else:
            no += 1
    return yes >= need


_SPIN_MS: dict = {}  # device index -> milliseconds one _PROBE_CYCLES / 4 spin kernel takes on that chip (measured once)


def _overtakes_once(side: "torch.cuda.Stream", main: Optional["torch.cuda.Stream"] = None) -> bool:
    """The pattern the callers actually issue, timed:  main: spin T;  side waits for main's event, then spins T;  main: spin T again.
    If main's second kernel runs NEXT TO side's, main is done after 2 T; if it is held behind it, after 3 T.

    Round 6 (tools/probe_stream_overlap_*.py, profiles/r06_experiments.md): the earlier probes -- "a launch on side finishes while a
    kernel on main is still running", also in both directions with two launches each -- pass on pairs of streams that nevertheless
    SERIALISE under this pattern: once side has waited for an event of main, main's following kernel does not start before side's
    has finished (12 x (500 us + 200 us) took 8.4 ms instead of 6.2).  Which streams of a process pair up like that depends on what
    was created before them (a one-rank RCCL communicator's streams are enough to move it); nothing but timing the pattern tells."""
    main = main or torch.cuda.current_stream(side.device)
    cyc = _PROBE_CYCLES // 4
    with torch.cuda.stream(side):  # first use of a stream can take milliseconds (its hardware queue is created): not part of the verdict
        torch.cuda._sleep(1000)
    main.synchronize()
    side.synchronize()
    dkey = side.device.index
    if dkey not in _SPIN_MS:
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(main):
            torch.cuda._sleep(cyc)  # (the first spin kernel of a process also loads its code object)
            c0.record(main)
            torch.cuda._sleep(cyc)
            c1.record(main)
        main.synchronize()
        _SPIN_MS[dkey] = c0.elapsed_time(c1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(main):
        e0.record(main)
        torch.cuda._sleep(cyc)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        torch.cuda._sleep(cyc)
    with torch.cuda.stream(main):
        torch.cuda._sleep(cyc)
        e1.record(main)
    main.synchronize()
    side.synchronize()
    return e0.elapsed_time(e1) < 2.5 * _SPIN_MS[dkey]


def independent_stream(device, avoid=(), tries: int = 8, priority: int = 0) -> "torch.cuda.Stream":
    """A new stream on ``device`` that shares a hardware queue neither with the current stream nor with any stream in ``avoid``
    (the engine's gradient stream when the loss read-back stream is made, and the other way round): candidates are created one
    after the other (each takes the next queue in the rotation) until one overtakes a kernel on each of them.  During a graph
    capture, or if no candidate qualifies, the last one created is returned as it is.
    Normal priority on purpose: with one high-priority stream in the process, the THIRD engine created in it ran its two-stream
    backward at 56 instead of 47 ms/step although this probe had cleared its gradient stream (profiles/r04_experiments.md section 16)."""
    device = torch.device(device)
    with torch.cuda.device(device):
        if os.environ.get("C2W_PLAIN_STREAMS") == "1" or torch.cuda.is_current_stream_capturing():
            return torch.cuda.Stream(device=device, priority=priority)
        others, seen = [], set()
        for st in [torch.cuda.current_stream(device)] + [a for a in avoid if a is not None]:
            if st.cuda_stream not in seen:
                seen.add(st.cuda_stream)
                others.append(st)
        rejected = []  # kept alive until the end: a destroyed stream's queue slot would be handed to the next candidate again
        s, ok = None, False
        for _ in range(max(1, tries)):
            s = torch.cuda.Stream(device=device, priority=priority)
            ok = all(overtakes(s, o) for o in others)
            if ok:
                break
            rejected.append(s)
        if not ok:
            import warnings
            warnings.warn(f"climate2weather_amd: none of {max(1, tries)} new HIP streams runs next to the current one (all share its hardware "
                          "queue): launches meant to overlap (weight gradients beside input gradients, window batches) will serialise",
                          RuntimeWarning, stacklevel=2)
        if os.environ.get("C2W_STREAM_DEBUG") == "1":
            print("independent_stream: %d rejected, %d to avoid, verdict %s" % (len(rejected), len(others), ok), flush=True)
        return s
