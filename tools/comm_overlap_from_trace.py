#!/usr/bin/env python3
"""Do the gradient all-reduce buckets of a training step run NEXT TO its backward?  Reads a rocprofv3 kernel trace of
`C2W_FORCE_DIST=1 python3 bench.py ...` (one rank through a real RCCL communicator) and prints, for ONE step (the launches between
two consecutive optimizer kernels): the kernels that ran on another hardware queue than the conv kernels (RCCL's collective kernels,
the wire-format casts, the chased update), for each the conv / weight-gradient kernels it overlapped in time, and the summary line
the round-6 verdict item asks for: bucket kernels overlapping the backward, and how long the compute queue sat idle inside the step.
    python tools/comm_overlap_from_trace.py <kernel_trace.csv> [step index from the end, default 2]"""
import csv
import sys

path = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "adamw_ema_kernel" in r["Kernel_Name"]]
i0, i1 = marks[-back - 1], marks[-back]
step = rows[i0 + 1: i1 + 1]
t0 = int(rows[i0]["End_Timestamp"])
span = (int(rows[i1]["End_Timestamp"]) - t0) / 1e6
GEMM = ("conv_patch", "wgrad_patch", "conv_igemm", "wgrad_kernel", "wgrad_group")


def short(n):
    n = n.replace("void ", "").replace("(anonymous namespace)::", "")
    return (n.split("(")[0] if not n.startswith("_Z") else n)[:90]


# the compute stream = the one most conv kernels ran on (rocprofv3 reports ONE Queue_Id for every HIP stream of the process on this
# stack -- the column is the HSA queue of the profiler's interception, not the stream's hardware queue -- so streams are told apart
# by Stream_Id, and what "next to each other" means is read off the timestamps)
KEY = "Stream_Id" if len({r["Stream_Id"] for r in step}) > 1 else "Queue_Id"
qcount = {}
for r in step:
    if any(g in r["Kernel_Name"] for g in GEMM):
        qcount[r[KEY]] = qcount.get(r[KEY], 0) + 1
main_q = max(qcount, key=qcount.get)
comp = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in step if r[KEY] == main_q]
side = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r[KEY]) for r in step if r[KEY] != main_q]
print(f"{path}: step {back} from the end: {len(step)} launches in {span:.3f} ms; compute stream ({KEY} {main_q}): {len(comp)} launches, "
      f"other streams: {len(side)} launches")
busy = sum(e - s for s, e, _ in comp) / 1e6
idle = span - busy
print(f"compute stream busy {busy:.3f} ms, idle {idle:.3f} ms of the step")
tot = ov = 0.0
n_ov = 0
first_bwd = next((s for s, e, n in comp if "wgrad" in n), None)
print(f"{'start ms':>9} {'us':>8} {'overlap us':>10} {'queue':>5}  kernel  [compute kernels it ran beside]")
for s, e, n, q in side:
    o = 0
    names = []
    for cs, ce, cn in comp:
        if ce <= s:
            continue
        if cs >= e:
            break
        d = min(e, ce) - max(s, cs)
        if d > 0:
            o += d
            names.append(cn.split("<")[0])
    tot += (e - s) / 1e3
    ov += o / 1e3
    n_ov += o > 0
    print(f"{(s - t0) / 1e6:9.3f} {(e - s) / 1e3:8.1f} {o / 1e3:10.1f} {q:>5}  {n}  [{', '.join(sorted(set(names)))}]")
in_bwd = [x for x in side if first_bwd is not None and x[0] >= first_bwd]
print(f"SUMMARY: {len(side)} kernels on other streams ({tot / 1e3:.3f} ms), {n_ov} of them overlap compute kernels ({ov / 1e3:.3f} ms overlapped = "
      f"{100.0 * ov / max(tot, 1e-9):.1f} %); {len(in_bwd)} started after the backward's first weight-gradient launch; step {span:.3f} ms")
