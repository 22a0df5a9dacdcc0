#!/usr/bin/env python3
"""Per-STEP kernel table out of a rocprofv3 kernel trace of bench.py / tools/bench_module_api.py: the launches between two consecutive
optimizer kernels (adamw_ema_kernel), i.e. without the process's set-up launches that `--stats` folds into its per-kernel totals
(the 228 host->device parameter copies of net.to(device), the first packing of every weight matrix, ...).
    python tools/step_from_trace.py <kernel_trace.csv> [step index from the end, default 2]"""
import csv
import sys

path = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "adamw_ema_kernel" in r["Kernel_Name"]]
i0, i1 = marks[-back - 1], marks[-back]
span = (int(rows[i1]["End_Timestamp"]) - int(rows[i0]["End_Timestamp"])) / 1e6
agg = {}
for r in rows[i0 + 1: i1 + 1]:
    n = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
    n = n.split("(")[0] if not n.startswith("_Z") else n
    a = agg.setdefault(n[:110], [0, 0.0])
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
gemm = ("conv_patch", "wgrad_patch", "conv_igemm", "wgrad_kernel")
busy = sum(v[1] for v in agg.values()) / 1e3
non = sum(v[1] for k, v in agg.items() if not any(g in k for g in gemm)) / 1e3
print(f"{path}: step {back} from the end: {i1 - i0} launches, {span:.3f} ms between optimizer kernels, {busy:.3f} ms of kernel time "
      f"(both streams), {non:.3f} ms of it outside the implicit-GEMM kernels")
print(f"{'ms/step':>9} {'launches':>8} {'avg us':>9}  kernel")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{v[1] / 1e3:9.3f} {v[0]:8d} {v[1] / v[0]:9.1f}  {k}")
