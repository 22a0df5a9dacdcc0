#!/usr/bin/env python3
"""Summarise tools/pmc_step.sh: per (kernel, grid size) mean counter values per launch + mean duration, for the kernels matching a
substring; FETCH_SIZE is doubled (gfx950: the counter tallies 128-B requests at 64 B, MI355X_MICROARCH.md "HBM") and both sizes are
converted from KB to bytes.  Writes JSON to stdout.
    python tools/pmc_step_summary.py gpurun_out/pmc_step_r03 conv_patch_t3 wgrad_patch"""
import collections, csv, glob, json, sys
root, pats = sys.argv[1], sys.argv[2:]
out = collections.defaultdict(dict)
for g in ("fetch", "write", "mfma"):
    for f in sorted(glob.glob(f"{root}/{g}/**/*counter_collection.csv", recursive=True)):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if not any(p in name for p in pats):
                continue
            key = (name.split("(")[0].replace("void (anonymous namespace)::", ""), int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1))
            agg[(key, r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (key, cname), v in agg.items():
            out[key][cname] = sum(v) / len(v)
            out[key]["launches_" + g] = len(v)
    for f in sorted(glob.glob(f"{root}/{g}/**/*kernel_trace.csv", recursive=True))[:1]:
        dur = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if not any(p in name for p in pats):
                continue
            wg = int(r["Workgroup_Size"]) if "Workgroup_Size" in r else int(r.get("Workgroup_Size_X", 1))
            key = (name.split("(")[0].replace("void (anonymous namespace)::", ""), int(r["Grid_Size"]) // max(wg, 1))
            dur[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for key, v in dur.items():
            out[key]["duration_us_" + g] = sum(v) / len(v) / 1e3
res = []
for (name, wgs), d in sorted(out.items(), key=lambda kv: -kv[1].get("duration_us_fetch", 0) * kv[1].get("launches_fetch", 0)):
    e = dict(kernel=name, workgroups=wgs, **{k: round(v, 3) for k, v in d.items()})
    if "FETCH_SIZE" in d:
        e["hbm_read_bytes"] = round(2 * d["FETCH_SIZE"] * 1024)
    if "WRITE_SIZE" in d:
        e["hbm_write_bytes"] = round(d["WRITE_SIZE"] * 1024)
    if "hbm_read_bytes" in e and "hbm_write_bytes" in e:
        e["hbm_bytes"] = e["hbm_read_bytes"] + e["hbm_write_bytes"]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "GRBM_GUI_ACTIVE" in d:
        e["mfma_busy"] = round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * d["GRBM_GUI_ACTIVE"] / 8), 4)
        e["clock_ghz"] = round(d["GRBM_GUI_ACTIVE"] / 8 / (d["duration_us_mfma"] * 1e3), 3)
    res.append(e)
print(json.dumps(res, indent=1))
