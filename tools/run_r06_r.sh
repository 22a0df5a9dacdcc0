#!/bin/bash
# PMC passes on the 512 -> 512 @8x8 conv (B = 128: 256 workgroups, paired images) for the 4-wave and the 8-wave 8x16-tile kernel
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06r
mkdir -p $O
groups=("SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA")
for m in half8 half4; do
  [ $m = half4 ] && export C2W_NO_HALF8=1 || unset C2W_NO_HALF8
  for i in 0 1 2 3 4; do
    timeout -s KILL 150 rocprofv3 --kernel-trace --pmc ${groups[$i]} --output-format csv -d $O/$m/g$i -- python3 tools/bench_kernels.py --batch 128 --dtypes bf16 --only 4 --kind conv --iters 3 --act 0 > $O/${m}_g$i.log 2>&1
    echo "$m group $i rc=$?"
  done
done
python3 - <<'PY' | tee gpurun_out/r06r/pmc_8x8_summary.txt
import csv, glob, collections
for m in ("half8", "half4"):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/r06r/{m}/g*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "conv_patch_half" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = []
    for f in glob.glob(f"gpurun_out/r06r/{m}/g0/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "conv_patch_half" in r["Kernel_Name"]:
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print(m, "launches", len(dur), "mean us %.1f" % (sum(dur) / max(len(dur), 1)))
    for k, v in sorted(agg.items()):
        print(f"   {k:28s} {sum(v) / len(v):16.0f}")
PY
rm -rf $O/half8 $O/half4
