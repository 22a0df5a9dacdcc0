#!/bin/bash
# rocprofv3 PMC passes over the TRAINING STEP itself (bench.py, the fused epilogues the step really runs): one counter group per pass,
# --pmc only together with --kernel-trace (MI355X_MICROARCH.md, "rocprofv3 PMC slots": FETCH_SIZE and WRITE_SIZE do not fit one pass).
# usage: TAG=r03 bash tools/pmc_step.sh     -> gpurun_out/pmc_step_$TAG/{fetch,write,mfma}/...  then tools/pmc_step_summary.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${TAG:-r04}
declare -A groups=( [fetch]="FETCH_SIZE" [write]="WRITE_SIZE" [mfma]="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" )
for g in fetch write mfma; do
  timeout -s KILL 400 rocprofv3 --kernel-trace --pmc ${groups[$g]} --output-format csv -d gpurun_out/pmc_step_$TAG/$g -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/pmc_step_${TAG}_$g.log 2>&1
  echo "pass $g rc=$?"
done
