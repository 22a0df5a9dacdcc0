#!/bin/bash
# MFMA-busy / clock pass over the fp16 training step (the same counters as tools/pmc_step.sh's third pass): is fp16 slower than bf16
# because the chip holds a lower clock under it?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_step_r04f16
timeout -s KILL 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_step_r04f16/mfma -- python3 bench.py --precision fp16 --steps 3 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/pmc_step_r04f16_mfma.log 2>&1
echo "pass mfma fp16 rc=$?"
python tools/pmc_step_summary.py gpurun_out/pmc_step_r04f16 conv_patch_t3 wgrad_patch > gpurun_out/r04_pmc_step_fp16_mfma.json
