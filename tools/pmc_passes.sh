# rocprofv3 PMC passes for the kernels (TAG names the output directory: gpurun_out/pmc_$TAG) (one counter group per pass, each under its own timeout).
# usage: bash tools/pmc_${TAG:-r01k}.sh <conv|wgrad> <group indices...>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
kind=$1; shift
groups=("SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum")
for i in "$@"; do
  timeout -s KILL 150 rocprofv3 --kernel-trace --pmc ${groups[$i]} --output-format csv -d gpurun_out/pmc_${TAG:-r01k}/$kind/g$i -- python3 tools/bench_kernels.py --batch 128 --dtypes bf16 --only 0 --kind $kind --iters 3 --act 0 > gpurun_out/pmc_${TAG:-r01k}_${kind}_g$i.log 2>&1
  echo "group $i rc=$?"
done
