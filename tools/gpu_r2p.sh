#!/bin/bash
# round-2 profile session: rocprofv3 kernel-trace stats of the bench command, PMC passes on the layer1 / layer2 launches
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$PWD"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/r2p_prof" -o bench -- python "$R/bench.py" --steps 9 --warmup 3 --no-cpu-baseline --no-extras --no-traffic > "$R/gpurun_out/r2p_prof.log" 2>&1; echo "prof rc=$?"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_MFMA" \
           "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  for sh in 64,64,56,1,3 128,128,28,1,3; do
    timeout 300 rocprofv3 --pmc $set --kernel-trace -d "$R/gpurun_out/r2p_pmc${i}_${sh//,/_}" -o pmc -- python "$R/tools/gpu_diag.py" one --prec bf16 --iters 6 --shape $sh > /dev/null 2>&1; echo "pmc$i $sh rc=$?"
  done
done
cd "$R"
