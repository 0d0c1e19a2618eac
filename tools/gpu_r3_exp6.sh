#!/bin/bash
# PMC: MFMA busy and clock of the taps kernel in the many-tiles regime (batch 256 = 4 MC lanes worth of tiles)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R="$PWD"; cd /tmp
for sh in 128,128,28,1,3 64,64,56,1,3; do
  tag=$(echo $sh | tr , _)
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
             "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_MFMA" \
             "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $set --kernel-trace -d "$R/gpurun_out/r3_pmc${i}_$tag" -o pmc -- python "$R/tools/gpu_diag.py" one --throughput-plan --prec bf16 --iters 6 --bs 256 --shape $sh > "$R/gpurun_out/r3_pmc${i}_$tag.log" 2>&1; echo "pmc$i $tag rc=$?"
  done
done
cd "$R"
python tools/pmc_report.py 'gpurun_out/r3_pmc*/pmc_results.db' --kernel taps > gpurun_out/r3_pmc_bs256.txt 2>&1
cat gpurun_out/r3_pmc_bs256.txt
