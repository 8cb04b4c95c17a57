# rocprofv3 PMC counters of the hot kernels, one pass per counter set (FETCH_SIZE and WRITE_SIZE separately: together they
# exceed the TCC slots).  Workloads: the plain forwards (fp32, fp16x3) at batch 2, and one forward + half-precision backward at
# batch 2 (fp16x3 forward keeping its activations, chain16, weight_grad16).  scripts/pmc_table.py renders the table.
set -e
R=$PWD
cd /tmp && export TMPDIR=/tmp && cd $R
i=0
for C in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  for W in fwd32 fwd16 bwd16; do
    if [ $W = fwd32 ]; then export CNERF_PRECISION=fp32 CNERF_WORKLOAD=field; CMD="python3 scripts/profile_workload.py 2 2"; fi
    if [ $W = fwd16 ]; then export CNERF_PRECISION=fp16x3 CNERF_WORKLOAD=field; CMD="python3 scripts/profile_workload.py 2 2"; fi
    if [ $W = bwd16 ]; then CMD="python3 scripts/profile_backward.py 2 fp16x3 fp16 2"; fi
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C -d gpurun_out/pmck_${W}_$i -o x --output-format csv -- $CMD > gpurun_out/pmck_${W}_$i.log 2>&1 || echo "set $i $W failed"
  done
done
python3 scripts/pmc_table.py > gpurun_out/r3/kernel_counters_table.md
