# one rocprofv3 pass per counter set; FETCH_SIZE and WRITE_SIZE are separate sets (together they exceed the TCC slots and
# rocprofv3 aborts with "Request exceeds the capabilities of the hardware")
set -e
R=$PWD
cd /tmp && export TMPDIR=/tmp && cd $R
export CNERF_PRECISION=fp16x3
i=0
for C in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $C -d gpurun_out/pmc_h3_$i -o x --output-format csv -- python3 scripts/profile_workload.py 2 2 > gpurun_out/pmc_h3_$i.log 2>&1 || echo "set $i failed"
done
ls gpurun_out/pmc_h3_*/ | head -30
