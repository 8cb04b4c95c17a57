# rocprofv3 PMC counters of the unfused lookup (cnerf_gather_features) in its two forms -- CNERF_GATHER_HINT=0: gather_kernel, point by
# point in ray order; 1: gather_box_kernel, 4x4x2 patches with every distinct corner line fetched once into LDS -- to see what bounds it: texture
mkdir -p gpurun_out/r3
# addresser, vector L1 (hits, pending-miss stalls), L2.  One pass per counter set, batch 2 at 128x128x(64+64).
# scripts/pmc_gather.py renders gpurun_out/r3/pmc_gather.md
R=$PWD
cd /tmp && export TMPDIR=/tmp && cd $R
export CNERF_WORKLOAD=unfused CNERF_PRECISION=fp32
i=0
for C in "GRBM_GUI_ACTIVE TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES" \
         "TCP_PERF_SEL_TOTAL_READ TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TCP_READ_TAGCONFLICT_STALL_CYCLES" \
         "TCP_TCC_READ_REQ_LATENCY TCP_TCP_LATENCY TCP_TA_TCP_STATE_READ TCP_GATE_EN1" \
         "TCC_HIT TCC_MISS TCC_READ TCC_BUSY" \
         "TA_FLAT_READ_WAVEFRONTS TA_TOTAL_WAVEFRONTS TCP_TCP_TA_ADDR_STALL_CYCLES TCP_LFIFO_STALL_CYCLES" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  for M in 0 1; do
    CNERF_GATHER_HINT=$M timeout -k 10 200 rocprofv3 --kernel-trace --pmc $C -d gpurun_out/r3/pmcg_${M}_$i -o x --output-format csv -- python3 scripts/profile_workload.py 2 3 > gpurun_out/r3/pmcg_${M}_$i.log 2>&1 || echo "set $i mode $M failed"
  done
done
python3 scripts/pmc_gather.py > gpurun_out/r3/pmc_gather.md
