# rocprofv3 PMC counters of the per-point FiLM family's fp16 kernels (TALLSIREN, batch 2, 128x128x(64+64), H 256): plain and storing
# forward, gradient chain (chain and dry run), weight_grad16 -- one pass per counter set; scripts/pmc_table.py renders the table.
set -e
R=$PWD
cd /tmp && export TMPDIR=/tmp && cd $R
i=0
for C in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C -d gpurun_out/pmck_pfilm_$i -o x --output-format csv -- python3 scripts/pfilm_probe.py 2 fp16 > gpurun_out/pmck_pfilm_$i.log 2>&1 || echo "set $i failed"
done
mkdir -p gpurun_out/r3
python3 scripts/pmc_table.py > gpurun_out/r3/kernel_counters_table_pfilm.md
