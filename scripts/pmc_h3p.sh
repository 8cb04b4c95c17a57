# rocprofv3 PMC counters of the paired split-precision kernel (field_h3p_kernel, CNERF_H3_PAIRED=1) next to the single-wave one,
# plain fp16x3 forward at batch 2; one pass per counter set.  Output: gpurun_out/pmcp_<variant>_<set>/.
set -e
R=$PWD
cd /tmp && export TMPDIR=/tmp && cd $R
export CNERF_PRECISION=fp16x3 CNERF_WORKLOAD=field
i=0
for C in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  for V in 0 1; do
    CNERF_H3_PAIRED=$V timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C -d gpurun_out/pmcp_${V}_$i -o x --output-format csv -- python3 scripts/profile_workload.py 2 2 > gpurun_out/pmcp_${V}_$i.log 2>&1 || echo "set $i variant $V failed"
  done
done
python3 - <<'PY'
import csv, glob, collections
for V in (0, 1):
    agg = collections.defaultdict(list)
    for f in sorted(glob.glob(f"gpurun_out/pmcp_{V}_*/**/x_counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            if "field_h3" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m = {k: sum(v) / len(v) for k, v in agg.items()}
    wc = m.get("SQ_WAVE_CYCLES", 1)
    print(f"paired={V}", {k: (round(v / wc, 3) if k.startswith("SQ_") and "INSTS" not in k else v) for k, v in sorted(m.items())})
PY
