# HBM-side traffic of the render kernels by rocprofv3 PMC counters: ONE counter per pass (FETCH_SIZE and WRITE_SIZE do not
# fit the TCC slots together: MI355X_MICROARCH.md "rocprofv3 PMC slots"), batch 2 at 128x128x64, both forward precisions.
# Writes gpurun_out/pmc_<precision>_<counter>/ ; scripts/pmc_traffic.py turns them into profiles/pmc_traffic.json.
set -e
R=$PWD
cd /tmp && export TMPDIR=/tmp && cd $R
for P in fp32 fp16x3 unfused; do
  for C in FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum; do
    if [ $P = unfused ]; then export CNERF_WORKLOAD=unfused; PP=fp32; else export CNERF_WORKLOAD=field; PP=$P; fi
    CNERF_PRECISION=$PP timeout -k 10 200 rocprofv3 --kernel-trace --pmc $C -d gpurun_out/pmc_${P}_$C -o x --output-format csv -- python3 scripts/profile_workload.py 2 2 > gpurun_out/pmc_${P}_$C.log 2>&1 || echo "$P $C failed"
  done
done
python3 scripts/pmc_traffic.py
