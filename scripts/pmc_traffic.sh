set -e
R=$PWD
cd /tmp && export TMPDIR=/tmp && cd $R
for P in fp32 fp16x3; do
  for C in FETCH_SIZE WRITE_SIZE; do
    CNERF_PRECISION=$P timeout -k 10 200 rocprofv3 --kernel-trace --pmc $C -d gpurun_out/pmc_${P}_$C -o x --output-format csv -- python3 scripts/profile_workload.py 2 2 > gpurun_out/pmc_${P}_$C.log 2>&1 || echo "$P $C failed"
  done
done
ls gpurun_out | grep pmc_fp
