# Round-3 evidence run on the GPU box (one call): kernel-trace stats of the bench command and of the training step, PMC traffic of
# the shipped kernels, PMC counter table, parity and gradient reports.  Summaries are copied into profiles/ by hand afterwards.
mkdir -p gpurun_out/r3
R=$PWD
cd /tmp && export TMPDIR=/tmp && cd $R
rocprofv3 --kernel-trace --stats -d gpurun_out/r3/prof_bench -o x --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-train-step --no-gan-step > gpurun_out/r3/bench_under_rocprof.json 2> gpurun_out/r3/bench_under_rocprof.err
echo "bench under rocprof rc $?"
rocprofv3 --kernel-trace --stats -d gpurun_out/r3/prof_train -o x --output-format csv -- python3 scripts/profile_backward.py 8 fp16x3 fp16 3 > gpurun_out/r3/prof_train.log 2>&1
echo "train profile rc $?"
bash scripts/pmc_traffic.sh > gpurun_out/r3/pmc_traffic.log 2>&1
echo "pmc traffic rc $?"
cp profiles/pmc_traffic.json gpurun_out/r3/pmc_traffic.json
bash scripts/pmc_kernels.sh > gpurun_out/r3/pmc_kernels.log 2>&1
echo "pmc kernels rc $?"
python scripts/parity_report.py > gpurun_out/r3/parity_report.md 2> gpurun_out/r3/parity_report.err
python scripts/grad_report.py > gpurun_out/r3/grad_report.md 2> gpurun_out/r3/grad_report.err
echo done
