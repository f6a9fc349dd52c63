set -x
mkdir -p gpurun_out/r3f
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r3f/gpu_tests.log 2>&1 || exit 1
STEPS=30 bash tools/profile_step.sh gpurun_out/r3f/step || exit 1
timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/r3f/step_bench.json 2> gpurun_out/r3f/step_bench.err || exit 1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3f/sweep_stats -o sweep -- python3 bench.py --workload sweep --steps 200 --warmup 20 > gpurun_out/r3f/sweep_bench_profiled.json 2> gpurun_out/r3f/sweep.err || exit 1
find gpurun_out/r3f/sweep_stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/r3f/sweep_kernel_stats.csv \;
rm -rf gpurun_out/r3f/sweep_stats
timeout -k 10 200 python bench.py --workload sweep > gpurun_out/r3f/sweep_bench.json 2>> gpurun_out/r3f/sweep.err || exit 1
echo done
