set -x
mkdir -p gpurun_out/r3j
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r3j/gpu_tests.log 2>&1
echo rc=$?
timeout -k 10 200 python bench.py --no-cpu-baseline --channel-memory 33 --steps 100 --warmup 10 > gpurun_out/r3j/L33_bench.json 2>&1
timeout -k 10 200 python bench.py --no-cpu-baseline --workload cfg4 --steps 30 --warmup 5 > gpurun_out/r3j/cfg4_bench.json 2>&1
echo done
