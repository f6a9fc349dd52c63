set -x
mkdir -p gpurun_out/r3k
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r3k/gpu_tests.log 2>&1
echo rc=$?
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 100 --warmup 10 > gpurun_out/r3k/bench.json 2>&1
echo done
