set -x
mkdir -p gpurun_out/r3e
timeout -k 10 300 python tools/diag_cx_eq_parity.py > gpurun_out/r3e/diag_cx_eq.txt 2>&1 &&
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q -s --timeout 300 --timeout-method thread > gpurun_out/r3e/gpu_tests.log 2>&1 ;
echo rc=$?
timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/r3e/bench.json 2> gpurun_out/r3e/bench.err
echo rc=$?
