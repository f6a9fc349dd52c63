set -x
mkdir -p gpurun_out/r3l
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_sharded.py tests/test_gpu_complex_linear.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r3l/tests.log 2>&1
echo rc=$?
bash tools/ab_bench.sh tools/libprev.so volterra_vae_amd/libv2vae_hip.so --steps 300 --warmup 30 > gpurun_out/r3l/ab.txt 2>&1
echo done
