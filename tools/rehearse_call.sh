# GPU box: (1) parity diagnostic, (2) the multi-rank bench tests, (3) the gloo device-tensor stall with torch kernels only (last: it is expected to stall)
set -x
mkdir -p gpurun_out/r3b
TR="python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533"
timeout -k 10 300 python tools/diag_cx_eq_parity.py > gpurun_out/r3b/diag_cx_eq.txt 2>&1 &&
timeout -k 10 900 python -u -m pytest tests/test_gpu_bench_multirank.py -x -v --timeout 450 --timeout-method thread > gpurun_out/r3b/multirank_tests.log 2>&1 &&
timeout -k 10 100 $TR tools/gloo_cuda_repro.py --mode host > gpurun_out/r3b/repro_host.log 2>&1 &&
timeout -k 10 100 $TR tools/gloo_cuda_repro.py --mode cuda > gpurun_out/r3b/repro_cuda.log 2>&1
echo rc=$?
