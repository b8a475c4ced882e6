#!/bin/bash
# parity (fast subset) + quick bench + A/B of the two fused kernels: the inner loop of kernel tuning
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
rc=$?; tail -4 gpurun_out/pytest_gpu.log; [ $rc -ne 0 ] && exit $rc
echo "== wave-specialised (default) =="
timeout -k 10 120 python tools/quick_bench.py 3 65536 3 > gpurun_out/quick_bench.log 2>&1; grep -E "synth|plan" gpurun_out/quick_bench.log
echo "== single-wave kernel =="
VS_KERNEL=single timeout -k 10 120 python tools/quick_bench.py 3 65536 3 > gpurun_out/quick_bench_single.log 2>&1; grep -E "synth|source|filter" gpurun_out/quick_bench_single.log
