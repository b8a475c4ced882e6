#!/bin/bash
# parity (fast subset) + quick bench + phase breakdown: the inner loop of kernel tuning
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
rc=$?; tail -4 gpurun_out/pytest_gpu.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 120 python tools/quick_bench.py 3 65536 3 > gpurun_out/quick_bench.log 2>&1; grep -E "synth|source|filter|plan" gpurun_out/quick_bench.log
timeout -k 10 120 python tools/diag_bench.py 3 65536 > gpurun_out/diag.log 2>&1; head -10 gpurun_out/diag.log
