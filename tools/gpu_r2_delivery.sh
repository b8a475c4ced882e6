#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_delivery.py tests/test_gpu_custom.py -m gpu -x -q > gpurun_out/pytest_delivery.log 2>&1
rc=$?; tail -25 gpurun_out/pytest_delivery.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python tools/host_path_bench.py > gpurun_out/host_path.log 2>&1 || { tail -5 gpurun_out/host_path.log; exit 1; }
cat gpurun_out/host_path.log
