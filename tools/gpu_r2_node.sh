#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_node.py tests/test_gpu_delivery.py -m gpu -x -q > gpurun_out/pytest_node.log 2>&1
rc=$?; tail -25 gpurun_out/pytest_node.log; [ $rc -ne 0 ] && exit $rc
exit 0
