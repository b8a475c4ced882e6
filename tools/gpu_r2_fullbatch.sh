#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
nproc; free -g | head -2
timeout -k 10 1100 python -m pytest tests/test_gpu_full_batches.py -m gpu -x -q -s > gpurun_out/pytest_full.log 2>&1
rc=$?; tail -15 gpurun_out/pytest_full.log; exit $rc
