#!/bin/bash
# One development iteration on the GPU box: parity tests, then timings of config 3 (and optionally more).
#   tools/gpu_iter.sh [pytest-args...]      default: all -m gpu tests
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ "$1" != "--notest" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q "$@" > gpurun_out/pytest_gpu.log 2>&1
  rc=$?; tail -15 gpurun_out/pytest_gpu.log; [ $rc -ne 0 ] && exit $rc
fi
timeout -k 10 300 python tools/quick_bench.py 3 65536 > gpurun_out/qb3.log 2>&1 || { tail -5 gpurun_out/qb3.log; exit 1; }
cat gpurun_out/qb3.log
timeout -k 10 300 python tools/diag_bench.py 3 65536 > gpurun_out/diag3.log 2>&1 || { tail -5 gpurun_out/diag3.log; exit 1; }
head -20 gpurun_out/diag3.log
