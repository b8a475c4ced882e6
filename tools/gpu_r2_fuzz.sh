#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests/test_gpu_properties.py -m gpu -x -q -s -k fuzz > gpurun_out/pytest_fuzz.log 2>&1
rc=$?; tail -12 gpurun_out/pytest_fuzz.log; exit $rc
