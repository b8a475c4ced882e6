#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s > gpurun_out/pytest_gpu.log 2>&1
rc=$?; grep -E "fma vs exact|passed|failed|Error" gpurun_out/pytest_gpu.log | tail -10; exit $rc
