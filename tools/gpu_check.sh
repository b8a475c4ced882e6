#!/bin/bash
# First-look GPU run: parity tests, then a quick timing of config 3.
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
rc=$?
tail -30 gpurun_out/pytest_gpu.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python tools/quick_bench.py 3 65536 5 > gpurun_out/quick_bench.log 2>&1
rc=$?
cat gpurun_out/quick_bench.log
exit $rc
