#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
rc=$?; tail -5 gpurun_out/pytest_gpu.log; [ $rc -ne 0 ] && exit $rc
for cfg in "3 65536" "5 65536" "4 32768" "2 65536" "2 1024"; do
echo "== config/lanes $cfg"; timeout -k 10 200 python tools/quick_bench.py $cfg 3 | grep -E "plan|/synth|/source|/filter"
done
