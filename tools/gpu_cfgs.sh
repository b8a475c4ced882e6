#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
rc=$?; tail -4 gpurun_out/pytest_gpu.log; [ $rc -ne 0 ] && exit $rc
: > gpurun_out/cfgs.log
for c in "4 32768" "5 65536" "3 32768" "3 16384"; do
  set -- $c
  for k in auto single; do
    echo "--- config $1 lanes $2 kernel=$k" >> gpurun_out/cfgs.log
    if [ $k = auto ]; then timeout -k 10 200 python tools/quick_bench.py $1 $2 3 2>&1 | grep -E "plan|exact/synth|fma/synth" >> gpurun_out/cfgs.log
    else VS_KERNEL=single timeout -k 10 200 python tools/quick_bench.py $1 $2 3 2>&1 | grep -E "plan|exact/synth|fma/synth" >> gpurun_out/cfgs.log; fi
  done
done
cat gpurun_out/cfgs.log
