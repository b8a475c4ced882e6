#!/bin/bash
# after a kernel change: full parity (every test incl. the every-sample batches), then A/B against the previous build
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
rc=$?; tail -6 gpurun_out/pytest_gpu.log; [ $rc -ne 0 ] && exit $rc
for rep in 1 2 3; do for v in libvoicesynth_base.so libvoicesynth.so; do
  echo "== rep $rep $v"; VS_LIB=$v timeout -k 10 120 python tools/quick_bench.py 3 65536 5 | grep -E "exact/synth|fma/synth"
done; done
for cfg in "5 65536" "4 32768" "2 65536"; do for v in libvoicesynth_base.so libvoicesynth.so; do
  echo "== config $cfg $v"; VS_LIB=$v timeout -k 10 120 python tools/quick_bench.py $cfg 3 | grep -E "exact/synth|fma/synth"
done; done
