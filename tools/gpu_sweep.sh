#!/bin/bash
mkdir -p gpurun_out
: > gpurun_out/sweep.log
for gm in 16 32 48 64; do for rm in 32 48 58 64; do
  echo "GEN_MIN=$gm READY_MIN=$rm" >> gpurun_out/sweep.log
  VS_GEN_MIN=$gm VS_READY_MIN=$rm timeout -k 10 60 python tools/quick_bench.py 3 65536 2 2>&1 | grep -E "exact/synth|fma/synth" >> gpurun_out/sweep.log
done; done
cat gpurun_out/sweep.log
