#!/bin/bash
mkdir -p gpurun_out
: > gpurun_out/sweep.log
for lanes in 16384 32768; do
for gm in 8 24 40 64; do for rm in 40 64; do for rs in 312 432; do
  echo "lanes=$lanes GEN_MIN=$gm READY_MIN=$rm RING=$rs" >> gpurun_out/sweep.log
  VS_RING_SLOTS=$rs VS_GEN_MIN=$gm VS_READY_MIN=$rm timeout -k 10 60 python tools/quick_bench.py 3 $lanes 2 2>&1 | grep -E "exact/synth" >> gpurun_out/sweep.log
done; done; done; done
cat gpurun_out/sweep.log
