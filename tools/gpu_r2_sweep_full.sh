#!/bin/bash
# thresholds of the wave-specialised kernel on the FULL grid (config 3, 65536 utterances), shipped kernels:
# generator start threshold (gen_min), super-step threshold (ready_min), ring slots.  One box, interleaved.
mkdir -p gpurun_out
export VS_DEBUG_TUNING=1
: > gpurun_out/sweep_full.log
for rep in 1 2; do
for spec in "0 0 0" "16 0 0" "24 0 0" "48 0 0" "64 0 0" "0 48 0" "0 56 0" "0 64 0" "0 0 240" "0 0 264" "0 0 216" "32 64 264"; do
  set -- $spec
  echo "rep=$rep GEN_MIN=$1 READY_MIN=$2 RING=$3" >> gpurun_out/sweep_full.log
  VS_GEN_MIN=$1 VS_READY_MIN=$2 VS_RING_SLOTS=$3 timeout -k 10 90 python tools/quick_bench.py 3 65536 5 2>&1 | grep -E "exact/synth|fma/synth" >> gpurun_out/sweep_full.log
done; done
cat gpurun_out/sweep_full.log
