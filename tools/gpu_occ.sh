#!/bin/bash
mkdir -p gpurun_out
for n in 65536 98304 131072 196608; do
  timeout -k 10 300 python tools/quick_bench.py 3 $n 3 2>&1 | grep -E "synth|source|plan" 
done > gpurun_out/occ.log 2>&1
cat gpurun_out/occ.log
