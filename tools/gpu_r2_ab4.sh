#!/bin/bash
echo "== config 5 single"; VS_DEBUG_TUNING=1 VS_KERNEL=single timeout -k 10 200 python tools/quick_bench.py 5 65536 3 | grep -E "/synth"
for rm in 24 32 40 48 56 64; do for gm in 16 32; do
echo "== config 5 ws ready_min=$rm gen_min=$gm"; VS_DEBUG_TUNING=1 VS_READY_MIN=$rm VS_GEN_MIN=$gm timeout -k 10 200 python tools/quick_bench.py 5 65536 3 | grep -E "/synth"
done; done
echo "== config 5 single"; VS_DEBUG_TUNING=1 VS_KERNEL=single timeout -k 10 200 python tools/quick_bench.py 5 65536 3 | grep -E "/synth"
