#!/bin/bash
# same box: one-wave kernel forced (VS_KERNEL=single) against the default choice, per configuration
for cfg in "5 65536" "4 32768" "3 65536" "2 65536" "3 131072"; do
for rep in 1 2; do
echo "== config/lanes $cfg default"; timeout -k 10 200 python tools/quick_bench.py $cfg 3 | grep -E "plan|/synth"
echo "== config/lanes $cfg single"; VS_DEBUG_TUNING=1 VS_KERNEL=single timeout -k 10 200 python tools/quick_bench.py $cfg 3 | grep -E "plan|/synth"
done; done
