#!/bin/bash
bash tools/gpu_ab.sh base split6 split12 split18
echo "== prio -1 (none) with base lib"
for rep in 1 2; do VS_LIB=libvoicesynth_base.so VS_DEBUG_TUNING=1 VS_WS_PRIO=0 timeout -k 10 120 python tools/quick_bench.py 3 65536 5 | grep -E "exact/synth|fma/synth"; done
