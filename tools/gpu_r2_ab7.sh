#!/bin/bash
for rep in 1 2 3; do for v in libvoicesynth_base.so libvoicesynth.so; do
  echo "== config 5 rep $rep $v"; VS_LIB=$v timeout -k 10 120 python tools/quick_bench.py 5 65536 5 | grep -E "exact/synth|fma/synth"
done; done
