#!/bin/bash
# interleaved A/B of library builds in one box (VS_LIB selects the .so)
for rep in 1 2 3; do for v in "$@"; do
  echo "== rep $rep $v"; VS_LIB=libvoicesynth_$v.so timeout -k 10 120 python tools/quick_bench.py 3 65536 5 | grep -E "exact/synth|fma/synth"
done; done
