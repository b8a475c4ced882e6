#!/bin/bash
# interleaved A/B of library builds in ONE box (VS_LIB selects the .so; boxes differ by up to 10 %,
# so variants are only ever compared within one call):  tools/gpu_ab.sh base sleep2 pub4 ...
mkdir -p gpurun_out
for rep in 1 2 3; do for v in "$@"; do
  echo "== rep $rep $v"; VS_LIB=libvoicesynth_$v.so timeout -k 10 120 python tools/quick_bench.py ${AB_CONFIG:-3} ${AB_LANES:-65536} 5 | grep -E "exact/synth|fma/synth"
done; done
