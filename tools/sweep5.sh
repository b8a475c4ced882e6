#!/bin/bash
# config 5 over mixed rings: ring floor x round-start knobs (VS_MIXED_RINGS: 1 = the library's floor, n >= 48 = that floor)
#   tools/sweep5.sh ["floors"] ["gen_mins"] ["gen_lows"]
for fl in ${1:-1 192 240}; do for gm in ${2:-64 48}; do for gl in ${3:-32 48 72}; do
  echo -n "floor $fl gen_min $gm gen_low $gl: "
  VS_DEBUG_TUNING=1 VS_MIXED_RINGS=$fl VS_GEN_MIN=$gm VS_GEN_LOW=$gl timeout -k 10 120 python tools/quick_bench.py 5 65536 6 | grep -E "exact/synth|fma/synth|plan create" | awk '{printf "%s %s   ", $1, $2}'; echo
done; done; done
