#!/bin/bash
# config 4 shard (half-filled chip), three roles in the spread layout: ring depth x round-start thresholds (finer)
cd "$(dirname "$0")/.."
cfg="4 32768"
for rs in 456 528 552 600; do for gm in 48 64; do for gl in 96 120 144; do
  echo -n "config $cfg roles 3 ring $rs gen_min $gm gen_low $gl: "
  VS_DEBUG_TUNING=1 VS_WS_ROLES=3 VS_RING_SLOTS=$rs VS_GEN_MIN=$gm VS_GEN_LOW=$gl timeout -k 10 120 python tools/quick_bench.py $cfg 3 | grep -E "exact/synth|fma/synth|ring_slots" | awk '{printf "%s %s ms   ", $1, $2}'; echo
done; done; done
cfg="3 16384"
for rs in 288 408 600 900; do for gm in 16 64; do for gl in 96 144; do
  echo -n "config $cfg roles 3 ring $rs gen_min $gm gen_low $gl: "
  VS_DEBUG_TUNING=1 VS_WS_ROLES=3 VS_RING_SLOTS=$rs VS_GEN_MIN=$gm VS_GEN_LOW=$gl timeout -k 10 120 python tools/quick_bench.py $cfg 3 | grep -E "exact/synth|fma/synth" | awk '{printf "%s %s ms   ", $1, $2}'; echo
done; done; done
