#!/bin/bash
# sweep of the generator's round-start knobs (vs_tuning.gen_min / gen_low) for a given role count
cd "$(dirname "$0")/.."
roles=${1:-3}; cfg=${2:-3}; lanes=${3:-65536}
for gm in ${GEN_MINS:-32 48 56 64}; do for gl in ${GEN_LOWS:-48 72 96}; do
  echo -n "roles $roles gen_min $gm gen_low $gl: "
  VS_DEBUG_TUNING=1 VS_WS_ROLES=$roles VS_GEN_MIN=$gm VS_GEN_LOW=$gl timeout -k 10 120 python tools/quick_bench.py $cfg $lanes 4 | grep -E "exact/synth|fma/synth" | awk '{printf "%s %s ms   ", $1, $2}'; echo
done; done
