#!/bin/bash
# A/B of the two-role and the three-role wave-specialised kernel (and of the timing-only builds, if present)
# inside one box:  tools/ab_roles.sh [config] [lanes]
cd "$(dirname "$0")/.."
cfg=${1:-3}; lanes=${2:-65536}
for rep in 1 2; do
 for lib in libvoicesynth.so libvoicesynth_gonly.so libvoicesynth_fonly.so; do
  [ -f voice_synth_amd/lib/$lib ] || continue
  for roles in 2 3; do
   echo "== rep $rep $lib roles $roles"
   VS_LIB=$lib VS_DEBUG_TUNING=1 VS_WS_ROLES=$roles timeout -k 10 120 python tools/quick_bench.py $cfg $lanes 5 | grep -E "exact/synth|fma/synth"
  done
 done
done
