#!/bin/bash
# tools/insts.sh [config] [lanes] [samples] -- variants...   dynamic instruction counts per sample, one pass per library variant
# ("default" = libvoicesynth.so, else libvoicesynth_<v>.so); on the GPU box (through tools/gpurun.sh)
set -o pipefail
cfg=${1:-3}; lanes=${2:-65536}; ns=${3:-16000}; shift 3
mkdir -p gpurun_out; export TMPDIR=/tmp
root=$(pwd)
for v in "$@"; do
  lib=libvoicesynth_$v.so; [ "$v" = default ] && lib=libvoicesynth.so
  rm -rf gpurun_out/insts_$v
  ( cd /tmp && VS_LIB=$lib timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $root/gpurun_out/insts_$v -o p -- python3 $root/tools/quick_bench.py $cfg $lanes 3 > $root/gpurun_out/insts_$v.log 2>&1 ) || { tail -5 gpurun_out/insts_$v.log; exit 1; }
  echo "== $v (config $cfg)"
  python3 tools/insts_probe.py gpurun_out/insts_$v/p_counter_collection.csv $lanes $ns
  grep -E "exact/synth|fma/synth" gpurun_out/insts_$v.log
done
