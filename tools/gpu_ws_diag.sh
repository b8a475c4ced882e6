#!/bin/bash
mkdir -p gpurun_out
: > gpurun_out/ws_diag.log
for cfg in "32 64" "64 64" "64 48" "64 32" "60 56"; do
  set -- $cfg
  echo "== GEN_MIN=$1 READY_MIN=$2 ==" >> gpurun_out/ws_diag.log
  VS_KERNEL=ws VS_GEN_MIN=$1 VS_READY_MIN=$2 timeout -k 5 120 python tools/diag_ws.py >> gpurun_out/ws_diag.log 2>&1
done
cat gpurun_out/ws_diag.log
