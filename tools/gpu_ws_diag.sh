#!/bin/bash
mkdir -p gpurun_out
: > gpurun_out/ws_diag.log
for cfg in "312 48 64 48"; do
  set -- $cfg
  echo "== RING=$1 GEN_MIN=$2 READY_MIN=$3 GEN_LOW=$4 ==" >> gpurun_out/ws_diag.log
  VS_KERNEL=ws VS_RING_SLOTS=$1 VS_GEN_MIN=$2 VS_READY_MIN=$3 VS_GEN_LOW=$4 timeout -k 5 120 python tools/diag_ws.py >> gpurun_out/ws_diag.log 2>&1
done
cat gpurun_out/ws_diag.log
