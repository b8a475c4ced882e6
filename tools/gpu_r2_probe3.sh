#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
for prio in 0 3; do for rm in 40 48 64; do for gm in 8 32; do
echo "== ws full grid prio=$prio ready_min=$rm gen_min=$gm"
VS_WS_PRIO=$prio VS_DEBUG_TUNING=1 VS_KERNEL=ws VS_READY_MIN=$rm VS_GEN_MIN=$gm timeout -k 10 300 python tools/quick_bench.py 3 65536 3 2>&1 | grep -E "synth"
done; done; done
echo "== single"
timeout -k 10 300 python tools/quick_bench.py 3 65536 3 2>&1 | grep -E "synth"
