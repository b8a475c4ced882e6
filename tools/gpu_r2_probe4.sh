#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== single"
timeout -k 10 300 python tools/quick_bench.py 3 65536 3 2>&1 | grep -E "synth"
for pairs in 1 4; do for prio in 0 3; do for rm in 48 64; do for gm in 16 32 64; do
echo "== ws full grid pairs=$pairs prio=$prio ready_min=$rm gen_min=$gm"
VS_WS_PRIO=$prio VS_DEBUG_TUNING=1 VS_KERNEL=ws VS_WS_PAIRS=$pairs VS_READY_MIN=$rm VS_GEN_MIN=$gm timeout -k 10 300 python tools/quick_bench.py 3 65536 3 2>&1 | grep -E "synth"
done; done; done; done
echo "== single"
timeout -k 10 300 python tools/quick_bench.py 3 65536 3 2>&1 | grep -E "synth"
