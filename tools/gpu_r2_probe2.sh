#!/bin/bash
# Round 2, probe 2: fp32 issue costs, other configurations, the wave-specialised kernel on a full grid
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
rc=$?; tail -5 gpurun_out/pytest_gpu.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 tools/ubench/ubench2 > gpurun_out/r2_ubench2b.log 2>&1 || { tail -5 gpurun_out/r2_ubench2b.log; exit 1; }
cat gpurun_out/r2_ubench2b.log
for cfg in "5 65536" "4 32768" "2 65536" "2 1024"; do
  timeout -k 10 300 python tools/quick_bench.py $cfg 3 > gpurun_out/r2_qb_cfg.log 2>&1 || { tail -5 gpurun_out/r2_qb_cfg.log; exit 1; }
  grep -E "config|synth|source|plan" gpurun_out/r2_qb_cfg.log
done
echo "== wave-specialised kernel forced on the full grid (65536)"
VS_DEBUG_TUNING=1 VS_KERNEL=ws timeout -k 10 300 python tools/quick_bench.py 3 65536 3 > gpurun_out/r2_qb_ws.log 2>&1 || { tail -5 gpurun_out/r2_qb_ws.log; exit 1; }
grep -E "plan|synth" gpurun_out/r2_qb_ws.log
for rm in 32 48 64; do for gm in 8 32; do
echo "== ws full grid ready_min=$rm gen_min=$gm"
VS_DEBUG_TUNING=1 VS_KERNEL=ws VS_READY_MIN=$rm VS_GEN_MIN=$gm timeout -k 10 300 python tools/quick_bench.py 3 65536 3 2>&1 | grep -E "synth"
done; done
