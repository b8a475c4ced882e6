#!/bin/bash
# Round 2, probe 1: issue-cost micro-benchmarks at 1..4 waves/SIMD, filter-only / source-only scaling
# with occupancy, per-wave spread of the fused kernel (diagnostic build).
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 300 tools/ubench/ubench2 > gpurun_out/r2_ubench2.log 2>&1 || { tail -5 gpurun_out/r2_ubench2.log; exit 1; }
cat gpurun_out/r2_ubench2.log
timeout -k 10 300 python tools/quick_bench.py 3 65536 > gpurun_out/r2_qb_65536.log 2>&1 || { tail -5 gpurun_out/r2_qb_65536.log; exit 1; }
timeout -k 10 300 python tools/quick_bench.py 3 131072 3 > gpurun_out/r2_qb_131072.log 2>&1 || { tail -5 gpurun_out/r2_qb_131072.log; exit 1; }
timeout -k 10 300 python tools/quick_bench.py 3 196608 3 > gpurun_out/r2_qb_196608.log 2>&1 || { tail -5 gpurun_out/r2_qb_196608.log; exit 1; }
cat gpurun_out/r2_qb_*.log
timeout -k 10 300 python tools/diag_bench.py 3 65536 > gpurun_out/r2_diag.log 2>&1 || { tail -5 gpurun_out/r2_diag.log; exit 1; }
cat gpurun_out/r2_diag.log
timeout -k 10 300 python tools/diag_occ.py > gpurun_out/r2_diag_occ.log 2>&1 || { tail -5 gpurun_out/r2_diag_occ.log; exit 1; }
cat gpurun_out/r2_diag_occ.log
