#!/bin/bash
# rocprofv3 kernel trace + a few PMC passes of the quick bench (config 3, full batch)
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$(pwd)
timeout -k 10 300 python tools/quick_bench.py 3 65536 3 > gpurun_out/quick_bench.log 2>&1 || { cat gpurun_out/quick_bench.log; exit 1; }
cat gpurun_out/quick_bench.log
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_trace -o qb -- python3 $ROOT/tools/quick_bench.py 3 65536 3 > $ROOT/gpurun_out/prof_trace.log 2>&1 || { tail -20 $ROOT/gpurun_out/prof_trace.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $ROOT/gpurun_out/prof_pmc1 -o qb -- python3 $ROOT/tools/quick_bench.py 3 65536 1 > $ROOT/gpurun_out/prof_pmc1.log 2>&1 || { tail -20 $ROOT/gpurun_out/prof_pmc1.log; exit 1; }
cd $ROOT
find gpurun_out/prof_trace -name "*stats*" | head; 
for f in $(find gpurun_out/prof_trace -name "*kernel_stats.csv"); do cat $f; done
