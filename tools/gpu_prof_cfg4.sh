#!/bin/bash
# rocprofv3 kernel trace of the per-GPU shard of BASELINE config 4 (32768 x 44100: wave-specialised kernel)
mkdir -p gpurun_out; export TMPDIR=/tmp; ROOT=$(pwd); cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_cfg4 -o bench -- python3 $ROOT/bench.py --config 4 --no-cpu-baseline --steps 10 --warmup 3 > $ROOT/gpurun_out/prof_cfg4.log 2>&1
rc=$?; cd $ROOT; tail -2 gpurun_out/prof_cfg4.log; cat gpurun_out/prof_cfg4/bench_kernel_stats.csv; exit $rc
