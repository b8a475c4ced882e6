#!/bin/bash
# Full GPU pass: all gpu tests, smoke, bench (N=1), rocprofv3 kernel trace of the bench command.
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$(pwd)
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
rc=$?; tail -15 gpurun_out/pytest_gpu.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
rc=$?; tail -3 gpurun_out/smoke.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err
rc=$?; cat gpurun_out/bench.json; tail -3 gpurun_out/bench.err; [ $rc -ne 0 ] && exit $rc
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_bench -o bench -- python3 $ROOT/bench.py --no-cpu-baseline > $ROOT/gpurun_out/prof_bench.log 2>&1
rc=$?; cd $ROOT; tail -2 gpurun_out/prof_bench.log; cat gpurun_out/prof_bench/bench_kernel_stats.csv
exit $rc
