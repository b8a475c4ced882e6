#!/bin/bash
# Full GPU pass: all gpu tests, smoke, bench (N=1), rocprofv3 kernel trace + PMC traffic passes.
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$(pwd)
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
rc=$?; tail -5 gpurun_out/pytest_gpu.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
rc=$?; tail -3 gpurun_out/smoke.log; [ $rc -ne 0 ] && exit $rc
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_bench -o bench -- python3 $ROOT/bench.py --no-cpu-baseline > $ROOT/gpurun_out/prof_bench.log 2>&1
rc=$?; cd $ROOT; [ $rc -ne 0 ] && { tail -5 gpurun_out/prof_bench.log; exit $rc; }
cat gpurun_out/prof_bench/bench_kernel_stats.csv
cd /tmp
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $ROOT/gpurun_out/prof_fetch -o bench -- python3 $ROOT/bench.py --no-cpu-baseline --steps 5 --warmup 2 > $ROOT/gpurun_out/prof_fetch.log 2>&1
rc=$?; [ $rc -ne 0 ] && { tail -5 $ROOT/gpurun_out/prof_fetch.log; exit $rc; }
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $ROOT/gpurun_out/prof_write -o bench -- python3 $ROOT/bench.py --no-cpu-baseline --steps 5 --warmup 2 > $ROOT/gpurun_out/prof_write.log 2>&1
rc=$?; cd $ROOT; [ $rc -ne 0 ] && { tail -5 gpurun_out/prof_write.log; exit $rc; }
python tools/summarize_pmc.py gpurun_out/prof_fetch/bench_counter_collection.csv gpurun_out/prof_write/bench_counter_collection.csv config3_exact_65536
cp profiles/pmc_traffic.json gpurun_out/pmc_traffic.json
timeout -k 10 600 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err
rc=$?; cat gpurun_out/bench.json; tail -3 gpurun_out/bench.err
exit $rc
