#!/bin/bash
# Full GPU pass: all gpu tests, smoke, bench (N=1), rocprofv3 kernel trace + PMC passes (traffic, SQ).
# Outputs under gpurun_out/; copy what is to be judged into profiles/ afterwards.
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$(pwd)
TAG=${1:-r02}
if [ "$2" != "--notest" ]; then
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
rc=$?; tail -5 gpurun_out/pytest_gpu.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
rc=$?; tail -3 gpurun_out/smoke.log; [ $rc -ne 0 ] && exit $rc
fi
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_bench -o bench -- python3 $ROOT/bench.py --no-cpu-baseline > $ROOT/gpurun_out/prof_bench.log 2>&1
rc=$?; cd $ROOT; [ $rc -ne 0 ] && { tail -5 gpurun_out/prof_bench.log; exit $rc; }
cat gpurun_out/prof_bench/bench_kernel_stats.csv
cp gpurun_out/prof_bench/bench_kernel_stats.csv gpurun_out/${TAG}_bench_kernel_stats.csv
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
timeout -k 10 600 rocprofv3 --pmc $C --output-format csv -d $ROOT/gpurun_out/prof_$C -o bench -- python3 $ROOT/bench.py --no-cpu-baseline --steps 5 --warmup 2 > $ROOT/gpurun_out/prof_$C.log 2>&1
rc=$?; [ $rc -ne 0 ] && { tail -5 $ROOT/gpurun_out/prof_$C.log; exit $rc; }
done
timeout -k 10 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $ROOT/gpurun_out/prof_sq1 -o bench -- python3 $ROOT/bench.py --no-cpu-baseline --steps 5 --warmup 2 > $ROOT/gpurun_out/prof_sq1.log 2>&1
rc=$?; [ $rc -ne 0 ] && { tail -5 $ROOT/gpurun_out/prof_sq1.log; exit $rc; }
timeout -k 10 600 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $ROOT/gpurun_out/prof_sq2 -o bench -- python3 $ROOT/bench.py --no-cpu-baseline --steps 5 --warmup 2 > $ROOT/gpurun_out/prof_sq2.log 2>&1
rc=$?; cd $ROOT; [ $rc -ne 0 ] && { tail -5 gpurun_out/prof_sq2.log; }
python tools/summarize_pmc.py traffic gpurun_out/prof_FETCH_SIZE/bench_counter_collection.csv gpurun_out/prof_WRITE_SIZE/bench_counter_collection.csv config3_exact_65536 "vs_synth_ws_kernel<0"
python tools/summarize_pmc.py traffic gpurun_out/prof_FETCH_SIZE/bench_counter_collection.csv gpurun_out/prof_WRITE_SIZE/bench_counter_collection.csv config3_fma_65536 "vs_synth_ws_kernel<1"
SQ="gpurun_out/prof_sq1/bench_counter_collection.csv"; [ -f gpurun_out/prof_sq2/bench_counter_collection.csv ] && SQ="$SQ gpurun_out/prof_sq2/bench_counter_collection.csv"
python tools/summarize_pmc.py valu $SQ config3_exact_65536 --kernel "vs_synth_ws_kernel<0"
python tools/summarize_pmc.py valu $SQ config3_fma_65536 --kernel "vs_synth_ws_kernel<1"
cp profiles/pmc_traffic.json profiles/pmc_valu.json gpurun_out/
timeout -k 10 900 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err
rc=$?; cat gpurun_out/bench.json; tail -3 gpurun_out/bench.err
cp gpurun_out/bench.json gpurun_out/${TAG}_bench_n1.json
exit $rc
