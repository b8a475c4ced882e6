#!/bin/bash
# HBM traffic (two separate PMC passes, as MI355X_MICROARCH.md prescribes) of the other BASELINE
# configurations -> profiles/pmc_traffic.json keys config<N>_exact_<lanes>
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$(pwd)
for SPEC in "2 1024" "4 32768" "5 65536"; do
  set -- $SPEC; CFG=$1; LANES=$2
  for C in FETCH_SIZE WRITE_SIZE; do
    cd /tmp
    timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $ROOT/gpurun_out/prof_${C}_cfg$CFG -o bench -- python3 $ROOT/bench.py --config $CFG --no-cpu-baseline --steps 5 --warmup 2 > $ROOT/gpurun_out/prof_${C}_cfg$CFG.log 2>&1
    rc=$?; cd $ROOT; [ $rc -ne 0 ] && { tail -3 gpurun_out/prof_${C}_cfg$CFG.log; exit $rc; }
  done
  python tools/summarize_pmc.py traffic gpurun_out/prof_FETCH_SIZE_cfg$CFG/bench_counter_collection.csv gpurun_out/prof_WRITE_SIZE_cfg$CFG/bench_counter_collection.csv config${CFG}_exact_$LANES "vs_synth_ws_kernel<0"
done
cp profiles/pmc_traffic.json gpurun_out/pmc_traffic_all.json
