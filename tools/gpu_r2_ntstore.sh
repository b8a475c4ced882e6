#!/bin/bash
# write traffic (WRITE_SIZE) and time of the output stores as they are vs non-temporal
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$(pwd)
for V in base nt; do
  cd /tmp
  VS_LIB=libvoicesynth_$V.so timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $ROOT/gpurun_out/prof_w_$V -o bench -- python3 $ROOT/bench.py --no-cpu-baseline --steps 5 --warmup 2 > $ROOT/gpurun_out/prof_w_$V.log 2>&1
  cd $ROOT
  python - <<PY
import csv
v={}
for r in csv.DictReader(open("gpurun_out/prof_w_$V/bench_counter_collection.csv")):
    if r["Counter_Name"]=="WRITE_SIZE": v.setdefault(r["Kernel_Name"],[]).append(float(r["Counter_Value"]))
for k,x in v.items():
    if "ws_kernel<0" in k: print("$V", k, "WRITE_SIZE GB per launch %.3f over %d launches"%(sum(x)/len(x)*1024/1e9, len(x)))
PY
done
for rep in 1 2 3; do for V in base nt; do echo "== $V"; VS_LIB=libvoicesynth_$V.so timeout -k 10 120 python tools/quick_bench.py 3 65536 5 | grep -E "exact/synth|fma/synth"; done; done
