#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 400 tools/ubench/ubench2 > gpurun_out/r2_ubench2c.log 2>&1 || { tail -5 gpurun_out/r2_ubench2c.log; exit 1; }
tail -12 gpurun_out/r2_ubench2c.log
cd /tmp
ROOT=$GRAFT_REPO_ROOT
timeout -k 10 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $ROOT/gpurun_out/prof_clk -o bench -- python3 $ROOT/bench.py --no-cpu-baseline --steps 5 --warmup 2 > $ROOT/gpurun_out/prof_clk.log 2>&1
rc=$?; cd $ROOT; [ $rc -ne 0 ] && { tail -5 gpurun_out/prof_clk.log; exit 1; }
python - <<'PY'
import csv, collections
v=collections.defaultdict(lambda: collections.defaultdict(list)); d={}
for r in csv.DictReader(open("gpurun_out/prof_clk/bench_counter_collection.csv")):
    k=r["Kernel_Name"]
    if "vs_synth" not in k: continue
    v[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    d.setdefault(k,{})[r["Dispatch_Id"]]=float(r["End_Timestamp"])-float(r["Start_Timestamp"])
for k in v:
    ns=sorted(d[k].values()); ns=ns[len(ns)//2]
    g=sum(v[k]["GRBM_GUI_ACTIVE"])/len(v[k]["GRBM_GUI_ACTIVE"])
    print(k, "median %.3f ms under the profiler, GRBM_GUI_ACTIVE/8/time = %.3f GHz" % (ns/1e6, g/8/ns))
PY
