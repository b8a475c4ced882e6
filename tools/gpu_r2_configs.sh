#!/bin/bash
# bench.py + rocprofv3 kernel stats for the BASELINE configurations that are not the headline
# (2: batch 1024, 4: per-GPU shard 32768 x 44100, 5: F0 sweep) -> gpurun_out/r02_d_config<N>_*
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$(pwd)
for CFG in 2 4 5; do
  cd /tmp
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_cfg$CFG -o bench -- python3 $ROOT/bench.py --config $CFG --no-cpu-baseline > $ROOT/gpurun_out/r02_d_config${CFG}_bench.json 2> $ROOT/gpurun_out/prof_cfg$CFG.err
  rc=$?; cd $ROOT
  [ $rc -ne 0 ] && { tail -5 gpurun_out/prof_cfg$CFG.err; exit $rc; }
  cp gpurun_out/prof_cfg$CFG/bench_kernel_stats.csv gpurun_out/r02_d_config${CFG}_kernel_stats.csv
  echo "== config $CFG"; head -3 gpurun_out/r02_d_config${CFG}_kernel_stats.csv; python - <<PY
import json
d=json.loads(open("gpurun_out/r02_d_config${CFG}_bench.json").read().strip().splitlines()[-1])
print(d["config"]["workload"], d["value"], d["unit"], "ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"], d["roofline"]["kernel"], "fma", d["other_arith"]["kernel_ms_avg"])
PY
done
