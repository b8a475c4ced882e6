set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
for a in exact fma f32; do
  timeout -k 10 120 voice_synth_amd/bin/vs_bench --arith $a --steps 50 --warmup 10 >> gpurun_out/vs_bench_fresh.log || exit 1
  timeout -k 10 120 voice_synth_amd/bin/vs_bench --arith $a --fresh >> gpurun_out/vs_bench_fresh.log || exit 1
done
python - <<'PY'
import json
for l in open("gpurun_out/vs_bench_fresh.log"):
    d=json.loads(l)
    print(d["arith"], d.get("ms_per_batch", d.get("ms_per_step")), d.get("plan_create_wall_ms_avg"), d.get("last_batch_equals_a_plain_launch"))
PY
timeout -k 10 600 python -m pytest tests/test_gpu_cli.py tests/test_gpu_properties.py -m gpu -x -q -k "vs_bench or blocks_of" > gpurun_out/pytest_d.log 2>&1; echo "rc=$?"; tail -4 gpurun_out/pytest_d.log
