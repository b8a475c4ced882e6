set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_full_batches.py tests/test_gpu_golden.py tests/test_gpu_custom.py tests/test_gpu_wide.py tests/test_gpu_properties.py tests/test_gpu_cli.py -m gpu -x -q -k "noise or onoise or golden or wide or mixed or fuzz or random or tables" > gpurun_out/pytest_b.log 2>&1; echo "rc=$?"; tail -5 gpurun_out/pytest_b.log
( cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_pow -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/onoise_probe.py 3 65536 10 > $GRAFT_REPO_ROOT/gpurun_out/onoise_pow.log 2>&1 ) || exit 1
grep -E "median|adds" gpurun_out/onoise_pow.log
cut -d'"' -f2,3 gpurun_out/prof_pow/*kernel_stats.csv | head -12
