set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "rc=$?"; tail -6 gpurun_out/pytest_gpu.log
