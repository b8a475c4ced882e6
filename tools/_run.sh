set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
tools/insts.sh 3 65536 16000 pairok | grep -E "^==|ws_kernel|kernel<0, 1|synth:"
VS_LIB=libvoicesynth_pairok.so timeout -k 10 300 python -m pytest tests/test_gpu_golden.py -m gpu -x -q 2>&1 | tail -2
