set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 200 python tools/diag_ws.py 3 > gpurun_out/diag3.txt 2>&1 || { tail -5 gpurun_out/diag3.txt; exit 1; }
timeout -k 10 200 python tools/diag_ws.py 5 > gpurun_out/diag5.txt 2>&1 || { tail -5 gpurun_out/diag5.txt; exit 1; }
cat gpurun_out/diag3.txt; cat gpurun_out/diag5.txt
