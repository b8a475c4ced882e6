#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python tools/diag_bench.py 3 65536 > gpurun_out/diag.log 2>&1
cat gpurun_out/diag.log
