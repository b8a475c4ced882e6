#!/bin/bash
# whole config 4 on one device + node tests + node entry timing
set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_properties.py::test_config4_whole_batch_on_one_device tests/test_gpu_node.py -x -q -s > gpurun_out/r2_whole4.log 2>&1
timeout -k 10 300 python tools/node_bench.py > gpurun_out/r2_node_bench2.log 2>&1
tail -5 gpurun_out/r2_whole4.log; cat gpurun_out/r2_node_bench2.log
