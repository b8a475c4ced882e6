#!/bin/bash
# carried-T4 fix: the new tests must pass on the fixed library and FAIL on the old kernel
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== old kernel (expected to fail)" > gpurun_out/r2_t4.log
VS_LIB=$PWD/voice_synth_amd/lib/libvoicesynth_stalet4.so timeout -k 10 300 python -m pytest tests/test_gpu_carried_t4.py tests/test_gpu_golden.py -q >> gpurun_out/r2_t4.log 2>&1
echo "rc(old)=$?" >> gpurun_out/r2_t4.log
echo "== fixed kernel" >> gpurun_out/r2_t4.log
timeout -k 10 300 python -m pytest tests/test_gpu_carried_t4.py tests/test_gpu_golden.py -q >> gpurun_out/r2_t4.log 2>&1
rc=$?
echo "rc(new)=$rc" >> gpurun_out/r2_t4.log
grep -E "passed|failed|rc\(|==" gpurun_out/r2_t4.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 900 python tools/fuzz_soak.py 1000 60 12000 5000 > gpurun_out/r2_fuzz_soak.log 2>&1; rc=$?
grep -v ": ok" gpurun_out/r2_fuzz_soak.log | tail -8
exit $rc
