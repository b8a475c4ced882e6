#!/bin/bash
mkdir -p gpurun_out
: > gpurun_out/iter.log
for lib in libvoicesynth.so libvoicesynth_s4.so libvoicesynth_s16.so libvoicesynth_s32.so; do
echo "== ws full grid $lib" >> gpurun_out/iter.log
VS_LIB=$lib VS_KERNEL=ws VS_RING_SLOTS=312 VS_GEN_MIN=48 VS_READY_MIN=64 timeout -k 10 120 python tools/quick_bench.py 3 65536 3 2>&1 | grep -E "exact/synth|fma/synth" >> gpurun_out/iter.log
done
cat gpurun_out/iter.log
