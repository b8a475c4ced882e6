"""Time of the wide path (coefficient sets of 23..40 taps: source kernel -> flow in HBM -> wide filter
kernel) next to the fused path, BASELINE config 3 shape (65536 utterances x 16000 samples).

    python tools/wide_bench.py
"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import voice_synth_amd as vs  # noqa: E402
from voice_synth_amd import configs  # noqa: E402


def timed(eng, plan, kind, out, in_ptr=None, reps=5):
    plan.launch(kind, out, in_ptr=in_ptr)
    eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        plan.launch(kind, out, in_ptr=in_ptr)
    eng.synchronize()
    plan.status()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    n_lanes, n = 65536, 16000
    eng = vs.Engine(0)
    out = eng.dev_alloc(n_lanes * n * 2)
    flow = eng.dev_alloc(n_lanes * n * 2)
    specs, fs, dur, label = configs.config_specs(3, n_lanes)
    lanes, _ = vs.lanes_from_specs(specs)
    plan = eng.plan(lanes, n)
    print("tables (order 22), %s: %.2f ms fused (%s)" % (label, timed(eng, plan, vs.VS_KIND_SYNTH, out), plan.kernel_name(vs.VS_KIND_SYNTH)))
    print("   source only %.2f ms, filter only %.2f ms" % (timed(eng, plan, vs.VS_KIND_SOURCE, flow),
                                                        timed(eng, plan, vs.VS_KIND_FILTER, out, in_ptr=flow)))
    plan.close()
    for order in (24, 40):
        proto, _, _ = configs.wide_order_lanes([order] * 256)
        arr = (vs.Lane * n_lanes)()
        for k in range(0, n_lanes, 256):
            C.memmove(C.byref(arr, k * C.sizeof(vs.Lane)), proto, 256 * C.sizeof(vs.Lane))
        for k in range(n_lanes):
            arr[k].seed = 1 + k
        plan = eng.plan(arr, n)
        for arith, name in ((vs.VS_ARITH_EXACT, "exact"), (vs.VS_ARITH_FMA, "fma")):
            eng.set_arith(arith)
            print("sets of %d taps, %s: synth %.2f ms (%s), filter only %.2f ms" % (
                order, name, timed(eng, plan, vs.VS_KIND_SYNTH, out), plan.kernel_name(vs.VS_KIND_SYNTH),
                timed(eng, plan, vs.VS_KIND_FILTER, out, in_ptr=flow)))
        eng.set_arith(vs.VS_ARITH_EXACT)
        plan.close()
    eng.dev_free(out)
    eng.dev_free(flow)
    eng.close()


main()
