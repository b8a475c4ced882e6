"""What sorting a homogeneous batch by each utterance's CYCLE COUNT buys: the
generator's rounds cost as much as their longest lane and last as long as the lane with the most
cycles, so lanes that need the same number of cycles make full rounds.  The counts come from the
engine's own cycle log here; a plan-time pre-pass would have to produce them.

    python tools/sort_probe.py [config] [lanes]
"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import voice_synth_amd as vs  # noqa: E402
from voice_synth_amd import configs  # noqa: E402


def timed(eng, plan, out, reps=5):
    plan.launch(vs.VS_KIND_SYNTH, out)
    eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        plan.launch(vs.VS_KIND_SYNTH, out)
    eng.synchronize()
    plan.status()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    n_lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    specs, fs, dur, label = configs.config_specs(cfg, n_lanes)
    lanes, d = vs.lanes_from_specs(specs)
    n = vs.num_samples(fs, d)
    eng = vs.Engine(0)
    ncyc = np.zeros(n_lanes, dtype=np.int32)
    for lo in range(0, n_lanes, 8192):
        sub = (vs.Lane * 8192)()
        C.memmove(sub, C.byref(lanes, lo * C.sizeof(vs.Lane)), 8192 * C.sizeof(vs.Lane))
        ncyc[lo:lo + 8192] = eng.source(sub, n, log_cycles=1)[2]
    print("%s: cycles per utterance min %d mean %.1f max %d" % (label, ncyc.min(), ncyc.mean(), ncyc.max()))
    order = np.argsort(ncyc, kind="stable")
    sorted_lanes = (vs.Lane * n_lanes)()
    for k, src in enumerate(order):
        C.memmove(C.byref(sorted_lanes, k * C.sizeof(vs.Lane)), C.byref(lanes, int(src) * C.sizeof(vs.Lane)), C.sizeof(vs.Lane))
    out = eng.dev_alloc(n_lanes * n * 2)
    for arith, name in ((vs.VS_ARITH_EXACT, "exact"), (vs.VS_ARITH_FMA, "fma")):
        eng.set_arith(arith)
        pa = eng.plan(lanes, n)
        pb = eng.plan(sorted_lanes, n)
        ta, tb = [], []
        for _ in range(3):
            ta.append(timed(eng, pa, out))
            tb.append(timed(eng, pb, out))
        print("%s: as given %.3f ms, sorted by cycle count %.3f ms  (%.1f %%)" % (
            name, min(ta), min(tb), 100.0 * (min(ta) - min(tb)) / min(ta)))
        pa.close()
        pb.close()
    eng.dev_free(out)
    eng.close()


main()
