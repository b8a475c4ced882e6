"""What sorting a homogeneous batch by each utterance's CYCLE COUNT would buy (attendance): the counts come
from a source-only launch (vs_plan_launch ... ncyc), the lanes are permuted on the host, the fused launch is timed on
both orders.  Same plan shape, same kernel; the output rows follow the lanes.
    python tools/sort_probe.py [config] [lanes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import voice_synth_amd as vs
from voice_synth_amd import configs


def timed(eng, plan, out, reps=7):
    ts = []
    for _ in range(reps + 2):
        t0 = time.perf_counter()
        plan.launch(vs.VS_KIND_SYNTH, out)
        eng.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts[2:]), float(np.median(ts[2:]))


def main():
    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    specs, fs, dur, label = configs.config_specs(cfg, n)
    lanes, d = vs.lanes_from_specs(specs)
    ns = vs.num_samples(fs, d)
    eng = vs.Engine(0)
    out = eng.dev_alloc(n * ns * 2)
    ncyc_dev = eng.dev_alloc(n * 4)
    plan = eng.plan(lanes, ns)
    plan.launch(vs.VS_KIND_SOURCE, out, ncyc_ptr=ncyc_dev)
    eng.synchronize()
    ncyc = eng.dev_download(ncyc_dev, (n,), np.int32)
    print(label, "cycles per utterance: min %d mean %.1f max %d" % (ncyc.min(), ncyc.mean(), ncyc.max()))
    g = ncyc.reshape(-1, 64)
    print("as given: rounds per group (max of 64) mean %.1f -> attendance %.3f" % (g.max(axis=1).mean(), ncyc.mean() / g.max(axis=1).mean()))
    order = np.argsort(ncyc, kind="stable")
    gs = ncyc[order].reshape(-1, 64)
    print("sorted  : rounds per group mean %.1f -> attendance %.3f" % (gs.max(axis=1).mean(), ncyc.mean() / gs.max(axis=1).mean()))
    sorted_lanes = (vs.Lane * n)(*[lanes[int(i)] for i in order])
    plan_s = eng.plan(sorted_lanes, ns)
    for arith, name in ((vs.VS_ARITH_EXACT, "exact"), (vs.VS_ARITH_FMA, "fma")):
        eng.set_arith(arith)
        for rep in range(3):
            a = timed(eng, plan, out)
            b = timed(eng, plan_s, out)
            print("%s rep %d: as given %.3f / %.3f ms (min / median), sorted by cycle count %.3f / %.3f ms  -> %.1f %%"
                  % (name, rep, a[0], a[1], b[0], b[1], 100.0 * (b[1] / a[1] - 1.0)), flush=True)
    print(plan.kernel_name(), plan_s.kernel_name())


main()
