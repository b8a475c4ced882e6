"""Quick look at kernel time (wall clock around launch+sync); bench.py is the real harness.
   tools/quick_bench.py [config] [lanes] [reps]      QB_ONOISE=<dB>: every utterance with "vowel -n <dB>" as well"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import voice_synth_amd as vs
from voice_synth_amd import configs

def main():
    index = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    onoise = os.environ.get("QB_ONOISE")
    specs, fs, dur, label = configs.config_specs(index, n, out_noise_db=float(onoise) if onoise else None)
    t0 = time.time()
    lanes, d = vs.lanes_from_specs(specs)
    ns = vs.num_samples(fs, d)
    print(label, "lanes", n, "samples", ns, "host spec build %.2fs" % (time.time() - t0), flush=True)
    eng = vs.Engine(0)
    print(eng.device_info(), flush=True)
    t0 = time.time()
    plan = eng.plan(lanes, ns)
    print("plan create %.3fs" % (time.time() - t0), plan.info(), plan.kernel_name(vs.VS_KIND_SYNTH), flush=True)
    out = eng.dev_alloc(n * ns * 2)
    for arith, name in ((vs.VS_ARITH_EXACT, "exact"), (vs.VS_ARITH_FMA, "fma")):
        eng.set_arith(arith)
        for kind, kname in ((vs.VS_KIND_SYNTH, "synth"), (vs.VS_KIND_SOURCE, "source")):
            if arith == vs.VS_ARITH_FMA and kind == vs.VS_KIND_SOURCE:
                continue
            ts = []
            for r in range(reps + 1):
                t0 = time.perf_counter()
                plan.launch(kind, out)
                eng.synchronize()
                ts.append(time.perf_counter() - t0)
            t = min(ts[1:])
            print("%s/%s: %.3f ms  %.1f Msamples/s  %.3f TB/s (2 B/sample)  first %.1f ms" %
                  (name, kname, t * 1e3, n * ns / t / 1e6, 2 * n * ns / t / 1e12, ts[0] * 1e3), flush=True)
    # filter-only: flow from a source launch as input
    flow = eng.dev_alloc(n * ns * 2)
    eng.set_arith(vs.VS_ARITH_EXACT)
    plan.launch(vs.VS_KIND_SOURCE, flow)
    eng.synchronize()
    for arith, name in ((vs.VS_ARITH_EXACT, "exact"), (vs.VS_ARITH_FMA, "fma")):
        eng.set_arith(arith)
        ts = []
        for r in range(reps + 1):
            t0 = time.perf_counter()
            plan.launch(vs.VS_KIND_FILTER, out, in_ptr=flow)
            eng.synchronize()
            ts.append(time.perf_counter() - t0)
        t = min(ts[1:])
        print("%s/filter: %.3f ms  %.1f Msamples/s  %.3f TB/s (2 B/sample)" %
              (name, t * 1e3, n * ns / t / 1e6, 2 * n * ns / t / 1e12), flush=True)
    eng.dev_free(flow)
    eng.dev_free(out)

main()
