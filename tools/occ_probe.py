"""Does a second wave per SIMD help the REAL instruction mix?  Short periods (F0 = 300 Hz) need a
small ring, so 131072 utterances fit as 2048 one-wave workgroups = two waves per SIMD; compare the
per-utterance rate with 65536 utterances = one wave per SIMD (same kernel, same work per lane)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import voice_synth_amd as vs
fa = ["-r", "16000", "-d", "1", "-f", "300", "-g", "313", "-j", "1", "-s", "5.76", "-n", "20"]
eng = vs.Engine(0)
for n in (65536, 131072, 196608, 262144):
    specs = [(fa, ["-v", "12467"[l % 5]], 1 + l) for l in range(n)]
    lanes, d = vs.lanes_from_specs(specs)
    ns = vs.num_samples(16000, d)
    plan = eng.plan(lanes, ns)
    out = eng.dev_alloc(n * ns * 2)
    for arith, name in ((vs.VS_ARITH_EXACT, "exact"), (vs.VS_ARITH_FMA, "fma")):
        eng.set_arith(arith)
        ts = []
        for r in range(4):
            t0 = time.perf_counter(); plan.launch(vs.VS_KIND_SYNTH, out); eng.synchronize(); ts.append(time.perf_counter() - t0)
        t = min(ts[1:])
        print("%6d utterances %s %s: %.3f ms  %.1f Gsamples/s" % (n, plan.info(), name, t * 1e3, n * ns / t / 1e9), flush=True)
    eng.dev_free(out); plan.close()
