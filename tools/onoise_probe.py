"""vowel -n behind the fused kernel: time per launch of the whole sequence and of each kernel (HIP events would need the
library's stream; here: wall clock around launch + synchronize, and rocprofv3 --kernel-trace --stats around this script
for the per-kernel figures).   tools/onoise_probe.py [config] [lanes] [reps]     VS_LIB selects a variant library"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import voice_synth_amd as vs
from voice_synth_amd import configs

index = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
eng = vs.Engine(0)
res = {}
for db in (None, 20.0):
    specs, fs, dur, label = configs.config_specs(index, n, out_noise_db=db)
    lanes, d = vs.lanes_from_specs(specs)
    ns = vs.num_samples(fs, d)
    pitch = vs.row_pitch(ns)
    plan = eng.plan(lanes, ns)
    out = eng.dev_alloc(n * pitch * 2)
    for arith, nm in ((vs.VS_ARITH_EXACT, "exact"), (vs.VS_ARITH_FMA, "fma")):
        eng.set_arith(arith)
        ts = []
        for r in range(reps + 2):
            t0 = time.perf_counter()
            plan.launch(vs.VS_KIND_SYNTH, out, out_pitch=pitch)
            eng.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        ts = sorted(ts[2:])
        res[(db, nm)] = ts[len(ts) // 2]
        print("%s | %s | %s: median %.3f ms, min %.3f" % (os.environ.get("VS_LIB", "default"), label, nm, ts[len(ts) // 2], ts[0]), flush=True)
    eng.set_arith(vs.VS_ARITH_EXACT)
    eng.dev_free(out)
    plan.close()
for nm in ("exact", "fma"):
    print("%s: vowel -n adds %.3f ms to %.3f" % (nm, res[(20.0, nm)] - res[(None, nm)], res[(None, nm)]))
