import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import voice_synth_amd as vs
from voice_synth_amd import configs
def strip_noise(specs):
    out = []
    for fa, va, seed in specs:
        fa = list(fa); i = fa.index("-n"); del fa[i:i + 2]
        out.append((fa, va, seed))
    return out
for cfg, n in ((5, 65536), (5, 32768), (4, 32768)):
    specs, fs, dur, label = configs.config_specs(cfg, n)
    specs = strip_noise(specs)
    lanes, d = vs.lanes_from_specs(specs); ns = vs.num_samples(fs, d); pitch = vs.row_pitch(ns)
    for roles in (2, 3):
        eng = vs.Engine(0)
        eng.set_tuning(ws_roles=roles)
        plan = eng.plan(lanes, ns); out = eng.dev_alloc(n * pitch * 2)
        res = {}
        for ar, nm in ((vs.VS_ARITH_EXACT, "exact"), (vs.VS_ARITH_FMA, "fma"), (vs.VS_ARITH_F32, "f32")):
            eng.set_arith(ar)
            ts = []
            for r in range(9):
                t0 = time.perf_counter(); plan.launch(vs.VS_KIND_SYNTH, out, out_pitch=pitch); eng.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
            res[nm] = sorted(ts[2:])[len(ts[2:]) // 2]
        print("config %d WITHOUT glottal noise, lanes %d roles %s (%s %s): exact %.3f  fma %.3f  f32 %.3f ms" % (cfg, n, roles, plan.roles(), plan.info(), res["exact"], res["fma"], res["f32"]), flush=True)
        eng.dev_free(out); plan.close(); eng.close()
