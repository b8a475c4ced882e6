"""Mixed rings with three roles: which SIMD a group's open-phase / noise wavefronts run on -- the experiment recorded in
profiles/r06_crossed_roles.txt (every crossing is slower; NOT shipped).  Needs the patch at the end of that file (it adds
vs_tuning.ws_cross); against the shipped library the field does not exist and every code measures the shipped layout.
   tools/cross_sweep.py [config] [lanes] [reps] [codes...]
Every code is a plan of its own on the same utterances; the PCM of every code is compared with code -1's (the three
wavefronts of a group on one SIMD) over the whole batch, and the device time of `reps` launches (context timer) printed
for the three arithmetics."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import voice_synth_amd as vs
from voice_synth_amd import configs


def timed(eng, plan, out, reps):
    plan.launch(vs.VS_KIND_SYNTH, out)
    eng.synchronize()
    best, tot = None, 0.0
    for _ in range(reps):
        eng.timer_mark(0)
        plan.launch(vs.VS_KIND_SYNTH, out)
        eng.timer_mark(1)
        eng.synchronize()
        ms = eng.timer_elapsed()
        tot += ms
        best = ms if best is None or ms < best else best
    return best, tot / reps


def main():
    index = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    codes = [int(c) for c in sys.argv[4:]] or [-1, 12, 3, 4, 8, 1, 2, 15, 7, 13, 9, 6]
    specs, fs, dur, label = configs.config_specs(index, n)
    lanes, d = vs.lanes_from_specs(specs)
    ns = vs.num_samples(fs, d)
    eng = vs.Engine(0)
    print(label, eng.device_info(), flush=True)
    out = eng.dev_alloc(n * ns * 2)
    ref = None
    for code in codes:
        eng.set_tuning(ws_cross=code)
        plan = eng.plan(lanes, ns)
        line = "cross %3d (open ^%d, noise ^%d) %s:" % (code, max(code, 0) & 3, (max(code, 0) >> 2) & 3, plan.roles())
        for arith, name in ((vs.VS_ARITH_EXACT, "exact"), (vs.VS_ARITH_FMA, "fma"), (vs.VS_ARITH_F32, "f32")):
            eng.set_arith(arith)
            best, mean = timed(eng, plan, out, reps)
            line += "  %s %.3f (mean %.3f) ms" % (name, best, mean)
            if arith == vs.VS_ARITH_EXACT:
                got = eng.dev_download(out, (n, ns))
                if ref is None:
                    ref = got
                else:
                    line += " [%d samples differ]" % int((got != ref).sum())
                del got
        print(line, flush=True)
        plan.close()
    eng.dev_free(out)
    eng.close()


main()
