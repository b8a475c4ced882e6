"""Phase breakdown of the fused kernel from the DIAGNOSTIC build (make diag): s_memtime stamps
summed per wavefront.  Shares, not absolute times (the stamps serialise the phases)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import voice_synth_amd._ffi as ffi
ffi.LIB_PATH = os.path.join(os.path.dirname(ffi.LIB_PATH), "libvoicesynth_diag.so")
import voice_synth_amd as vs
from voice_synth_amd import configs

NAMES = ["jitter+shimmer", "rising", "Knew+falling", "closed(no noise)", "noise", "bookkeeping", "filter", "loop ctl"]

def main():
    index = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    specs, fs, dur, label = configs.config_specs(index, n)
    lanes, d = vs.lanes_from_specs(specs)
    ns = vs.num_samples(fs, d)
    eng = vs.Engine(0)
    lib = vs.load()
    lib.vs_plan_set_diag.restype = C.c_int
    lib.vs_plan_set_diag.argtypes = [C.c_void_p, C.c_void_p]
    plan = eng.plan(lanes, ns)
    grid = plan.info()["workgroups"]
    out = eng.dev_alloc(n * ns * 2)
    dg = eng.dev_alloc(grid * 8 * 8)
    lib.vs_plan_set_diag(plan._plan, C.c_void_p(dg))
    for arith, name in ((vs.VS_ARITH_EXACT, "exact"), (vs.VS_ARITH_FMA, "fma")):
        eng.set_arith(arith)
        for kind, kname in ((vs.VS_KIND_SYNTH, "synth"), (vs.VS_KIND_SOURCE, "source")):
            plan.launch(kind, out); eng.synchronize()
            plan.launch(kind, out); eng.synchronize()
            a = eng.dev_download(dg, (grid, 8), np.uint64).astype(np.float64)
            tot = a.sum(axis=1)
            print("%s/%s: cycles per wave %.3e (per sample %.0f); over waves: min %.3e p50 %.3e p99 %.3e max %.3e" %
                  (name, kname, tot.mean(), tot.mean() / ns, tot.min(), np.percentile(tot, 50), np.percentile(tot, 99), tot.max()))
            for k in range(8):
                print("    %-18s %6.1f%%   %7.1f cycles/sample" % (NAMES[k], 100 * a[:, k].mean() / tot.mean(), a[:, k].mean() / ns))
main()
