"""Phase breakdown (diagnostic build) of the one-wave fused kernel at 1 and 2 waves per SIMD:
the F0 = 300 Hz workload of occ_probe.py (12 KiB ring) at 65536 and 131072 utterances.  Shows
WHICH phases get cheaper per wave-sample when a second wave shares the SIMD."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import voice_synth_amd._ffi as ffi
ffi.LIB_PATH = os.path.join(os.path.dirname(ffi.LIB_PATH), "libvoicesynth_diag.so")
import voice_synth_amd as vs

NAMES = ["jitter+shimmer", "rising", "Knew+falling", "closed(no noise)", "noise", "bookkeeping", "filter", "loop ctl"]
fa = ["-r", "16000", "-d", "1", "-f", "300", "-g", "313", "-j", "1", "-s", "5.76", "-n", "20"]
eng = vs.Engine(0)
lib = vs.load()
lib.vs_plan_set_diag.restype = C.c_int
lib.vs_plan_set_diag.argtypes = [C.c_void_p, C.c_void_p]
for n in (65536, 131072, 262144):
    specs = [(fa, ["-v", "12467"[l % 5]], 1 + l) for l in range(n)]
    lanes, d = vs.lanes_from_specs(specs)
    ns = vs.num_samples(16000, d)
    plan = eng.plan(lanes, ns)
    grid = plan.info()["workgroups"]
    out = eng.dev_alloc(n * ns * 2)
    dg = eng.dev_alloc(grid * 8 * 8)
    lib.vs_plan_set_diag(plan._plan, C.c_void_p(dg))
    for arith, name in ((vs.VS_ARITH_EXACT, "exact"), (vs.VS_ARITH_FMA, "fma")):
        eng.set_arith(arith)
        plan.launch(vs.VS_KIND_SYNTH, out); eng.synchronize()
        plan.launch(vs.VS_KIND_SYNTH, out); eng.synchronize()
        a = eng.dev_download(dg, (grid, 8), np.uint64).astype(np.float64)
        tot = a.sum(axis=1)
        print("%d utterances %s %s: ticks per wave mean %.3e min %.3e max %.3e (per sample %.0f)" %
              (n, plan.info(), name, tot.mean(), tot.min(), tot.max(), tot.mean() / ns))
        for k in range(8):
            print("    %-18s %6.1f%%   %7.1f ticks/sample" % (NAMES[k], 100 * a[:, k].mean() / tot.mean(), a[:, k].mean() / ns))
    eng.dev_free(dg); eng.dev_free(out); plan.close()
