"""Phase breakdown of the WAVE-SPECIALISED fused kernel from the diagnostic build."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import voice_synth_amd._ffi as ffi
ffi.LIB_PATH = os.path.join(os.path.dirname(ffi.LIB_PATH), os.environ.get("VS_DIAG_LIB", "libvoicesynth_diag.so"))
import voice_synth_amd as vs
from voice_synth_amd import configs
GN = ["jitter+shimmer", "rising", "Knew+falling", "closed", "noise", "bookkeeping", "sleep/poll", "loop ctl"]
FN = ["superstep+publish", "-", "-", "-", "-", "-", "sleep/poll", "poll+decide"]
def main():
    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    specs, fs, dur, label = configs.config_specs(cfg, n)
    print(label)
    lanes, d = vs.lanes_from_specs(specs); ns = vs.num_samples(fs, d)
    eng = vs.Engine(0); lib = vs.load()
    if len(sys.argv) > 3 and sys.argv[3] == "fma":
        eng.set_arith(vs.VS_ARITH_FMA); print("VS_ARITH_FMA")
    lib.vs_plan_set_diag.restype = C.c_int; lib.vs_plan_set_diag.argtypes = [C.c_void_p, C.c_void_p]
    plan = eng.plan(lanes, ns); grid = plan.info()["workgroups"]
    out = eng.dev_alloc(n * ns * 2); dg = eng.dev_alloc(grid * 16 * 8)
    lib.vs_plan_set_diag(plan._plan, C.c_void_p(dg))
    for _ in range(2):
        plan.launch(vs.VS_KIND_SYNTH, out); eng.synchronize()
    a = eng.dev_download(dg, (grid, 16), np.uint64).astype(np.float64)
    print("generator rounds per group %.1f (a lane has %.1f cycles on average), lanes per round %.1f" %
          (a[:, 9].mean(), 0.0 if a[:, 9].mean() == 0 else a[:, 10].sum() / 64.0 / a.shape[0], a[:, 10].sum() / max(a[:, 9].sum(), 1.0)))
    if a[:, 11].sum() > 0:
        print("noise wavefront (three-role kernel): working %.1f, asleep %.1f ticks/sample" % (a[:, 11].mean() / ns, a[:, 12].mean() / ns))
    for w, names in ((0, GN), (1, FN)):
        part = a[:, 8 * w:8 * w + 8].copy()
        if w == 1: part[:, 1:6] = 0
        tot = part.sum(axis=1).mean()
        print("wave %d (%s): ticks per sample %.0f" % (w, "generator" if w == 0 else "filter", tot / ns))
        for k in range(8):
            if names[k] != "-":
                print("    %-18s %7.1f ticks/sample" % (names[k], part[:, k].mean() / ns))
    # per group (groups are cut from the lanes sorted by period): where the slowest ones spend their time
    wall = a[:, 8] + a[:, 14] + a[:, 15]
    order = np.argsort(wall)
    print("filter wavefront per group, ticks/sample: wall min %.0f median %.0f max %.0f" % (wall.min() / ns, np.median(wall) / ns, wall.max() / ns))
    print("   group  wall  work  sleep  rounds  lanes/round  noise-work  noise-sleep   (every 64th group in order of period)")
    for g in list(range(0, a.shape[0], 64)) + [int(order[-1])]:
        print("   %5d %5.0f %5.0f %6.0f %7.0f %12.1f %11.0f %12.0f" % (g, wall[g] / ns, a[g, 8] / ns, a[g, 14] / ns, a[g, 9], a[g, 10] / max(a[g, 9], 1), a[g, 11] / ns, a[g, 12] / ns))
main()
