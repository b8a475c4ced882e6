"""Where the first vs_plan_create of a process spends its time (90-140 ms of `upload_ms` on a fresh
process against 4 ms afterwards).  Prints (host_ms, upload_ms) of three plans in a row in one process, with and without
torch having touched the device first; run under `rocprofv3 --hip-trace --stats` for the per-call picture.
usage: python tools/plan_cost.py [--torch-first] [lanes]"""
import sys
import time

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
args = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(args[0]) if args else 65536
if "--torch-first" in sys.argv:
    import torch
    torch.cuda.set_device(0)
    torch.empty(1, device="cuda")
    torch.cuda.synchronize()

import voice_synth_amd as vs
from voice_synth_amd import configs

specs, fs, dur, _ = configs.config_specs(3, n)
lanes, d = vs.lanes_from_specs(specs)
ns = vs.num_samples(fs, d)
t0 = time.perf_counter()
eng = vs.Engine(0)
t1 = time.perf_counter()
print("vs_ctx_create: %.1f ms" % ((t1 - t0) * 1e3))
for i in range(3):
    t0 = time.perf_counter()
    plan = eng.plan(lanes, ns)
    t1 = time.perf_counter()
    h, u = plan.timing()
    print("plan %d: wall %.1f ms, host_ms %.2f, upload_ms %.2f" % (i, (t1 - t0) * 1e3, h, u))
    if i == 0:
        out = eng.dev_alloc(n * ns * 2)
        t0 = time.perf_counter()
        plan.launch(vs.VS_KIND_SYNTH, out)
        plan.status()
        print("first launch + wait: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
        t0 = time.perf_counter()
        plan.launch(vs.VS_KIND_SYNTH, out)
        plan.status()
        print("second launch + wait: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
    plan.close()
eng.close()
