"""Host cost of vs_plan_create for the BASELINE batches (vs_plan_timing: host_ms = expansion + order + tables, upload_ms =
allocation + upload + wait), median of 5 plans each; VS_LIB selects another build for a same-box comparison."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import voice_synth_amd as vs
from voice_synth_amd import configs
eng = vs.Engine(0)
for cfg, n in ((3, 65536), (5, 65536), (4, 32768), (2, 1024)):
    specs, fs, dur, label = configs.config_specs(cfg, n)
    lanes, d = vs.lanes_from_specs(specs)
    ns = vs.num_samples(fs, d)
    hs, us = [], []
    for _ in range(6):
        plan = eng.plan(lanes, ns)
        h, u = plan.timing()
        hs.append(h); us.append(u)
        plan.close()
    print("config %d (%6d lanes): host %.2f ms (min %.2f), upload %.2f ms   [first plan %.2f + %.2f]"
          % (cfg, n, statistics.median(hs[1:]), min(hs[1:]), statistics.median(us[1:]), hs[0], us[0]), flush=True)
