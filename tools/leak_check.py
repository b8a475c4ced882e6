"""Repeated use of one context (plans, host pipeline, node entry, wide plans, trims) must not lose
device memory: free memory before and after, as torch sees it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import voice_synth_amd as vs
from voice_synth_amd import configs

specs, fs, dur, _ = configs.config_specs(3, 20000)
lanes, d = vs.lanes_from_specs(specs)
wide, _, _ = configs.wide_order_lanes([40] * 200)
rng = np.random.default_rng(1)
torch.cuda.init()
eng = vs.Engine(0)
eng.synth(lanes, 100)
eng.trim()
torch.cuda.synchronize()
free0 = torch.cuda.mem_get_info(0)[0]
for it in range(150):
    n_lanes = int(rng.integers(1, 20000))
    n = int(rng.integers(1, 3000))
    sub = (vs.Lane * n_lanes).from_buffer(lanes)
    eng.synth(sub, n)
    if it % 5 == 0:
        eng.source(sub, n)
        eng.filter(sub, np.zeros((n_lanes, n), dtype=np.int16))
    if it % 7 == 0:
        eng.synth(wide, n)
    if it % 11 == 0:
        node = vs.Node([0, 0, 0])
        buf = eng.dev_alloc(n_lanes * n * 2)
        node.synth_gather(sub, n, buf, n, vs.Node.OVERLAP | vs.Node.STAGE_ALL)
        eng.dev_free(buf)
        node.close()
    if it % 13 == 0:
        p = eng.plan(sub, n)
        p.close()
eng.trim()
torch.cuda.synchronize()
free1 = torch.cuda.mem_get_info(0)[0]
print("free device memory before %d MiB, after 150 rounds + trim %d MiB, difference %d KiB" % (free0 >> 20, free1 >> 20, (free0 - free1) >> 10))
eng.close()
sys.exit(0 if abs(free0 - free1) < (64 << 20) else 1)
