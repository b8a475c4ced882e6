"""How the config-3 launch time holds over tens of seconds (bench.py's `sustained` covers 2 s): back-to-back launches of
one plan for ~30 s, HIP-event time per block of 500 launches.  usage: python tools/long_sustained.py [seconds] [arith]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import voice_synth_amd as vs  # noqa: E402
from voice_synth_amd import configs  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
arith = vs.VS_ARITH_FMA if (len(sys.argv) > 2 and sys.argv[2] == "fma") else vs.VS_ARITH_EXACT
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream(dev)
specs, fs, dur, label = configs.config_specs(3, 65536)
lanes, d = vs.lanes_from_specs(specs)
ns = vs.num_samples(fs, d)
eng = vs.Engine(0, arith=arith, stream=stream.cuda_stream)
plan = eng.plan(lanes, ns)
out = torch.empty((65536, ns), dtype=torch.int16, device=dev)
print(label, plan.kernel_name(vs.VS_KIND_SYNTH))
for _ in range(5):
    plan.launch(vs.VS_KIND_SYNTH, out.data_ptr(), out_pitch=ns)
torch.cuda.synchronize(dev)
t_start = time.perf_counter()
block = 500
k = 0
while time.perf_counter() - t_start < seconds:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(block):
        plan.launch(vs.VS_KIND_SYNTH, out.data_ptr(), out_pitch=ns)
    e1.record(stream)
    torch.cuda.synchronize(dev)
    ms = e0.elapsed_time(e1) / block
    k += 1
    print("t = %5.1f s  block %2d: %.4f ms per launch = %.1f Gsamples/s" % (time.perf_counter() - t_start, k, ms, 65536 * ns / ms / 1e6), flush=True)
plan.status()
