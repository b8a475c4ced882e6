"""PCIe-inclusive rate of the one-call host-buffer path (vs_synth: plan, allocate, launch, copy
the PCM back into pageable host memory) -- quoted in DESIGN.md, never used as bench.py's value."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import voice_synth_amd as vs
from voice_synth_amd import configs
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
specs, fs, dur, label = configs.config_specs(3, n)
lanes, d = vs.lanes_from_specs(specs)
ns = vs.num_samples(fs, d)
eng = vs.Engine(0)
eng.synth(lanes[:64], ns)
ts = []
for _ in range(3):
    t0 = time.perf_counter(); eng.synth(lanes, ns); ts.append(time.perf_counter() - t0)
t = min(ts)
print("vs_synth host path, %d x %d: %.1f ms  %.1f Msamples/s  (%.2f GB/s of PCM to the host)" % (n, ns, t * 1e3, n * ns / t / 1e6, 2 * n * ns / t / 1e9))
