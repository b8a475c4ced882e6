"""Is the row-pitch effect a property of ONE launch, or of a launch that follows a launch with the same rows?
Blocks of 8 launches at one pitch against launches that alternate between two pitches (wall clock, one wait per block)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import voice_synth_amd as vs
from voice_synth_amd import configs
specs, fs, dur, label = configs.config_specs(3, 65536)
lanes, d = vs.lanes_from_specs(specs)
ns = vs.num_samples(fs, d)
eng = vs.Engine(0)
plan = eng.plan(lanes, ns)
P0, P1 = ns, vs.row_pitch(ns)
out = eng.dev_alloc(65536 * P1 * 2)
out2 = eng.dev_alloc(65536 * P1 * 2)
def run(seq, bufs):
    plan.launch(vs.VS_KIND_SYNTH, bufs[0], out_pitch=seq[0]); eng.synchronize()
    t0 = time.perf_counter()
    for i, p in enumerate(seq):
        plan.launch(vs.VS_KIND_SYNTH, bufs[i % len(bufs)], out_pitch=p)
    eng.synchronize()
    return (time.perf_counter() - t0) / len(seq) * 1e3
for rep in range(3):
    print("rep %d: dense x16 %.3f | pitched x16 %.3f | alternating dense/pitched %.3f | dense, two buffers in turn %.3f | pitched, two buffers in turn %.3f"
          % (rep, run([P0] * 16, [out]), run([P1] * 16, [out]), run([P0, P1] * 8, [out]), run([P0] * 16, [out, out2]), run([P1] * 16, [out, out2])), flush=True)
