"""How the row pitch of the caller's PCM buffer changes the launch time (the kernels store 16 bytes per lane into 64
rows per instruction; the rows' distance decides which L2 channels one instruction touches).
    python tools/pitch_probe.py [config] [lanes]   ->  best of three rotations per pitch, exact and fma;
    PITCH_B2B=1: 16 launches back to back per figure (default: one launch, one wait, median of 7); PITCH_EXTRA_BYTES=0,128,...;
    PITCH_TORCH=1: the buffer from PyTorch's allocator"""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("PITCH_TORCH"):
    import torch   # BEFORE the library: both bring a HIP runtime, and the one loaded first has to be PyTorch's
import voice_synth_amd as vs
from voice_synth_amd import configs

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
specs, fs, dur, label = configs.config_specs(cfg, n)
lanes, d = vs.lanes_from_specs(specs)
ns = vs.num_samples(fs, d)
if os.environ.get("PITCH_TORCH"):
    import torch
    torch.cuda.init()
    torch.zeros(1, device="cuda:0")
eng = vs.Engine(0)
plan = eng.plan(lanes, ns)
extra = [int(x) for x in os.environ.get("PITCH_EXTRA_BYTES", "0,4,16,32,64,128,192,256,384,512,768,1024,2048,4096").split(",")]
pitches = [ns + e // 2 for e in extra]
if os.environ.get("PITCH_TORCH"):   # the buffer from PyTorch's allocator instead of a plain hipMalloc
    import torch
    tbuf = torch.empty((n * max(pitches),), dtype=torch.int16, device="cuda:0")
    out = tbuf.data_ptr()
else:
    out = eng.dev_alloc(n * max(pitches) * 2)
print("buffer at 0x%x (%s)" % (out, "torch" if os.environ.get("PITCH_TORCH") else "hipMalloc"))
res = {}
for rep in range(3):
    for arith, name in ((vs.VS_ARITH_EXACT, "exact"), (vs.VS_ARITH_FMA, "fma")):
        eng.set_arith(arith)
        # the first launches of a series run on a clock that is still settling (an earlier version of this script
        # measured the dense pitch first and found it 3-5 % slower than every other: that was the clock): 16 launches
        # nobody times, and the pitches in a different rotation every repetition
        for r in range(16):
            plan.launch(vs.VS_KIND_SYNTH, out, out_pitch=pitches[0])
        eng.synchronize()
        k = (rep * 5) % len(pitches)
        for p in pitches[k:] + pitches[:k]:
            ts = []
            if os.environ.get("PITCH_B2B"):   # launches back to back, one wait behind the last
                plan.launch(vs.VS_KIND_SYNTH, out, out_pitch=p)
                eng.synchronize()
                t0 = time.perf_counter()
                for r in range(16):
                    plan.launch(vs.VS_KIND_SYNTH, out, out_pitch=p)
                eng.synchronize()
                ts = [0.0] + [(time.perf_counter() - t0) / 16] * 2
            else:                              # one launch, one wait
                for r in range(8):
                    t0 = time.perf_counter()
                    plan.launch(vs.VS_KIND_SYNTH, out, out_pitch=p)
                    eng.synchronize()
                    ts.append(time.perf_counter() - t0)
            res.setdefault((name, p), []).append(statistics.median(ts[1:]) * 1e3)
print(label, "lanes", n, "samples", ns)
base = {a: min(res[(a, pitches[0])]) for a in ("exact", "fma")}
for p in pitches:
    print("pitch %6d samples (%6d B, %+5d B): exact %.3f ms (%+.1f %%)   fma %.3f ms (%+.1f %%)" % (
        p, 2 * p, 2 * (p - ns), min(res[("exact", p)]), 100 * (min(res[("exact", p)]) / base["exact"] - 1),
        min(res[("fma", p)]), 100 * (min(res[("fma", p)]) / base["fma"] - 1)), flush=True)
