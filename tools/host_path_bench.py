"""PCIe-inclusive rate of the host-buffer path of the C ABI (vs_synth: plan per compute chunk,
launch, chunked delivery through pinned staging buffers) -- quoted in DESIGN.md, never used as
bench.py's value.  Three destinations: a pageable numpy array that is REUSED between calls (a
server's steady state; the first call also pays the page faults of a fresh 2 GB allocation), pinned
memory from vs_host_alloc (DMA straight into it), and the row callback (vs_synth_rows)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import voice_synth_amd as vs
from voice_synth_amd import configs
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
specs, fs, dur, label = configs.config_specs(3, n)
lanes, d = vs.lanes_from_specs(specs)
ns = vs.num_samples(fs, d)
eng = vs.Engine(0)
lib = vs.load()
eng.synth(lanes[:64], ns)
def rate(t):
    return "%.1f ms  %.1f Msamples/s  (%.2f GB/s of PCM to the host)" % (t * 1e3, n * ns / t / 1e6, 2 * n * ns / t / 1e9)
# pageable destination, reused
out = np.empty((n, ns), dtype=np.int16)
ts = []
for _ in range(4):
    t0 = time.perf_counter()
    vs.check(lib.vs_synth(eng._ctx, lanes, n, ns, out.ctypes.data), "vs_synth")
    ts.append(time.perf_counter() - t0)
print("vs_synth -> pageable buffer, first call (page faults of a fresh allocation): " + rate(ts[0]), flush=True)
print("vs_synth -> pageable buffer, reused:  " + rate(min(ts[1:])), flush=True)
ref = out[::997].copy()
# pinned destination
p = C.c_void_p()
vs.check(lib.vs_host_alloc(eng._ctx, n * ns * 2, C.byref(p)), "vs_host_alloc")
ts = []
for _ in range(4):
    t0 = time.perf_counter()
    vs.check(lib.vs_synth(eng._ctx, lanes, n, ns, p), "vs_synth")
    ts.append(time.perf_counter() - t0)
print("vs_synth -> pinned buffer (vs_host_alloc): " + rate(min(ts[1:])), flush=True)
pin = np.frombuffer((C.c_int16 * (n * ns)).from_address(p.value), dtype=np.int16).reshape(n, ns)
print("pinned result equals pageable result on sampled rows:", bool(np.array_equal(pin[::997], ref)), flush=True)
lib.vs_host_free(eng._ctx, p)
# callback that only touches the data
def fn(row0, rows):
    return 0
ts = []
for _ in range(3):
    t0 = time.perf_counter(); eng.synth_rows(lanes, ns, fn); ts.append(time.perf_counter() - t0)
print("vs_synth_rows, callback that returns at once (staging DMA only): " + rate(min(ts[1:])), flush=True)
# plan creation cost, for the record
plan = eng.plan(lanes, ns)
print("vs_plan_create of the whole batch: host %.2f ms, upload %.2f ms" % plan.timing(), flush=True)
