"""What can run NEXT TO a launch of the fused kernel (it fills every CU for its whole duration)?  Times, from the host,
a few runtime calls issued right behind a launch: a pinned host-to-device copy on another stream, a pageable one,
hipMalloc / hipFree -- each against the same call on an idle device."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import voice_synth_amd as vs
from voice_synth_amd import configs

specs, fs, dur, label = configs.config_specs(3, 65536)
lanes, d = vs.lanes_from_specs(specs)
ns = vs.num_samples(fs, d)
dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
main = torch.cuda.current_stream(dev)
eng = vs.Engine(0, stream=main.cuda_stream)
plan = eng.plan(lanes, ns)
out = torch.empty((65536, ns), dtype=torch.int16, device=dev)
hip = C.CDLL("libamdhip64.so", mode=C.RTLD_GLOBAL)
side = torch.cuda.Stream(dev)
n = 8 << 20
pinned = torch.empty(n, dtype=torch.uint8).pin_memory()
pageable = torch.empty(n, dtype=torch.uint8)
dst = torch.empty(n, dtype=torch.uint8, device=dev)
small_pin = torch.zeros(64, dtype=torch.uint8).pin_memory()
small_dst = torch.empty(64, dtype=torch.uint8, device=dev)

def copy_pinned():
    with torch.cuda.stream(side):
        dst.copy_(pinned, non_blocking=True)
    side.synchronize()
def copy_small():
    with torch.cuda.stream(side):
        small_dst.copy_(small_pin, non_blocking=True)
    side.synchronize()
def copy_pageable():
    with torch.cuda.stream(side):
        dst.copy_(pageable, non_blocking=True)
    side.synchronize()
def malloc_free():
    p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), C.c_size_t(n)) == 0
    assert hip.hipFree(p) == 0
def malloc_only():
    p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), C.c_size_t(n)) == 0
    keep.append(p)
def sized(nbytes):
    src = torch.zeros(nbytes, dtype=torch.uint8).pin_memory()
    dstn = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    def fn():
        with torch.cuda.stream(side):
            dstn.copy_(src, non_blocking=True)
        side.synchronize()
    return fn
sizes = [("%d B pinned -> device" % b, sized(b)) for b in (1024, 4096, 16384, 65536, 262144, 1 << 20)]
keep = []
for name, fn in tuple(sizes) + (("8 MB pinned -> device, other stream, wait", copy_pinned), ("64 B pinned -> device", copy_small),
                 ("8 MB pageable -> device", copy_pageable), ("hipMalloc 8 MB (kept)", malloc_only), ("hipMalloc + hipFree 8 MB", malloc_free)):
    idle, busy = [], []
    for r in range(6):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter(); fn(); idle.append(time.perf_counter() - t0)
        torch.cuda.synchronize(dev)
        plan.launch(vs.VS_KIND_SYNTH, out.data_ptr(), out_pitch=ns)
        time.sleep(0.0003)
        t0 = time.perf_counter(); fn(); busy.append(time.perf_counter() - t0)
        torch.cuda.synchronize(dev)
    print("%-44s idle %.3f ms   behind a launch (2.6 ms kernel, 0.3 ms in) %.3f ms" % (name, sorted(idle)[2] * 1e3, sorted(busy)[2] * 1e3), flush=True)
for p in keep:
    hip.hipFree(p)
