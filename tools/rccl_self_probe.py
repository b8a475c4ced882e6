"""One rank, backend "nccl" (= RCCL), on the one GPU of the box: does torch.distributed's batched point-to-point path --
the one voice_synth_amd/dist.py's gather uses on both ends -- work in this image at all?  A batch of one isend and one irecv
to the rank itself (the only exchange RCCL allows on one device), on a side stream behind an event, as PipelinedGather
posts its chunks; then an all_reduce and an all_gather_object as bench.py's pre-flight makes them.
    python tools/rccl_self_probe.py"""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voice_synth_amd.dist import wire_view  # noqa: E402  (int16 PCM travels as bytes: RCCL's process group refuses int16)

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29551")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
t0 = time.time()
dist.init_process_group("nccl", device_id=dev)
print("init_process_group(nccl, device_id) %.2f s, backend %s" % (time.time() - t0, dist.get_backend()), flush=True)
t = torch.ones(1, device=dev)
dist.all_reduce(t)
t64 = torch.tensor([1.5, 2.5], dtype=torch.float64, device=dev)   # the types bench.py reduces: float64 (MAX), int32 (MIN)
dist.all_reduce(t64, op=dist.ReduceOp.MAX)
ok32 = torch.tensor([1], dtype=torch.int32, device=dev)
dist.all_reduce(ok32, op=dist.ReduceOp.MIN)
seen = [None]
dist.all_gather_object(seen, {"rank": 0})
print("all_reduce, all_gather_object ok:", float(t.item()), seen, flush=True)
rows, n = 4096, 16000
src = torch.randint(-32767, 32767, (rows, n), dtype=torch.int16, device=dev)
dst = torch.zeros_like(src)
side = torch.cuda.Stream(device=dev)
for rep in range(3):
    dst.zero_()
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream())
    t0 = time.time()
    with torch.cuda.stream(side):
        side.wait_event(ev)
        work = list(dist.batch_isend_irecv([dist.P2POp(dist.irecv, wire_view(dst), 0), dist.P2POp(dist.isend, wire_view(src), 0)]))
    for w in work:
        w.wait()
    side.synchronize()
    torch.cuda.current_stream().synchronize()
    ok = bool(torch.equal(src, dst))
    print("batch of irecv + isend to self, %d MB: %s, %.1f ms, %d work object(s)" % (src.numel() * 2 >> 20, "equal" if ok else "DIFFERENT", (time.time() - t0) * 1e3, len(work)), flush=True)
    if not ok:
        sys.exit(1)
# the ordering PipelinedGather relies on, with the real pieces: the LIBRARY's kernel enqueued on torch's current stream
# (through the stream handle bench.py hands to vs_ctx_set_stream), a torch event behind it, the side stream waits for the
# event, RCCL sends the chunk -- a send that did not wait for the kernel would ship the zeros written in front of it
import voice_synth_amd as vs  # noqa: E402
from voice_synth_amd import configs  # noqa: E402

rows = 16384
specs, fs, dur, label = configs.config_specs(3, rows)
lanes, d = vs.lanes_from_specs(specs)
ns = vs.num_samples(fs, d)
eng = vs.Engine(0, stream=torch.cuda.current_stream(dev).cuda_stream)
plan = eng.plan(lanes, ns)
want = torch.empty((rows, ns), dtype=torch.int16, device=dev)
plan.launch(vs.VS_KIND_SYNTH, want.data_ptr(), out_pitch=ns)
torch.cuda.synchronize(dev)
chunk = torch.empty_like(want)
got = torch.empty_like(want)
for rep in range(3):
    chunk.zero_()
    got.fill_(-1)
    plan.launch(vs.VS_KIND_SYNTH, chunk.data_ptr(), out_pitch=ns)      # enqueued, not waited for
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        side.wait_event(ev)
        work = list(dist.batch_isend_irecv([dist.P2POp(dist.irecv, wire_view(got), 0), dist.P2POp(dist.isend, wire_view(chunk), 0)]))
    for w in work:
        w.wait()
    side.synchronize()
    torch.cuda.current_stream().synchronize()
    ok = bool(torch.equal(got, want))
    print("library kernel -> event -> RCCL send to self, %d x %d samples: %s" % (rows, ns, "equal" if ok else "DIFFERENT"), flush=True)
    if not ok:
        sys.exit(1)
plan.close()
eng.close()
dist.barrier()
dist.destroy_process_group()
print("ok")
