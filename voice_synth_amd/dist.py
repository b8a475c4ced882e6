"""Multi-GPU plumbing: utterances shard across ranks, one optional gather of the final PCM.

The path has no exchange step while it computes: utterances are independent (all carried state
is per utterance, reference flowgen_shimmer.c:121-122 and vowel_new.c:90), so ranks own
contiguous lane blocks and never talk during synthesis.  A lane's Philox key is a function of
its GLOBAL lane index, so an N-rank result equals the 1-rank result byte for byte.  The only
collective is the delivery of the finished int16 PCM to rank 0 (RCCL over xGMI on the GPU box,
gloo in the CPU tests): point-to-point sends, one per peer link, no ring.
"""
import torch
import torch.distributed as dist

from .configs import shard_range  # noqa: F401  (the cut itself needs no torch)


def wire_view(t):
    """The tensor as the transport sees it: PCM is int16, and PyTorch's NCCL/RCCL process group refuses that type
    ("Input tensor data type is not supported for NCCL process group: Short" -- found by tools/rccl_self_probe.py on the one
    GPU of a test box, before the first multi-GPU run could): the same memory as bytes, which every backend moves."""
    return t.view(torch.uint8) if t.dtype == torch.int16 else t


def gather_pcm(local, n_lanes_total, dst=0, chunk_rows=None):
    """Gathers row blocks [lanes_r, n_samples] (int16, any device) into one
    [n_lanes_total, n_samples] tensor on rank `dst`; other ranks get None.

    Each peer sends straight to `dst` (isend/irecv pairs posted together, so on xGMI every
    link into the root carries one stream).  chunk_rows splits a block into several messages
    so a caller can overlap delivery with the next batch."""
    rank, world = dist.get_rank(), dist.get_world_size()
    n_samples = local.shape[1]
    if world == 1:
        return local
    ops = []
    out = None
    if rank == dst:
        out = torch.empty((n_lanes_total, n_samples), dtype=local.dtype, device=local.device)
        lo, hi = shard_range(n_lanes_total, rank, world)
        out[lo:hi].copy_(local)
        for r in range(world):
            if r == dst:
                continue
            lo, hi = shard_range(n_lanes_total, r, world)
            step = chunk_rows or max(1, hi - lo)
            for a in range(lo, hi, step):
                ops.append(dist.P2POp(dist.irecv, wire_view(out[a:min(hi, a + step)]), r))
    else:
        lo, hi = shard_range(n_lanes_total, rank, world)
        step = chunk_rows or max(1, hi - lo)
        for a in range(0, hi - lo, step):
            ops.append(dist.P2POp(dist.isend, wire_view(local[a:min(hi - lo, a + step)].contiguous()), dst))
    # one coalesced group: with RCCL the seven receives of the root run concurrently, one per
    # xGMI link, instead of one after the other
    for q in dist.batch_isend_irecv(ops):
        q.wait()
    return out


def chunk_edges(rows, chunk_rows):
    """[(a, b), ...] covering [0, rows) in steps of chunk_rows"""
    step = max(1, int(chunk_rows))
    return [(a, min(rows, a + step)) for a in range(0, rows, step)]


class PipelinedGather:
    """Synthesis in chunks with the delivery to rank `dst` riding behind it: chunk k of every rank
    travels to the root (RCCL send/recv over xGMI: one transfer per peer link, grouped on the
    root, no ring) while chunk k+1 is being synthesised.

    Every rank owns the contiguous lane block shard_range(n_lanes_total, rank, world) and cuts it
    into chunks of chunk_rows utterances; the root synthesises its own chunks in place in the
    gathered buffer.  run(launch) calls launch(k, tensor) for each local chunk, in order; on a
    GPU the call must only ENQUEUE the work on the current stream.  Backend-agnostic: the CPU
    tests drive it with gloo and a synchronous launch."""

    def __init__(self, n_lanes_total, n_samples, chunk_rows, device, dtype=torch.int16, dst=0):
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.dst = dst
        self.n_total, self.n_samples, self.chunk_rows = int(n_lanes_total), int(n_samples), int(chunk_rows)
        self.lo, self.hi = shard_range(self.n_total, self.rank, self.world)
        self.edges = chunk_edges(self.hi - self.lo, self.chunk_rows)
        self.cuda = torch.device(device).type == "cuda"
        if self.rank == dst:
            self.full = torch.empty((self.n_total, self.n_samples), dtype=dtype, device=device)
            base = self.full[self.lo:self.hi]
        else:
            self.full = None
            base = torch.empty((self.hi - self.lo, self.n_samples), dtype=dtype, device=device)
        self.base = base  # this rank's rows, contiguous: [hi - lo, n_samples]
        self.chunks = [base[a:b] for a, b in self.edges]
        self.comm_stream = torch.cuda.Stream(device=device) if self.cuda else None

    def _send(self, chunk):
        """One chunk to the root, as a batch of ONE operation: the root posts its receives as a batch, and on the NCCL/RCCL
        backend a batched operation travels on the group's communicator of ALL ranks while a lone isend() may be given a
        two-rank communicator of its own (PyTorch's ProcessGroupNCCL keeps that for single point-to-point calls) -- a send
        and a receive on different communicators never meet.  Both ends batched: the same communicator whatever the
        version.  (gloo has no batches: the call falls back to the plain operation.)"""
        return list(dist.batch_isend_irecv([dist.P2POp(dist.isend, wire_view(chunk), self.dst)]))

    def run(self, launch, progress=None):
        """Returns the gathered tensor on the root, None elsewhere.  Blocks until delivery is done.
        progress (optional) is called once per round of chunks handed to the transport and once per completed
        transfer: a watchdog's sign of life (bench.py).

        A launch that RAISES on one rank (no room for a plan, a refused kernel) must not leave the others waiting for
        transfers that never come -- an unmatched send or receive never completes.  The failing rank therefore stops
        launching but keeps to the schedule: it hands over the rest of its chunks as they are (their rows are then
        whatever the buffer held: the caller must not use the gathered tensor), waits for the exchange like everybody
        else, and raises the launch's exception only THEN.  Every rank so leaves run() in step, and the agreement that
        ends the caller's phase (bench.py run_phase: an all-reduce of an ok flag) takes them out together."""
        failed = None
        work = []
        peers = [r for r in range(self.world) if r != self.dst]
        peer_edges = {}
        for r in peers:
            lo, hi = shard_range(self.n_total, r, self.world)
            peer_edges[r] = [(lo + a, lo + b) for a, b in chunk_edges(hi - lo, self.chunk_rows)]
        n_rounds = max([len(self.edges)] + [len(e) for e in peer_edges.values()]) if self.rank == self.dst else len(self.edges)
        for k in range(n_rounds):
            ev = None
            if k < len(self.chunks):
                if failed is None:
                    try:
                        launch(k, self.chunks[k])
                    except Exception as exc:  # noqa: BLE001 - re-raised behind the exchange
                        failed = exc
                if self.cuda:
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream())
            if self.world == 1:
                continue
            if self.rank != self.dst:
                if self.cuda:
                    if dist.get_backend() != "nccl":
                        # gloo reads a device tensor from the host and knows nothing of streams: the
                        # chunk must be complete before it is handed over (rehearsal runs only; with
                        # RCCL the send is enqueued behind the event on the side stream)
                        ev.synchronize()
                    with torch.cuda.stream(self.comm_stream):
                        self.comm_stream.wait_event(ev)
                        work += self._send(self.chunks[k])
                else:
                    work += self._send(self.chunks[k])
            else:
                ops = [dist.P2POp(dist.irecv, wire_view(self.full[peer_edges[r][k][0]:peer_edges[r][k][1]]), r)
                       for r in peers if k < len(peer_edges[r])]
                if ops:
                    if self.cuda:
                        with torch.cuda.stream(self.comm_stream):
                            work += dist.batch_isend_irecv(ops)
                    else:
                        work += dist.batch_isend_irecv(ops)
            if progress:
                progress()
        for w in work:
            w.wait()
            if progress:
                progress()
        if self.cuda:
            self.comm_stream.synchronize()
            torch.cuda.current_stream().synchronize()
        if failed is not None:
            raise failed
        return self.full
