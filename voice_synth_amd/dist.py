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


def shard_range(n_lanes, rank, world):
    """Contiguous lane block [lo, hi) of `rank`; blocks differ by at most one lane."""
    base, rem = divmod(int(n_lanes), int(world))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


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
                ops.append(dist.P2POp(dist.irecv, out[a:min(hi, a + step)], r))
    else:
        lo, hi = shard_range(n_lanes_total, rank, world)
        step = chunk_rows or max(1, hi - lo)
        for a in range(0, hi - lo, step):
            ops.append(dist.P2POp(dist.isend, local[a:min(hi - lo, a + step)].contiguous(), dst))
    # one coalesced group: with RCCL the seven receives of the root run concurrently, one per
    # xGMI link, instead of one after the other
    for q in dist.batch_isend_irecv(ops):
        q.wait()
    return out
