"""The N>1 path on CPU: two gloo ranks shard a batch, synthesise their blocks (the CPU oracle
stands in for the device here -- this test is about the sharding/gather plumbing of
voice_synth_amd/dist.py, which is backend-agnostic) and gather the PCM to rank 0.  The result
must equal the single-rank result byte for byte: a lane's draw stream is keyed by its GLOBAL
index, not by its placement."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import voice_synth_amd as vs
from voice_synth_amd import configs
from voice_synth_amd.dist import gather_pcm, shard_range
from oracle import pyoracle as po

N_LANES = 11  # odd on purpose: ragged shards


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard_range(N_LANES, rank, world)
        specs, fs, dur, _ = configs.config_specs(3, hi - lo, lane0=lo)
        lanes, d = vs.lanes_from_specs(specs)
        n = vs.num_samples(fs, d)
        local = torch.from_numpy(po.synth(lanes, n, threads=1))
        dist.barrier()
        full = gather_pcm(local, N_LANES, dst=0, chunk_rows=2)
        if rank == 0:
            np.save(out_path, full.numpy())
        else:
            assert full is None
    finally:
        dist.destroy_process_group()


def test_shard_range_partitions_exactly():
    for n in (1, 7, 64, 65536, 262144, 11):
        for w in (1, 2, 3, 4, 8):
            blocks = [shard_range(n, r, w) for r in range(w)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gather_equals_single_rank(tmp_path):
    out = str(tmp_path / "gathered.npy")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    specs, fs, dur, _ = configs.config_specs(3, N_LANES)
    lanes, d = vs.lanes_from_specs(specs)
    want = po.synth(lanes, vs.num_samples(fs, d), threads=2)
    assert got.shape == want.shape and np.array_equal(got, want)
