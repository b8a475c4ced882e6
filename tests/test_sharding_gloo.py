"""The N>1 path on CPU: two gloo ranks shard a batch, synthesise their blocks (the CPU oracle
stands in for the device here -- this test is about the sharding/gather plumbing of
voice_synth_amd/dist.py, which is backend-agnostic) and gather the PCM to rank 0.  The result
must equal the single-rank result byte for byte: a lane's draw stream is keyed by its GLOBAL
index, not by its placement."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import voice_synth_amd as vs
from voice_synth_amd import configs
from voice_synth_amd.dist import PipelinedGather, gather_pcm, shard_range
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


def _worker_pipelined(rank, world, port, out_path, n_lanes, chunk_rows):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        specs, fs, dur, _ = configs.config_specs(3, n_lanes)
        lanes_all, d = vs.lanes_from_specs(specs)
        n = 1500
        pg = PipelinedGather(n_lanes, n, chunk_rows, "cpu")

        def launch(k, tensor):           # the CPU oracle stands in for the device
            a, b = pg.edges[k]
            rows = po.synth([lanes_all[pg.lo + i] for i in range(a, b)], n, threads=1)
            tensor.copy_(torch.from_numpy(rows))

        dist.barrier()
        full = pg.run(launch)
        if rank == 0:
            np.save(out_path, full.numpy())
        else:
            assert full is None
    finally:
        dist.destroy_process_group()


def test_pipelined_gather_three_ranks_ragged_chunks(tmp_path):
    """chunk k of every rank travels to the root while chunk k+1 is produced; ranks differ in their
    number of chunks (13 lanes over 3 ranks = 5 + 4 + 4, chunks of 2)"""
    out = str(tmp_path / "pipelined.npy")
    mp.spawn(_worker_pipelined, args=(3, _free_port(), out, 13, 2), nprocs=3, join=True)
    got = np.load(out)
    specs, fs, dur, _ = configs.config_specs(3, 13)
    lanes, d = vs.lanes_from_specs(specs)
    assert np.array_equal(got, po.synth(lanes, 1500, threads=2))


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


# ---- bench.py's multi-rank control flow (no device needed: the phases only need torch.distributed) ----

def _worker_phases(rank, world, port, out_dir, mode):
    import json
    import sys
    import time

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    dist.init_process_group("gloo", rank=rank, world_size=world)
    phases = bench.Phases()
    rec = {"rank": rank}
    try:
        rec["first"] = bench.run_phase(phases, "fine", 30, lambda: rank * 10, rank, "cpu")
        if mode == "one_rank_raises":
            def fn():
                if rank == 1:
                    raise MemoryError("no room on rank 1")
                return "ok"
            try:
                bench.run_phase(phases, "second", 30, fn, rank, "cpu")
                rec["second"] = "passed"
            except bench.PhaseFailed as exc:
                rec["second"] = str(exc)
            # every rank is still in step: a later collective works
            t = torch.tensor([rank + 1])
            dist.all_reduce(t)
            rec["after"] = int(t.item())
        elif mode == "silent_phase":
            phases.enter("silent", 0.4)
            time.sleep(0.1)
            rec["early"] = phases.stalled()
            phases.tick()
            time.sleep(0.3)
            rec["after_tick"] = phases.stalled()       # 0.3 s since the tick: still inside the allowance
            time.sleep(0.3)
            rec["late"] = list(phases.stalled() or [])
            phases.leave()
            rec["left"] = phases.stalled()
    finally:
        json.dump(rec, open(os.path.join(out_dir, "r%d.json" % rank), "w"))
        dist.destroy_process_group()


def test_a_phase_that_fails_on_one_rank_takes_all_ranks_out_together(tmp_path):
    """bench.py's config-4 block runs in phases that end with an all-reduce of an ok flag (run_phase): rank 1 raising
    inside a phase must end the phase with PhaseFailed on EVERY rank -- nobody is left behind in the next collective"""
    import json
    mp.spawn(_worker_phases, args=(2, _free_port(), str(tmp_path), "one_rank_raises"), nprocs=2, join=True)
    r0, r1 = (json.load(open(tmp_path / ("r%d.json" % r))) for r in (0, 1))
    assert r0["first"] == 0 and r1["first"] == 10
    assert "another rank" in r0["second"] and "no room on rank 1" in r1["second"]
    assert r0["after"] == 3 and r1["after"] == 3


def _worker_gather_with_a_failing_launch(rank, world, port, out_dir):
    import json
    import sys

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    dist.init_process_group("gloo", rank=rank, world_size=world)
    phases = bench.Phases()
    rec = {"rank": rank, "launched": []}
    try:
        n_lanes, n, chunk_rows = 17, 64, 2          # ragged shards of 6 / 6 / 5 rows: three rounds
        pg = PipelinedGather(n_lanes, n, chunk_rows, "cpu")

        def launch(k, tensor):
            if rank == 1 and k == 1:
                raise MemoryError("no room for chunk 1 on rank 1")     # mid-round: rank 1 has already sent chunk 0
            tensor.fill_(100 * rank + k)
            rec["launched"].append(k)

        def gather():
            pg.run(launch, progress=phases.tick)
            return "gathered"

        try:
            bench.run_phase(phases, "timed gather", 60, gather, rank, "cpu")
            rec["phase"] = "passed"
        except bench.PhaseFailed as exc:
            rec["phase"] = str(exc)
        # nobody is stuck in a transfer: the next collective works on every rank
        t = torch.tensor([rank + 1])
        dist.all_reduce(t)
        rec["after"] = int(t.item())
        if rank == 0:
            # what the healthy ranks sent did arrive (the caller discards the tensor all the same)
            lo2, hi2 = shard_range(n_lanes, 2, world)
            rec["rank2_rows"] = pg.full[lo2:hi2, 0].tolist()
    finally:
        json.dump(rec, open(os.path.join(out_dir, "g%d.json" % rank), "w"))
        dist.destroy_process_group()


def test_a_launch_that_fails_mid_gather_takes_all_ranks_out_together(tmp_path):
    """PipelinedGather.run with a launch that raises on ONE rank in the middle of the rounds (three gloo ranks): the
    failing rank keeps to the send schedule and raises behind the exchange, so the root's receives complete, nobody
    hangs, and the phase agreement (bench.run_phase) ends the phase with PhaseFailed on every rank together"""
    import json
    mp.spawn(_worker_gather_with_a_failing_launch, args=(3, _free_port(), str(tmp_path)), nprocs=3, join=True)
    r = [json.load(open(tmp_path / ("g%d.json" % k))) for k in range(3)]
    assert "no room for chunk 1 on rank 1" in r[1]["phase"] and r[1]["launched"] == [0]      # it stopped launching ...
    assert "another rank" in r[0]["phase"] and "another rank" in r[2]["phase"]               # ... and everybody knows
    assert r[0]["launched"] == [0, 1, 2] and r[2]["launched"] == [0, 1, 2]
    assert [x["after"] for x in r] == [6, 6, 6]
    assert r[0]["rank2_rows"] == [200, 200, 201, 201, 202]


def test_watchdog_stamps(tmp_path):
    """Phases: the watchdog sees a phase as stalled only once it has shown no progress for its allowance; tick() is
    progress; outside a phase nothing is ever stalled"""
    import json
    mp.spawn(_worker_phases, args=(1, _free_port(), str(tmp_path), "silent_phase"), nprocs=1, join=True)
    r = json.load(open(tmp_path / "r0.json"))
    assert r["early"] is None and r["after_tick"] is None and r["late"] == ["silent", 0] and r["left"] is None


def test_pipelined_gather_eight_ranks(tmp_path):
    """the world size the driver's scaling run ends at: eight ranks, 37 lanes (5 + 5 + 5 + 5 + 5 + 4 + 4 + 4), chunks of
    2 -- the root posts three rounds of receives, the peers with four lanes sit the third one out"""
    out = str(tmp_path / "pipelined8.npy")
    mp.spawn(_worker_pipelined, args=(8, _free_port(), out, 37, 2), nprocs=8, join=True)
    got = np.load(out)
    specs, fs, dur, _ = configs.config_specs(3, 37)
    lanes, d = vs.lanes_from_specs(specs)
    assert np.array_equal(got, po.synth(lanes, 1500, threads=2))


def test_plain_bench_invocation_starts_its_own_ranks_and_propagates_their_exit_code():
    """`python bench.py --gpus 2` without a launcher starts `python -m torch.distributed.run` as a CHILD process before
    it has imported torch; here (no GPU) both ranks fail at once, and the parent must leave with a non-zero code too --
    not with the usage message of earlier rounds, and not with 0."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: tests/test_gpu_bench.py runs the same path to the end")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, cwd=root, timeout=300, env=env)
    err = out.stderr.decode(errors="replace")
    assert out.returncode not in (0, None), err[-2000:]
    assert "without a launcher" in err and "torch.distributed.run" in err and "--nproc-per-node 2" in err
    assert "needs `python -m torch.distributed.run" not in err
    # under a launcher with the wrong world size it still refuses
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, cwd=root, timeout=120,
                         env=dict(env, WORLD_SIZE="3", RANK="0"))
    assert out.returncode != 0 and b"WORLD_SIZE=3 but --gpus 2" in out.stderr


def test_node_run_preflight_findings():
    """bench.py --gpus N refuses to time a mis-bound launch (bench.preflight_errors; the N > 1 run that matters is the
    driver's, unattended): N ranks on fewer than N devices, a rank that came up on another backend, a communicator that
    does not span all ranks"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    def seen(n, pci=lambda r: "0000:%02x:00.0" % (5 + r), backend="nccl", size=None):
        return [{"rank": r, "host": "box", "pci_bus_id": pci(r), "backend": backend, "comm_world_size": size or n} for r in range(n)]

    assert bench.preflight_errors(seen(8), 8) == []
    two_on_one = bench.preflight_errors(seen(8, pci=lambda r: "0000:%02x:00.0" % (5 + r // 2)), 8)
    assert len(two_on_one) == 1 and "8 ranks drive 4 distinct devices" in two_on_one[0]
    assert any("gloo" in w for w in bench.preflight_errors(seen(2, backend="gloo"), 2))
    assert any("communicator sizes" in w for w in bench.preflight_errors(seen(4, size=2), 4))
    assert any("ranks reported" in w for w in bench.preflight_errors(seen(3), 4))


def test_pcm_goes_on_the_wire_as_bytes():
    """int16 PCM is handed to the process group as a byte view of the same memory (RCCL's refuses int16)"""
    import torch
    from voice_synth_amd.dist import wire_view

    t = torch.arange(12, dtype=torch.int16).reshape(3, 4)
    v = wire_view(t[1:])
    assert v.dtype == torch.uint8 and v.shape == (2, 8) and v.data_ptr() == t[1:].data_ptr()
    v[0, 0] = 0x7F
    assert int(t[1, 0]) == (4 & 0xFF00) | 0x7F
    f = torch.zeros(2, dtype=torch.float32)
    assert wire_view(f) is f
