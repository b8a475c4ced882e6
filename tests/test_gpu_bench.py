"""bench.py contract on the GPU box: one JSON line with the driver's keys + roofline +
cpu_baseline; and the torch.distributed.run launch path (world size 1 here: RCCL init, barrier,
all-reduce of the timings are exercised; more ranks are the driver's to launch)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline"]


def _check(line, steps):
    d = json.loads(line)
    for k in KEYS:
        assert k in d, k
    assert d["unit"] == "Msamples/s" and d["steps"] == steps and d["scaling"] == "weak"
    assert d["dtype"] == "f64" and d["vs_baseline"] is None and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["kernel"].startswith("vs_synth") and r["kernel_ms_avg"] > 0
    assert d["launch_health_word"] == 0                     # vs_plan_status after the timed launches
    # the sustained figure next to the burst: >= 2 s of back-to-back launches of the same plan
    sus = d["sustained"]
    assert sus["seconds"] >= 1.9 and sus["launches"] >= 50 and sus["kernel_ms_avg"] > 0
    assert abs(sus["ratio_to_timed_region"] - sus["Msamples/s"] / d["value"]) < 2e-3
    # PMC figures are copied from profiles/: they say which tree they are from, and are null when it is not this one
    prof = r["profile"]
    assert len(prof["tree_kernel_sources_sha16"]) == 16
    for name in ("traffic", "valu"):
        if r[name] is not None:
            assert prof[name]["matches_tree"] is True and prof[name]["profile_head"]
        if prof[name] is not None and not prof[name]["matches_tree"]:
            assert r[name] is None
    # every rank says which device it drove
    assert len(d["ranks_seen"]) == d["n_gpus"] and d["distinct_devices"] >= 1
    assert all(len(x["pci_bus_id"]) >= 7 for x in d["ranks_seen"])
    assert d["plan"]["host_ms"] >= 0 and d["plan"]["upload_ms"] >= 0
    return d


def test_bench_single_process_small():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--lanes", "4096", "--steps", "3", "--warmup", "1",
                          "--cpu-seconds", "0.5"], capture_output=True, cwd=ROOT, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _check(out.stdout.decode().strip().splitlines()[-1], 3)
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1
    assert d["cpu_baseline"]["gpu_rows_checked"] >= 1 and d["cpu_baseline"]["gpu_mismatched_samples"] == 0
    _check_row_pitch(d, 16064)                               # rows at the pitch vs_row_pitch() names, compared with the oracle
    assert d["other_arith"]["launches"] >= 10 and d["other_arith"]["kernel_ms_min"] <= d["other_arith"]["kernel_ms_median"]
    # the tolerance modes carry their distance from the CPU sample: fma within one LSB, f32 within its measured bounds
    assert d["other_arith"]["rms_vs_c_ref"]["max_abs_lsb"] <= 1 and d["other_arith"]["rms_vs_c_ref"]["normalised"] <= 1e-5
    f32 = d["f32_arith"]
    assert f32["kernel"].startswith("vs_synth_ws_kernel<2,") and f32["launches"] >= 10 and f32["kernel_ms_min"] > 0
    assert f32["rms_vs_c_ref"]["rows_checked"] >= 1024 and 0 < f32["rms_vs_c_ref"]["normalised"] < 4e-5 and f32["rms_vs_c_ref"]["max_abs_lsb"] <= 32
    ref = d["cpu_baseline"].get("reference_as_shipped")
    if ref:                                                  # oracle/_ref travels to the GPU box
        assert ref["matches_port"] is True
        for key in ("O0_as_shipped", "O2"):
            assert ref[key]["pipelines"] >= 4096 and ref[key]["value"] > 0, ref[key]
            # the two programs account for the workers' time (no launcher in the loop)
            assert ref[key]["ms_in_flowgen"] + ref[key]["ms_in_vowel"] > 0.8 * ref[key]["ms_per_pipeline_per_worker"]
            assert ref[key]["workers"] >= 4


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _check_row_pitch(d, want):
    assert d["config"]["row_pitch_samples"] == want and d["config"]["samples_per_utterance"] == 16000
    ab = d["row_pitch_ab"]
    assert ab["pitched"]["row_pitch_samples"] == 16064 and ab["dense"]["row_pitch_samples"] == 16000
    for rows in ("pitched", "dense"):
        for arith in ("exact", "fma"):
            assert ab[rows][arith]["kernel_ms_avg"] > 0


def test_bench_dense_rows():
    """the PCM buffer's rows are the caller's: at the pitch vs_row_pitch() names by default (test_bench_single_process_small),
    dense with --dense-rows; the line says which and carries the interleaved comparison of the two either way"""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--lanes", "4096", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
           "--no-other-configs", "--dense-rows"]
    out = subprocess.run(cmd, capture_output=True, cwd=ROOT, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    _check_row_pitch(_check(out.stdout.decode().strip().splitlines()[-1], 2), 16000)


def test_bench_under_torch_distributed_run():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--lanes", "4096", "--steps", "3",
           "--warmup", "1", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, cwd=ROOT, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    # standard output is the ONE line: RCCL's banner (it writes five lines to stdout when a communicator is made) and
    # anything else a library prints have been sent to standard error (bench.reserve_stdout)
    lines = [l for l in out.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), lines[:8]
    assert b"RCCL version" not in out.stdout
    _check(lines[-1], 3)


def test_bench_two_ranks_rehearsal_on_one_device():
    """the N = 2 control flow of bench.py on a one-GPU box: two ranks share device 0 and talk over gloo
    (VS_BENCH_REHEARSAL=1; RCCL refuses two ranks on one device, so the gather leg is left out).  Each
    rank synthesises its own block of the global batch; rank 0 prints the whole-job line."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--lanes", "4096", "--steps", "3",
           "--warmup", "1", "--config4-lanes", "1000"]
    out = subprocess.run(cmd, capture_output=True, cwd=ROOT, timeout=600, env=dict(os.environ, VS_BENCH_REHEARSAL="1"))
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1                      # rank 0 only
    d = json.loads(lines[-1])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak"
    assert d["config"]["utterances_per_gpu"] == 4096
    # the line shows WHICH devices took part: two ranks, and -- rehearsal -- one device between them
    assert [x["rank"] for x in d["ranks_seen"]] == [0, 1] and d["distinct_devices"] == 1 and "rehearsal" in d
    assert d["config4"]["distinct_devices"] == 1
    # whole-job value = the units of both ranks over the slowest rank's time
    assert abs(d["value"] - 2 * 4096 * 16000 * 3 / (d["ms_per_step"] * 3e-3) / 1e6) / d["value"] < 0.01
    assert "cpu_baseline" not in d              # rank 0 at N = 1 only
    # the configuration BASELINE.json names for N > 1 rides along: config 4 cut over the ranks
    c4 = d["config4"]
    assert "error" not in c4, c4
    assert c4["utterances"] == 1000 and c4["utterances_per_gpu"] == 500 and c4["samples_per_utterance"] == 44100
    assert c4["value"] > 0 and c4["roofline_per_gpu"]["kernel"].startswith("vs_synth")
    assert "gather" not in c4                   # VS_BENCH_REHEARSAL=1: no exchange of device tensors


def test_bench_three_ranks_rehearsal_with_the_gather_leg():
    """VS_BENCH_REHEARSAL=2: three ranks on device 0 over gloo, the config-4 block WITH its pipelined gather
    (ragged: 1000 utterances over three ranks, chunks of 16384 -> one chunk each): the gathered PCM must
    equal the un-overlapped gather.  What this cannot show is RCCL itself."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "3", "--lanes", "2048", "--steps", "2",
           "--warmup", "1", "--config4-lanes", "1000"]
    out = subprocess.run(cmd, capture_output=True, cwd=ROOT, timeout=900, env=dict(os.environ, VS_BENCH_REHEARSAL="2"))
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith("{")]
    d = json.loads(lines[-1])
    c4 = d["config4"]
    assert d["n_gpus"] == 3 and "error" not in c4, c4
    assert c4["utterances_per_gpu"] == 334       # rank 0's block of 1000 over 3
    g = c4["gather"]
    assert g["backend"] == "gloo" and g["overlapped"] is True and g["equals_unoverlapped_gather"] is True
    assert g["distinct_devices"] == 1 and len(g["devices"]) == 3      # a rehearsal: three ranks, one device
    assert g["bytes_into_rank0"] == (1000 - 334) * 44100 * 2 and c4["value_with_gather"] > 0


def test_bench_plain_invocation_launches_its_own_ranks():
    """`python3 bench.py --gpus 2 ...` WITHOUT a launcher (the shape of the driver's N = 1 command): bench.py starts
    `python -m torch.distributed.run` as a child process before anything touches the GPU, relays rank 0's line and
    leaves with the child's exit code.  Rehearsal: both ranks share device 0 over gloo, config-4 block with its gather."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--lanes", "4096",
           "--config4-lanes", "1000"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    out = subprocess.run(cmd, capture_output=True, cwd=ROOT, timeout=900, env=dict(env, VS_BENCH_REHEARSAL="2"))
    assert out.returncode == 0, out.stderr[-3000:]
    assert b"without a launcher" in out.stderr
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1                      # ONE line: rank 0's, relayed unchanged
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and len(d["ranks_seen"]) == 2 and [x["rank"] for x in d["ranks_seen"]] == [0, 1]
    assert d["steps"] == 2 and "error" not in d["config4"], d["config4"]
    assert d["config4"]["gather"]["equals_unoverlapped_gather"] is True


def test_bench_plain_invocation_propagates_the_watchdog_exit():
    """a rank that never reaches a phase's collectives: the phase watchdog ends every rank with a non-zero code after
    rank 0 has printed the line with what it has -- and the plain invocation leaves with a non-zero code too"""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--lanes", "4096",
           "--config4-lanes", "1000"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    out = subprocess.run(cmd, capture_output=True, cwd=ROOT, timeout=900,
                         env=dict(env, VS_BENCH_REHEARSAL="1", VS_BENCH_FAULT="stall_rank1", VS_BENCH_PHASE_DEADLINE_S="8"))
    assert out.returncode != 0, out.stdout[-2000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and "no progress" in d["config4"]["error"]


def test_bench_line_carries_every_baseline_configuration():
    """the default workload (config 3, full size) with the CPU legs switched off: `other_configs` holds configs 2,
    4 (one GPU's shard) and 5 in both arithmetic contracts, each with its kernel, time and roofline fraction -- and config 3
    once more with the vowel stage's own noise (vowel -n) on every utterance"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                         capture_output=True, cwd=ROOT, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _check(out.stdout.decode().strip().splitlines()[-1], 3)
    oc = d["other_configs"]
    assert [(r["baseline_config_index"], r["arith"]) for r in oc] == [(c, a) for c in (1, 3, 4, 2) for a in ("exact", "fma", "f32")]
    # the last three: config 3 with "vowel -n 20" on every utterance -- the fused kernel that takes the frame powers along,
    # the scan, the streaming noise pass: one launch of the plan, bracketed as a whole
    for r in oc[9:]:
        assert r["vowel_n_db"] == 20 and r["bytes_per_sample"] == 6
        assert r["kernel"].startswith("vs_synth_ws_pow_kernel<") and r["kernel"].endswith("+ vs_out_power_fill_kernel + vs_out_noise_kernel")
    assert oc[9]["kernel_ms_avg"] > d["roofline"]["kernel_ms_min"]          # (exact against exact)
    assert all("profile" in r and "profile_key" in r["profile"] for r in oc)
    for r in oc:
        assert "error" not in r and r["kernel"].startswith("vs_synth") and r["kernel_ms_avg"] >= r["kernel_ms_min"] > 0
        assert abs(r["roofline_frac"] - 2 * r["utterances"] * r["samples_per_utterance"] / (r["kernel_ms_avg"] * 1e-3) / 8e12) < 2e-4
    assert oc[3]["utterances"] == 32768 and oc[3]["samples_per_utterance"] == 44100
    # ... and what a caller pays who makes a plan per batch of new utterances (plan k + 1 overlapped with kernel k)
    fb = d["fresh_batches"]
    assert "error" not in fb and fb["batches"] >= 10 and fb["utterances_per_batch"] == 65536
    assert fb["ms_per_batch"] >= 0.9 * d["roofline"]["kernel_ms_min"] and fb["plan_host_ms_median"] > 0
    assert abs(fb["Msamples/s"] - 65536 * 16000 / (fb["ms_per_batch"] * 1e-3) / 1e6) / fb["Msamples/s"] < 1e-3
    assert 0.9 * d["roofline"]["kernel_ms_min"] <= fb["reseed_ms_per_batch"] <= fb["ms_per_batch"]   # new draws cost less than new plans
    # ... and the same from plain C (cli/vs_bench.c --fresh: a second host thread plans, plans are destroyed as it goes):
    # close to the same program launching ONE plan over and over (1.02 x in profiles/r06_fresh_from_c.txt)
    fc = d["fresh_batches_c"]
    assert "error" not in fc and fc["batches"] == 50 and fc["last_batch_equals_a_plain_launch"] is True
    assert fc["plan_destroy_ms_avg"] < 0.5 and 0.9 <= fc["ratio_to_same_plan"] <= 1.12, fc   # (two process starts: a few per cent of noise)


def test_pcm_travels_through_rccls_process_group():
    """One rank over backend "nccl" (= RCCL) on this box's one GPU: the batched send / receive that both ends of
    voice_synth_amd/dist.py's gather post -- int16 PCM as bytes (wire_view: RCCL's process group refuses int16, which no
    gloo rehearsal could have shown), on a side stream behind an event -- and the reductions bench.py makes, against the real
    librccl (tools/rccl_self_probe.py; a send to the rank itself is the only exchange RCCL allows on one device)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["MASTER_PORT"] = str(_free_port())
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_self_probe.py")], capture_output=True, text=True,
                       timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    # three plain batches, then three in which the chunk is written by the LIBRARY's kernel on torch's stream and sent
    # behind a torch event -- the ordering PipelinedGather relies on (the gloo rehearsals wait on the host instead)
    assert r.stdout.count("equal") == 6 and "DIFFERENT" not in r.stdout and r.stdout.strip().endswith("ok")
