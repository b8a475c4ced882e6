"""The two drop-in programs on the GPU: same argv as the reference -> same file bytes (72-byte
LP64 header variant, which is what the compiled reference writes here) and the same per-cycle
stdout lines.  Expected data: tests/golden (reference outputs), not the oracle."""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

import voice_synth_amd as vs

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = {c["name"]: c for c in json.load(open(os.path.join(GOLD, "manifest.json")))["cases"]}
BIN = os.path.join(os.path.dirname(vs.__file__), "bin")


def sha(b):
    return hashlib.sha256(b).hexdigest()


def _run(name, tmp_path, header):
    c = CASES[name]
    env = dict(os.environ, VS_SEED=str(c["seed"]), VS_WAV_HEADER=str(header))
    fg = subprocess.run([os.path.join(BIN, "flowgen_shimmer"), "-o", "g.wav"] + c["flowgen_args"],
                        cwd=tmp_path, env=env, capture_output=True)
    assert fg.returncode == 0, fg.stderr
    vw = subprocess.run([os.path.join(BIN, "vowel"), "-i", "g.wav", "-o", "o.wav"] + c["vowel_args"],
                        cwd=tmp_path, env=env, capture_output=True)
    assert vw.returncode == 0, vw.stderr
    return c, fg, vw, open(tmp_path / "g.wav", "rb").read(), open(tmp_path / "o.wav", "rb").read()


@pytest.mark.parametrize("name", ["ka_g16_va", "cfg3_lane0", "edge_dc_kvar", "cfg4_lane2", "edge_dur_frac", "onoise_22k", "onoise_frac"])
@pytest.mark.parametrize("header", [44, 72])
def test_pipeline_files_match_reference(tmp_path, name, header):
    c, fg, vw, g, o = _run(name, tmp_path, header)
    assert len(g) == header + 2 * c["n_samples"] and len(o) == len(g)
    assert sha(g[header:]) == c["sha256_flow"]
    assert sha(o[header:]) == c["sha256_pcm"]
    assert o[:header] == g[:header]                      # vowel copies its input header verbatim
    assert fg.stdout.endswith(b"done\n") and vw.stdout.endswith(b"Wait...done\n")


@pytest.mark.parametrize("name", ["cfg3_lane0", "edge_dc_kvar"])
def test_flowgen_stdout_equals_reference(tmp_path, name):
    c, fg, vw, g, o = _run(name, tmp_path, 72)
    want = open(os.path.join(GOLD, "stdout_%s.txt" % name), "rb").read()
    assert fg.stdout == want                             # banner, per-cycle S / SNRdb lines, "done"


def test_vowel_stdout_banner(tmp_path):
    c, fg, vw, g, o = _run("cfg4_lane2", tmp_path, 44)
    assert vw.stdout.startswith(b"vowel /i/ MNV \nMaurilio N. Vieira, 28 mar 97. \n")
    assert b"pre_emphasis= 1.00, gain=10.00, snr= 0.00\n" in vw.stdout


def test_vs_batch_manifest(tmp_path):
    """vs_batch: N utterances described by the reference's own command lines -> N .wav files,
    one fused launch per distinct sample count; payloads equal the reference's outputs."""
    names = ["cfg3_lane0", "cfg3_lane1", "cfg5_lane3", "cfg4_lane2", "onoise_22k", "edge_dc_kvar", "ka_g16_va"]
    lines = ["# golden cases as a manifest", ""]
    for i, n in enumerate(names):
        c = CASES[n]
        lines.append("seed=%d -o out%d.wav %s | %s" % (c["seed"], i, " ".join(c["flowgen_args"]), " ".join(c["vowel_args"])))
    (tmp_path / "m.txt").write_text("\n".join(lines) + "\n")
    r = subprocess.run([os.path.join(BIN, "vs_batch"), "m.txt"], cwd=tmp_path, capture_output=True,
                       env=dict(os.environ, VS_WAV_HEADER="44"))
    assert r.returncode == 0, r.stderr
    assert b"7 utterances in 2 launch(es)" in r.stdout
    for i, n in enumerate(names):
        raw = open(tmp_path / ("out%d.wav" % i), "rb").read()
        assert len(raw) == 44 + 2 * CASES[n]["n_samples"]
        assert sha(raw[44:]) == CASES[n]["sha256_pcm"], n


def test_vs_batch_refuses_two_lines_with_one_output(tmp_path):
    c = CASES["cfg3_lane0"]
    line = "seed=%d -o same.wav %s | %s" % (c["seed"], " ".join(c["flowgen_args"]), " ".join(c["vowel_args"]))
    (tmp_path / "m.txt").write_text(line + "\n" + line.replace("seed=%d" % c["seed"], "seed=99") + "\n")
    r = subprocess.run([os.path.join(BIN, "vs_batch"), "m.txt"], cwd=tmp_path, capture_output=True)
    assert r.returncode == 1 and b"more than one line" in r.stderr
    assert not os.path.exists(tmp_path / "same.wav")


def test_vs_batch_sharded_over_logical_devices(tmp_path):
    """vs_batch --gpus 4 with VS_DEVICES=0,0,0,0: four logical shards of the one device, each a
    contiguous block of the manifest; the files must not depend on the sharding"""
    names = ["cfg3_lane0", "cfg3_lane1", "cfg3_lane2", "cfg2_lane0", "cfg2_lane1", "ka_g16_va", "cfg5_lane3"]
    names = [n for n in names if n in CASES and CASES[n]["n_samples"] == 16000]
    assert len(names) >= 5
    lines = ["seed=%d -o s%d.wav %s | %s" % (CASES[n]["seed"], i, " ".join(CASES[n]["flowgen_args"]), " ".join(CASES[n]["vowel_args"]))
             for i, n in enumerate(names)]
    (tmp_path / "m.txt").write_text("\n".join(lines) + "\n")
    r = subprocess.run([os.path.join(BIN, "vs_batch"), "--gpus", "4", "m.txt"], cwd=tmp_path, capture_output=True,
                       env=dict(os.environ, VS_WAV_HEADER="44", VS_DEVICES="0,0,0,0"))
    assert r.returncode == 0, r.stderr
    for i, n in enumerate(names):
        raw = open(tmp_path / ("s%d.wav" % i), "rb").read()
        assert sha(raw[44:]) == CASES[n]["sha256_pcm"], n


def test_vs_batch_rejects_what_the_reference_rejects(tmp_path):
    (tmp_path / "m.txt").write_text("-o a.wav -r 22050 | -v a\n")     # explicit 22050
    r = subprocess.run([os.path.join(BIN, "vs_batch"), "m.txt"], cwd=tmp_path, capture_output=True)
    assert r.returncode == 1 and b"usage()" in r.stderr
    (tmp_path / "m.txt").write_text("-o a.wav -r 16000 | -v e\n")     # no 'e' table
    r = subprocess.run([os.path.join(BIN, "vs_batch"), "m.txt"], cwd=tmp_path, capture_output=True)
    assert r.returncode == 1


def test_random_command_lines_equal_the_reference_programs():
    """tools/cli_fuzz.py on a small draw: both drop-in programs against the compiled reference
    programs (oracle/_ref travels to the GPU box) on random command lines -- files byte for byte,
    header included, and the complete stdout of both stages"""
    import sys
    from oracle import pyoracle as po
    if not po.have_reference():
        pytest.skip("oracle/_ref not built")
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import cli_fuzz
    old = sys.argv
    sys.argv = ["cli_fuzz.py", "2026", "10"]
    try:
        assert cli_fuzz.main() == 0
    finally:
        sys.argv = old


def test_vs_bench_runs_from_plain_c():
    """the C-only benchmark program: same workload description as bench.py, one JSON line"""
    for extra in ([], ["--host"], ["--arith", "fma"]):
        r = subprocess.run([os.path.join(BIN, "vs_bench"), "--lanes", "4096", "--steps", "2", "--warmup", "1"] + extra,
                           capture_output=True, timeout=300)
        assert r.returncode == 0, r.stderr
        d = json.loads(r.stdout.decode().strip().splitlines()[-1])
        assert d["utterances"] == 4096 and d["samples_per_utterance"] == 16000 and d["value"] > 100
    # a plan per batch of new utterances, made on a second host thread next to the running kernel, the plans of two batches
    # ago taken down meanwhile (their device blocks go to the plans to come): device time per batch, and the last batch's
    # PCM equals a plain launch of the same utterances
    for lanes, extra in ((4096, []), (70000, ["--arith", "fma"])):
        r = subprocess.run([os.path.join(BIN, "vs_bench"), "--lanes", str(lanes), "--steps", "12", "--warmup", "1", "--fresh"] + extra,
                           capture_output=True, timeout=300)
        assert r.returncode == 0, r.stderr
        d = json.loads(r.stdout.decode().strip().splitlines()[-1])
        assert d["batches"] == 12 and d["utterances_per_batch"] == lanes and d["ms_per_batch"] > 0 and d["value"] > 100
        assert d["last_batch_equals_a_plain_launch"] is True and d["plan_destroy_ms_avg"] < 1.0 and d["plan_create_wall_ms_avg"] > 0
    # the node entry over two logical shards of the one device
    r = subprocess.run([os.path.join(BIN, "vs_bench"), "--lanes", "3000", "--steps", "2", "--warmup", "1", "--gpus", "2"],
                       capture_output=True, timeout=300, env=dict(os.environ, VS_DEVICES="0,0"))
    assert r.returncode == 0, r.stderr
    d = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert d["n_gpus"] == 2 and d["utterances_per_gpu"] == 3000 and d["value"] > 100
    assert d["links"] == ["self", "self"] and d["rccl_comm_ranks"] == [0, 0]   # (peer transport: no communicator)
    # the line says which devices took part (two logical shards = ONE device) and that the gathered PCM is what one
    # device gives alone, row for row
    assert len(d["devices"]) == 2 and d["devices"][0] == d["devices"][1] and d["distinct_devices"] == 1
    assert d["gathered_equals_one_device"] is True and d["rows_verified_against_one_device"] == 6000
    # host code in C + RCCL gather, as far as one device goes: a one-rank communicator owned by the node
    r = subprocess.run([os.path.join(BIN, "vs_bench"), "--lanes", "3000", "--steps", "2", "--warmup", "1", "--gpus", "1", "--rccl"],
                       capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr
    d = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert d["n_gpus"] == 1 and d["value"] > 100 and d["links"] == ["self"]
    assert d["rccl_comm_ranks"] == [1]                      # what RCCL itself says the communicator spans (ncclCommCount)
    assert d["gathered_equals_one_device"] is True and d["rows_verified_against_one_device"] == 3000
    # ... and logical shards of one device are refused, loudly
    r = subprocess.run([os.path.join(BIN, "vs_bench"), "--lanes", "3000", "--steps", "1", "--warmup", "0", "--gpus", "2", "--rccl"],
                       capture_output=True, timeout=300, env=dict(os.environ, VS_DEVICES="0,0"))
    assert r.returncode != 0 and b"RCCL transport" in r.stderr
