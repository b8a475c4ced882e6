"""Differential fuzz of the two argv parsers (+ the CPU oracle behind them) against the COMPILED
reference (oracle/_ref, this container only): random command lines -- valid, out of range,
malformed, upper/lower case flags, words where flags should be -- must be answered alike:
usage() by both, or the same samples by both.  CPU only; skipped where oracle/_ref is absent."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

import voice_synth_amd as vs
from voice_synth_amd import _ffi
from oracle import pyoracle as po

pytestmark = pytest.mark.skipif(not po.have_reference(), reason="oracle/_ref not built")

FLAGS = list("rdjcfgkzsnal") + list("RDJCFGKZSNAL") + ["q", "x", "rate", "dur", "Fg"]
VALUES = {
    "r": ["8000", "11025", "16000", "22050", "44100", "32000", "16000.7"],   # rates <= 0 make the reference run (almost) forever
    "d": ["0.5", "0.49", "0.77", "1", "-1", "x"],
    "j": ["0", "1", "10", "1000", "1001", "-0.1", "3.3"],
    "c": ["0", "0.05", "0.55", "1", "1.01", "-0.2"],
    "f": ["49", "50", "80", "120", "300", "124.99", "125"],
    "g": ["49", "50", "84", "125", "313", "1000"],
    "k": ["0.49", "0.5", "0.65", "1.2", "3"],
    "z": ["0", "0.5", "1", "1.1", "-1"],
    "s": ["0", "5.76", "50", "100", "101", "-3"],
    "n": ["0", "20", "50", "51", "-1", "0.1"],
    "a": ["0", "1", "12000", "32766", "32767", "-4", "12000.9"],
    "l": ["0", "0.1", "0.29", "0.3", "0.31", "-0.1"],
}


def _random_flowgen_argv(rng):
    args = []
    for _ in range(int(rng.integers(0, 6))):
        flag = str(rng.choice(FLAGS))
        key = flag[0].lower()
        vals = VALUES.get(key, ["1", "abc"])
        args.append("-" + flag)
        if rng.random() < 0.95:
            args.append(str(rng.choice(vals)))
    if rng.random() < 0.1:
        args.append(str(rng.choice(["stray", "input", "i"])))
    return args


def _run_ref(name, argv, cwd, seed):
    env = dict(os.environ, VS_SEED=str(seed), VS_DRAWLOG=os.path.join(cwd, "draws.txt"))
    return subprocess.run([os.path.join(po.REF_DIR, name)] + argv, cwd=cwd, env=env, capture_output=True, timeout=60)


def test_flowgen_argv_fuzz_against_the_compiled_reference():
    rng = np.random.default_rng(20261004)
    accepted = rejected = 0
    for case in range(400):
        args = _random_flowgen_argv(rng)
        with_o = rng.random() < 0.93
        pos = int(rng.integers(0, len(args) // 2 + 1)) * 2 if args else 0
        argv = args[:pos] + (["-o", "g.wav"] if with_o else []) + args[pos:]
        rc, cmd = vs.parse_flowgen(argv)
        with tempfile.TemporaryDirectory(prefix="vsfz") as d:
            ref = _run_ref("flowgen_shimmer", argv, d, 77 + case)
            made = os.path.exists(os.path.join(d, "g.wav"))
            ref_usage = b"usage:" in ref.stdout and not made
            if ref.returncode < 0:
                # the reference died on a signal: its buffer x holds 2*fs/Fg samples (fg:569), a
                # period longer than that runs over the heap block.  Undefined there; we must at
                # least not have answered usage()
                assert rc == 0, argv
                continue
            assert ref.returncode == 0, (argv, ref.returncode)
            if rc == _ffi.VS_USAGE:
                assert ref_usage, ("we answer usage(), the reference runs", argv)
                rejected += 1
                continue
            assert rc == 0 and not ref_usage, ("the reference answers usage(), we parse", argv, rc)
            flow, _ = po._payload(os.path.join(d, "g.wav"))
        lane = cmd.lane
        lane.seed = 77 + case
        if vs.load().vs_lane_validate(vs.C.byref(lane)) != 0:
            continue              # undefined behaviour in the reference (rate <= 0, cq = 0 with -n, ...): not compared
        n = vs.num_samples(lane.fs, cmd.dur)
        assert n == len(flow), (argv, n, len(flow))
        if not _reference_defined(lane):
            continue              # heap overrun (fg:569) or the uninitialised T4 (fg:114) in the reference
        want, _, _, _ = po.source_one(lane, n)
        assert np.array_equal(want, flow), argv
        accepted += 1
    assert accepted > 60 and rejected > 60, (accepted, rejected)


VFLAGS = list("pgnv") + list("PGNV") + ["q", "vowel"]
VVALUES = {
    "p": ["0", "0.5", "1", "1.1", "-0.1"],
    "g": ["1", "0.99", "10", "250", "-2"],
    "n": ["0", "-1", "0.1", "20", "60"],
    "v": ["a", "i", "u", "1", "2", "3", "4", "5", "6", "7", "8", "0", "e", "ab", "1x"],
}


def test_vowel_argv_fuzz_against_the_compiled_reference():
    rng = np.random.default_rng(4102)
    fa = ["-r", "16000", "-d", "0.5", "-j", "1", "-s", "5.76", "-n", "20"]
    accepted = rejected = 0
    for case in range(250):
        args = []
        for _ in range(int(rng.integers(0, 5))):
            flag = str(rng.choice(VFLAGS))
            args.append("-" + flag)
            if rng.random() < 0.95:
                args.append(str(rng.choice(VVALUES.get(flag[0].lower(), ["1", "abc"]))))
        if rng.random() < 0.85 and "-v" not in args and "-V" not in args:
            args += ["-v", str(rng.choice(list("aiu1234567")))]
        io = (["-i", "g.wav"] if rng.random() < 0.95 else []) + (["-o", "o.wav"] if rng.random() < 0.95 else [])
        argv = io + args if rng.random() < 0.5 else args + io
        if rng.random() < 0.08:
            argv.append(str(rng.choice(["stray", "in"])))
        rc, cmd = vs.parse_vowel(argv)
        seed = 900 + case
        with tempfile.TemporaryDirectory(prefix="vsfz") as d:
            r0 = _run_ref("flowgen_shimmer", ["-o", "g.wav"] + fa, d, seed)
            assert r0.returncode == 0
            flow, _ = po._payload(os.path.join(d, "g.wav"))
            ref = _run_ref("vowel", argv, d, seed)
            made = os.path.exists(os.path.join(d, "o.wav"))
            if rc == _ffi.VS_USAGE:
                assert not made, ("we answer usage(), the reference filters", argv)
                rejected += 1
                continue
            assert rc == 0
            if cmd.output_arg == -1:
                continue          # no -o: the reference opens argv[-1] (vowel_new.c:212), nothing to compare
            assert made, ("the reference answers usage(), we parse", argv, ref.stdout[:80])
            pcm, _ = po._payload(os.path.join(d, "o.wav"))
        lane, _ = vs.lane_from_cli(fa, ["-v", "a"], seed)
        lane.gain = cmd.gain
        lane.pre_emphasis = cmd.pre_emphasis
        lane.vowel = cmd.vowel
        lane.out_snr = cmd.snr if cmd.noise_arg != -1 else 0.0
        if vs.load().vs_lane_validate(vs.C.byref(lane)) != 0:
            continue              # upper-case A/I/U: accepted by the reference's check, no table loaded (F11)
        want = po.filter([lane], flow[None, :])[0]
        assert np.array_equal(want, pcm), argv
        accepted += 1
    assert accepted > 40 and rejected > 40, (accepted, rejected)


def _reference_defined(lane):
    """the reference's heap block x holds 2*fs/Fg samples (flowgen_shimmer.c:569) and w[] 500
    (fg:115): a period beyond either is an overrun there.  And its T4 is an uninitialised stack
    variable (fg:114) that only a sample below the DC flow assigns: with noise on and a
    DC flow of zero it is read before it is written -- what the compiled reference does then depends
    on what the loader left on the stack (0 in this container, which is what the carry_t4_* goldens
    record and what the engine defines)."""
    tmax = int(1.2 * int(np.float32(lane.fs) / np.float32(lane.F0))) + 2
    if (lane.flags & _ffi.VS_FLAG_NOISE) and lane.DC <= 0:
        return False
    return tmax <= int(lane.fs / lane.Fg * 2) and tmax <= 500


@pytest.mark.parametrize("draw,dur", [("uniform", "0.5"), ("corners", "0.5"), ("uniform", "0.83")])
def test_oracle_against_the_compiled_reference_over_the_option_fuzz(draw, dur):
    """the generators of the GPU fuzz (tests/test_gpu_properties.py) feed the GPU-vs-oracle tests;
    here the SAME draws pin the oracle to the compiled reference: flow and speech, byte for byte"""
    from test_gpu_properties import _corner_lanes, _fuzz_lanes
    specs = []
    if draw == "uniform":
        lanes = _fuzz_lanes(97531 + int(float(dur) * 100), 400, dur=dur, specs=specs)
    else:
        lanes = _corner_lanes(97531, 400, specs=specs)
    compared = 0
    for lane, (fa, va, seed) in zip(lanes, specs):
        if not _reference_defined(lane):
            continue
        ref = po.run_reference(fa, va, seed)
        n = len(ref["flow"])
        assert n == vs.num_samples(lane.fs, float(dur))
        flow, _, _, nd = po.source_one(lane, n)
        assert np.array_equal(flow, ref["flow"]), (fa, seed)
        assert nd == ref["ndraws"], (fa, seed)
        assert np.array_equal(po.filter([lane], flow[None, :])[0], ref["pcm"]), (fa, va, seed)
        compared += 1
    assert compared > 100, compared


@pytest.mark.parametrize("prog,argv", [
    # flowgen_shimmer only notes WHERE a value stands and converts the last one: an early bad value is never seen
    ("flowgen_shimmer", ["-o", "g.wav", "-r", "16000", "-d", "0.1", "-d", "0.5"]),
    ("flowgen_shimmer", ["-o", "g.wav", "-r", "16000", "-d", "0.5", "-d", "0.1"]),
    ("flowgen_shimmer", ["-o", "g.wav", "-r", "16000", "-d", "0.5", "-f", "200", "-g", "250"]),   # F0 against the Fg of the same line
    ("flowgen_shimmer", ["-o", "g.wav", "-r", "16000", "-d", "0.5", "-g", "250", "-f", "250"]),
    ("flowgen_shimmer", ["-o", "g.wav", "-r", "16000", "-d", "0.5", "-l", "0.3"]),                 # 0.3f > 0.3
    ("flowgen_shimmer", ["-o", "g.wav", "-r", "16000", "-d", "0.5", "-n", "20", "-l", "0.1", "-a", "1000"]),
    # vowel converts every value where it stands
    ("vowel", ["-i", "g.wav", "-o", "o.wav", "-v", "a", "-g", "0", "-g", "5"]),
    ("vowel", ["-i", "g.wav", "-o", "o.wav", "-v", "a", "-g", "5", "-g", "2"]),
    ("vowel", ["-i", "g.wav", "-o", "o.wav", "-v", "8", "-v", "a"]),
    ("vowel", ["-i", "g.wav", "-o", "o.wav", "-v", "a", "-n", "0"]),
    ("vowel", ["-i", "g.wav", "-o", "o.wav", "-v", "a", "-p", "nan"]),
    # values that overflow a float: the reference tests only the LOWER bound of -g / -n / -k (vowel_new.c:132, 142;
    # flowgen_shimmer.c:484) and lets them pass
    ("vowel", ["-i", "g.wav", "-o", "o.wav", "-v", "a", "-g", "1e39"]),
    ("vowel", ["-i", "g.wav", "-o", "o.wav", "-v", "a", "-g", "inf"]),
    ("vowel", ["-i", "g.wav", "-o", "o.wav", "-v", "a", "-n", "inf"]),
    ("vowel", ["-i", "g.wav", "-o", "o.wav", "-v", "a", "-n", "1e39"]),
    ("vowel", ["-i", "g.wav", "-o", "o.wav", "-v", "a", "-p", "inf"]),
    ("flowgen_shimmer", ["-o", "g.wav", "-r", "16000", "-d", "0.5", "-k", "1e39"]),
    ("flowgen_shimmer", ["-o", "g.wav", "-r", "16000", "-d", "0.5", "-k", "inf"]),
    ("flowgen_shimmer", ["-o", "g.wav", "-r", "16000", "-d", "0.5", "-j", "inf"]),
])
def test_repeated_and_order_dependent_options_as_the_reference_takes_them(prog, argv):
    with tempfile.TemporaryDirectory(prefix="vsfz") as d:
        if prog == "vowel":
            assert _run_ref("flowgen_shimmer", ["-o", "g.wav", "-r", "16000", "-d", "0.5"], d, 5).returncode == 0
        out = "g.wav" if prog == "flowgen_shimmer" else "o.wav"
        if prog == "flowgen_shimmer" and os.path.exists(os.path.join(d, out)):
            os.remove(os.path.join(d, out))
        ref = _run_ref(prog, argv, d, 5)
        ref_ran = os.path.exists(os.path.join(d, out)) and b"usage:" not in ref.stdout
    rc, cmd = (vs.parse_flowgen if prog == "flowgen_shimmer" else vs.parse_vowel)(argv)
    assert (rc == 0) == ref_ran, (argv, rc, ref.stdout[:60])
    if rc == 0 and prog == "flowgen_shimmer":
        assert cmd.dur == np.float32(argv[argv.index("-d", argv.index("-d") + 1) + 1] if argv.count("-d") > 1 else 0.5)


def test_unbounded_options_accept_infinity_like_the_reference():
    """-d and -g (Fg) of flowgen_shimmer have no upper bound either (flowgen_shimmer.c:472, 496: only f < 0.5 / f < 50
    answer usage()), but the reference cannot be RUN on them: an infinite duration never ends, an infinite Fg sizes the
    sample buffer to zero bytes (fg:569).  The parser must accept what the reference's parser accepts; the engine
    refuses the lane afterwards (VS_ERR_UNSUPPORTED / VS_ERR_RANGE from validation), it does not call it usage()."""
    for argv in (["-o", "g.wav", "-d", "inf"], ["-o", "g.wav", "-d", "1e39"], ["-o", "g.wav", "-g", "1e39"], ["-o", "g.wav", "-g", "inf"]):
        rc, cmd = vs.parse_flowgen(argv)
        assert rc == 0, argv
    rc, cmd = vs.parse_flowgen(["-o", "g.wav", "-d", "inf"])
    assert np.isinf(cmd.dur)
    rc, cmd = vs.parse_flowgen(["-o", "g.wav", "-g", "1e39"])
    assert np.isinf(cmd.lane.Fg)
