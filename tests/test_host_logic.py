"""Host-side C of the product: the two argv parsers, validation, sample count, RIFF header,
lane expansion (no device needed).  Mirrors flowgen_shimmer.c:128-222, 463-565 and
vowel_new.c:116-192; the quirks are the ones SURVEY.md section 0 records."""
import ctypes as C
import math
import os
import struct
import subprocess

import numpy as np
import pytest

import voice_synth_amd as vs
from voice_synth_amd import _ffi
from oracle import pyoracle as po

F = np.float32


def fg(*args):
    return vs.parse_flowgen(["-o", "f.wav"] + list(args))


def test_defaults_match_reference_initialiser():
    rc, c = fg()
    assert rc == 0
    l = c.lane
    assert (l.jitter, l.DC, l.noise, l.Kvar, l.shimmer, l.flags) == (0, 0, 0, 0, 0, 0)
    assert F(l.cq) == F(0.55) and F(l.K) == F(0.65) and l.Fg == 125 and l.F0 == 120
    assert l.fs == 22050 and l.amp == 12000 and c.dur == 1.0
    assert l.gain == 10.0 and l.pre_emphasis == 1.0 and l.vowel == ord("a")


def test_usage_conditions():
    assert vs.parse_flowgen([])[0] == _ffi.VS_USAGE               # argc < 2
    assert vs.parse_flowgen(["-r", "16000"])[0] == _ffi.VS_USAGE  # no -o
    assert vs.parse_flowgen(["-o"])[0] == _ffi.VS_USAGE           # flag without value
    assert fg("-q", "1")[0] == _ffi.VS_USAGE                      # unknown flag
    assert fg("stray")[0] == _ffi.VS_USAGE                        # trailing word not starting with 'i'
    assert fg("ignored")[0] == 0                                  # ... but 'i...' is let through (fg:219)
    assert fg("-D", "2", "-R", "8000")[0] == 0                    # flags are case-insensitive


def test_ranges():
    assert fg("-d", "0.49")[0] == _ffi.VS_USAGE and fg("-d", "0.5")[0] == 0
    assert fg("-j", "1001")[0] == _ffi.VS_USAGE and fg("-j", "1000")[0] == 0  # checked after /100
    assert fg("-k", "0.49")[0] == _ffi.VS_USAGE
    assert fg("-c", "1.01")[0] == _ffi.VS_USAGE and fg("-c", "0")[0] == 0
    assert fg("-g", "49")[0] == _ffi.VS_USAGE
    assert fg("-f", "125")[0] == _ffi.VS_USAGE                   # F0 < Fg (default 125)
    assert fg("-f", "300", "-g", "313")[0] == 0
    assert fg("-f", "49", "-g", "60")[0] == _ffi.VS_USAGE
    assert fg("-n", "51")[0] == _ffi.VS_USAGE and fg("-n", "-1")[0] == _ffi.VS_USAGE
    assert fg("-a", "32767")[0] == _ffi.VS_USAGE and fg("-a", "32766")[0] == 0
    assert fg("-l", "0.31")[0] == _ffi.VS_USAGE
    assert fg("-l", "0.3")[0] == _ffi.VS_USAGE                   # (float)0.3 > 0.3
    assert fg("-z", "1.1")[0] == _ffi.VS_USAGE
    assert fg("-s", "101")[0] == _ffi.VS_USAGE


def test_quirk_explicit_22050_rejected():
    assert fg("-r", "22050")[0] == _ffi.VS_USAGE
    for r in ("44100", "11025", "16000", "8000"):
        rc, c = fg("-r", r)
        assert rc == 0 and c.lane.fs == int(r)


def test_quirk_noise_sets_dc():
    rc, c = fg("-n", "20")
    assert rc == 0 and c.lane.DC == 0.25 and c.lane.flags & vs.VS_FLAG_NOISE
    assert F(c.lane.noise) == F(math.pow(10, float(F(20.0) / F(10))))
    rc, c = fg("-n", "20", "-l", "0.1", "-a", "10000")   # -l is applied later and wins
    assert rc == 0 and F(c.lane.DC) == F(F(0.1) * F(10000))


def test_unit_conversions():
    rc, c = fg("-j", "1", "-s", "5.76")
    assert F(c.lane.jitter) == F(1.0 / 100.0)
    assert F(c.lane.shimmer) == F(F(5.76) / F(100))
    assert c.lane.flags == vs.VS_FLAG_JITTER | vs.VS_FLAG_SHIMMER
    rc, c = fg("-j", "0")                          # given but zero: flag set, generator stays off
    assert c.lane.flags == vs.VS_FLAG_JITTER and c.lane.jitter == 0


def test_vowel_parser():
    ok = ["-i", "a.wav", "-o", "b.wav"]
    rc, c = vs.parse_vowel(ok + ["-v", "4", "-g", "2.5", "-p", "0.25"])
    assert rc == 0 and c.vowel == ord("4") and c.gain == 2.5 and c.pre_emphasis == 0.25
    assert vs.parse_vowel(ok)[0] == _ffi.VS_USAGE                       # -v required
    assert vs.parse_vowel(["-o", "b.wav", "-v", "a"])[0] == _ffi.VS_USAGE  # -i required
    assert vs.parse_vowel(ok + ["-v", "e"])[0] == _ffi.VS_USAGE         # no 'e' entry (F11)
    assert vs.parse_vowel(ok + ["-v", "a", "-g", "0.9"])[0] == _ffi.VS_USAGE
    assert vs.parse_vowel(ok + ["-v", "a", "-p", "1.5"])[0] == _ffi.VS_USAGE
    assert vs.parse_vowel(ok + ["-v", "a", "-n", "0"])[0] == _ffi.VS_USAGE
    rc, c = vs.parse_vowel(ok + ["-v", "a", "-n", "20"])
    assert rc == 0 and c.noise_arg > 0 and F(c.snr) == F(100.0)
    rc, c = vs.parse_vowel(ok + ["-v", "A"])                            # passes the parser ...
    assert rc == 0
    lane = vs.default_lane()
    lane.vowel = ord("A")                                               # ... but loads nothing (F11)
    assert vs.load().vs_lane_validate(C.byref(lane)) == _ffi.VS_ERR_UNSUPPORTED


def test_num_samples_is_a_float_product():         # flowgen_shimmer.c:242
    assert vs.num_samples(16000, 1.0) == 16000
    assert vs.num_samples(22050, 2.0) == 44100
    assert vs.num_samples(11025, 0.77) == int(F(11025) * F(0.77))
    assert vs.num_samples(44100, 600.0) == int(F(44100) * F(600.0))


def test_validate():
    v = vs.load().vs_lane_validate
    lane = vs.default_lane()
    assert v(C.byref(lane)) == 0
    lane.cq = 0.0                                   # no pulse: x_pow = 0/0 in the reference
    assert v(C.byref(lane)) == _ffi.VS_ERR_UNSUPPORTED
    lane = vs.default_lane(); lane.F0 = 130
    assert v(C.byref(lane)) == _ffi.VS_ERR_RANGE
    lane = vs.default_lane(); lane.gain = 0.5
    assert v(C.byref(lane)) == _ffi.VS_ERR_RANGE
    lane = vs.default_lane(); lane.vowel = 0; lane.A[0] = 1.0
    assert v(C.byref(lane)) == 0
    lane.A[5] = float("nan")
    assert v(C.byref(lane)) == _ffi.VS_ERR_RANGE


def _hdr(nbytes, fs, dur):
    buf = (C.c_ubyte * 72)()
    n = vs.load().vs_wav_header_write(buf, nbytes, fs, dur)
    return bytes(buf[:n])


def test_wav_header_44():
    h = _hdr(44, 16000, 1.0)
    assert len(h) == 44
    riff, size, wave, fmt, fmtsize, tag, ch, sps, avg, align, bits, data, dsize = struct.unpack(
        "<4sI4s4sIHHIIHH4sI", h)
    assert (riff, wave, fmt, data) == (b"RIFF", b"WAVE", b"fmt ", b"data")
    assert (size, fmtsize, tag, ch, sps, avg, align, bits, dsize) == (32036, 16, 1, 1, 16000, 32000, 2, 16, 32000)


@pytest.mark.skipif(not po.have_reference(), reason="oracle/_ref not built")
def test_wav_header_72_equals_reference_bytes():
    for fa, fs, dur in ((["-r", "16000", "-d", "1"], 16000, 1.0), (["-d", "2"], 22050, 2.0),
                        (["-r", "11025", "-d", "0.77"], 11025, 0.77)):
        ref = po.run_reference(fa, None, 0)
        assert _hdr(72, fs, dur) == ref["flow_header"]


def test_wav_header_read_both_layouts():
    rd = vs.load().vs_wav_header_read
    for nbytes in (44, 72):
        h = _hdr(nbytes, 22050, 2.0)
        fs, tag, bits, db = C.c_int32(), C.c_int(), C.c_int(), C.c_uint64()
        buf = (C.c_ubyte * 72)(*h)
        assert rd(buf, len(h), C.byref(fs), C.byref(tag), C.byref(bits), C.byref(db)) == nbytes
        assert (fs.value, tag.value, bits.value, db.value) == (22050, 1, 16, 88200)
    assert rd((C.c_ubyte * 72)(), 72, None, None, None, None) == _ffi.VS_ERR_IO


def test_expand_lane_matches_reference_expressions():
    lane, dur = vs.lane_from_cli(["-r", "16000", "-d", "1", "-j", "1", "-s", "5.76", "-n", "20"], ["-v", "2"], 7)
    d = _ffi.DevLane()
    assert vs.load().vs_expand_lane(C.byref(lane), 5, C.byref(d)) == 0
    assert d.P == 133 and d.T2 == 37 and d.row == 5       # (int)(16000/120), ceil(.5*.55*133)
    assert F(d.t_hi) == F(1.2) * F(133) and F(d.t_lo) == F(0.8) * F(133)
    assert F(d.a_hi) == F(1.8) * F(12000) and F(d.a_lo) == F(0.2) * F(12000)
    # jitter | shimmer | noise | VS_DF_FAST: amplitude <= 1.8*12000 fits a short, (2*0.65-1)*21600 too,
    # and the open phase (2*37 samples) ends 8 samples before the shortest period ceil(0.8*133) = 107
    assert d.tbound == 159 and d.dcs == 0 and d.flags == 7 | _ffi.VS_DF_FAST
    assert d.thr == 1                                       # ceil(0.25): x < 0.25  <=>  x < 1
    assert (d.key0, d.key1) == (7, 0)
    assert d.tap_row == vs.load().vs_vowel_index(ord("2")) == 4   # a i u 1 2 ...: the plan's tap table row
    row = (C.c_double * d.T2)()
    vs.load().vs_cos_row(d.T2, row)
    assert list(row) == [math.cos(4.0 * math.atan(1.0) * k / d.T2) for k in range(d.T2)]


def test_fast_flag_bounds():
    """VS_DF_FAST is set only where the short generator sequences are proved equal to the general
    ones: samples fit a short before the cast, T2 >= 4, open phase ends 8 samples before min T"""
    def flags(fa):
        lane, _ = vs.lane_from_cli(fa, ["-v", "a"], 1)
        d = _ffi.DevLane()
        assert vs.load().vs_expand_lane(C.byref(lane), 0, C.byref(d)) == 0
        return d.flags
    assert flags(["-r", "16000"]) & _ffi.VS_DF_FAST                              # amp 12000, no shimmer
    assert flags(["-r", "16000", "-a", "20000", "-s", "5"]) & _ffi.VS_DF_FAST == 0   # 1.8*20000 > 32767
    assert flags(["-r", "16000", "-a", "20000"]) & _ffi.VS_DF_FAST                # without shimmer 20000 fits
    assert flags(["-r", "16000", "-k", "1.2", "-z", "1", "-a", "30000"]) & _ffi.VS_DF_FAST == 0  # (2*2.4-1)*30000
    assert flags(["-r", "16000", "-c", "1.0", "-j", "5"]) & _ffi.VS_DF_FAST == 0  # 2*T2 = P+1 > 0.8*P
    assert flags(["-r", "8000", "-f", "390", "-g", "400", "-c", "0.2"]) & _ffi.VS_DF_FAST == 0  # T2 = 2


def test_ring_slots():
    s = C.c_int()
    assert vs.load().vs_ring_slots_for(159, C.byref(s)) == 0 and s.value >= 24 + 159 + 8 and s.value % 24 == 0
    assert vs.load().vs_ring_slots_for(5000, C.byref(s)) == _ffi.VS_ERR_UNSUPPORTED


def test_cli_usage_text_and_exit_code():
    exe = os.path.join(os.path.dirname(vs.__file__), "bin", "flowgen_shimmer")
    r = subprocess.run([exe], capture_output=True)
    assert r.returncode == 0 and b"usage:" in r.stdout and b"-o x {Output file" in r.stdout
    exe = os.path.join(os.path.dirname(vs.__file__), "bin", "vowel")
    r = subprocess.run([exe], capture_output=True)
    assert r.returncode == 0 and r.stdout.startswith(b"lxfilter -i file1.wav")


@pytest.mark.skipif(not po.have_reference(), reason="oracle/_ref not built")
def test_cli_usage_text_equals_reference():
    for name in ("flowgen_shimmer", "vowel"):
        ours = subprocess.run([os.path.join(os.path.dirname(vs.__file__), "bin", name)], capture_output=True)
        ref = subprocess.run([os.path.join(po.REF_DIR, name)], capture_output=True)
        assert ours.stdout == ref.stdout and ours.returncode == ref.returncode == 0


def test_row_pitch_rule():
    """vs_row_pitch: >= n_samples, a 16-byte multiple; rows of 2 KiB and more a whole number of 128-byte lines, that number
    3 mod 4 (profiles/r05_row_pitch.txt: rows 256-byte multiples or powers of two apart are the slow ones)"""
    assert vs.row_pitch(16000) == 16064 and vs.row_pitch(44100) == 44224 and vs.row_pitch(16384) == 16576
    assert vs.row_pitch(1) == 8 and vs.row_pitch(1000) == 1000 and vs.row_pitch(1023) == 1024 and vs.row_pitch(1024) == 1216
    for n in list(range(1, 5000, 7)) + [2 ** k + d for k in range(10, 31) for d in (-1, 0, 1)]:
        p = vs.row_pitch(n)
        assert p >= n and p % 8 == 0 and p - n < 4 * 64 + 64
        if 2 * n >= 2048:
            assert (2 * p) % 128 == 0 and (2 * p // 128) % 4 == 3
    assert vs.row_pitch(2 ** 63) == 2 ** 63   # nothing to round with: the caller's rows as they are


def test_expand_lane_pads_a_low_order_set_with_zeros():
    """an explicit set of fewer than 22 taps rides the order-22 kernels: taps behind its order are zeros in its row of
    the plan's tap table (acc - 0*y == acc); rows 0..9 of that table are the reference's ten tables (vw:450-544)"""
    from voice_synth_amd import configs
    lib = vs.load()
    lanes, fs, dur = configs.wide_order_lanes([5, 22, 40])
    recs = (_ffi.DevLane * 3)()
    for i in range(3):
        assert lib.vs_expand_lane(C.byref(lanes[i]), i, C.byref(recs[i])) == 0
        assert recs[i].tap_row == -1 and recs[i].row == i
    taps, rows = C.POINTER(C.c_double)(), C.c_size_t()
    assert lib.vs_tap_table_build(lanes, recs, 3, 3, C.byref(taps), C.byref(rows)) == 0
    assert rows.value == 13 and [recs[i].tap_row for i in range(3)] == [10, 11, 12]
    row = lambda r: [taps[r * 22 + j] for j in range(22)]
    assert row(10)[:5] == [lanes[0].A[j + 1] for j in range(5)] and all(x == 0.0 for x in row(10)[5:])
    assert row(11) == [lanes[1].A[j + 1] for j in range(22)]
    # a wide set expands too (its first 22 taps; the plan carries all 40 separately)
    assert row(12) == [lanes[2].A[j + 1] for j in range(22)]
    A = (C.c_double * 41)()
    for t in range(10):
        v = lib.vs_vowel_by_index(t)
        assert lib.vs_vowel_index(v) == t and lib.vs_vowel_coefficients(v, A) == 0
        assert row(t) == [A[j + 1] for j in range(22)]
    assert lib.vs_vowel_index(ord("A")) == -1 and lib.vs_vowel_by_index(10) == 0
    C.CDLL(None).free(taps)
    d = _ffi.DevLane()
    lanes[2].order = 41
    assert lib.vs_expand_lane(C.byref(lanes[2]), 2, C.byref(d)) == _ffi.VS_ERR_RANGE
    lane, _ = vs.lane_from_cli(["-r", "16000", "-d", "1"], ["-v", "a"], 0)
    assert lib.vs_expand_lane(C.byref(lane), 0, C.byref(d)) == 0 and d.tap_row == 0


def test_integration_md_snippets_compile(tmp_path):
    """Every ```c block of INTEGRATION.md is compiled against include/voice_synth.h (gcc -fsyntax-only -Wall -Werror):
    the binding a maintainer of the reference reads must be the binding the header exports (seams it replaces:
    flowgen_shimmer.c:413-421, vowel_new.c:327).  Blocks marked <!-- c:file-scope --> go in at file scope, all others
    become the body of a function; the blocks of section 2 are written against the reference's own globals, which a
    few declarations stand for here (names and types as the reference declares them: struct PAR flowgen_shimmer.c:73-87,
    struct ARGS fg:90-102, vowel_new.c:45-81)."""
    import re
    import shutil

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not shutil.which("gcc"):
        pytest.skip("no gcc on this box")
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    blocks = [(m.group(1) is not None, m.group(2)) for m in re.finditer(r"(<!-- c:file-scope -->\n)?```c\n(.*?)```", text, flags=re.S)]
    assert len(blocks) >= 4
    ref_globals = """
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <time.h>
#include "voice_synth.h"
/* stand-ins for the reference's globals (section 2) */
static struct { float dur, jitter, cq, K, Fg, F0, DC, noise; long fs; int amp; float Kvar, Shimmer; } par;
static struct { int jitter, Shimmer, noise; } arg;
static struct { unsigned long nSamplesPerSec; } header;
static FILE *outfile;
static float gain, pre_emphasis, snr;
static int noise_arg, Order;
static char alg;
static double A[41];
static signed short *x_all, *y_all;
static size_t n;
"""
    src = [ref_globals]
    nbody = 0
    for file_scope, body in blocks:
        if file_scope:
            src.append(body)
        else:
            # section 2's vowel block uses a context made earlier in that main(); give every body one
            needs_ctx = "vs_ctx *ctx" not in body
            src.append("void snippet_%d(void)\n{\n%s%s\n}\n" % (nbody, "  vs_ctx *ctx = NULL;\n" if needs_ctx else "", body))
            nbody += 1
    path = tmp_path / "integration_snippets.c"
    path.write_text("\n".join(src))
    r = subprocess.run(["gcc", "-std=gnu11", "-fsyntax-only", "-Wall", "-Werror", "-Wno-unused-variable", "-Wno-unused-function",
                        "-Wno-unused-but-set-variable", "-I", os.path.join(root, "include"), str(path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_profile_provenance_nulls_counters_of_another_tree(tmp_path):
    """bench.py copies HBM traffic and SQ counters out of profiles/pmc_*.json (a --pmc pass cannot share a run with the
    timed region): each record names the tree it was taken on (kernel_sources_sha16 + profile_head) and bench.py keeps a
    figure only when that is the tree it runs from"""
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    from tools import provenance

    tree = provenance.kernel_sources_sha16()
    assert len(tree) == 16 and tree == provenance.kernel_sources_sha16()
    same = {"hbm_bytes_per_launch": 1.0, "kernel_sources_sha16": tree, "profile_head": "abc1234"}
    other = {"hbm_bytes_per_launch": 1.0, "kernel_sources_sha16": "0" * 16, "profile_head": "0ld0ld0"}
    old_style = {"hbm_bytes_per_launch": 1.0}
    p = bench._profile_provenance(same, other)
    assert p["tree_kernel_sources_sha16"] == tree
    assert p["traffic"]["matches_tree"] is True and p["traffic"]["profile_head"] == "abc1234"
    assert p["valu"]["matches_tree"] is False
    assert bench._profile_provenance(old_style, None)["traffic"]["matches_tree"] is False
    assert bench._profile_provenance(None, None)["valu"] is None
    # the content hash follows the kernel sources: one byte more in a copy of the tree changes it
    import shutil
    for d in provenance.SOURCE_DIRS:
        shutil.copytree(os.path.join(root, d), tmp_path / d)
    shutil.copy(os.path.join(root, "Makefile"), tmp_path / "Makefile")      # HIPFLAGS are part of the hash
    assert provenance.kernel_sources_sha16(str(tmp_path)) == tree
    # a comment or white space more does not orphan the PMC passes ...
    with open(tmp_path / "voice_synth_amd" / "csrc" / "vs_device.h", "a") as f:
        f.write("\n/* a remark */   // and another\n")
    assert provenance.kernel_sources_sha16(str(tmp_path)) == tree
    assert provenance.all_sources_sha16(str(tmp_path)) != provenance.all_sources_sha16()
    # ... a token more in the device code, in the plan policy or in the compiler flags does
    with open(tmp_path / "voice_synth_amd" / "csrc" / "vs_device.h", "a") as f:
        f.write("#define VS_ONE_MORE 1\n")
    assert provenance.kernel_sources_sha16(str(tmp_path)) != tree
    shutil.copy(os.path.join(root, "voice_synth_amd", "csrc", "vs_device.h"), tmp_path / "voice_synth_amd" / "csrc" / "vs_device.h")
    assert provenance.kernel_sources_sha16(str(tmp_path)) == tree
    mk = open(tmp_path / "Makefile").read().replace("HIPFLAGS := -O3", "HIPFLAGS := -O2")
    open(tmp_path / "Makefile", "w").write(mk)
    assert provenance.kernel_sources_sha16(str(tmp_path)) != tree
    # a string literal that looks like a comment is left alone
    assert provenance.strip_comments('a = "/* not */ // one";  /* gone */ b') == 'a = "/* not */ // one"; b'


def test_gather_bookkeeping_ragged_cuts_and_more_shards_than_chunks():
    """The rounds of vs_node_synth_gather (csrc/vs_host.c: vs_shard_cut, vs_gather_rounds, vs_gather_round -- plain C,
    walked by the sending shards and by the receiving root alike): every row of every shard travels exactly once, in
    order, in chunks of at most `chunk` rows; shards with fewer chunks than there are rounds (or with no lanes at all)
    sit the later rounds out; the cut equals voice_synth_amd.configs.shard_range."""
    from voice_synth_amd.configs import shard_range

    lib = vs.load()
    lib.vs_gather_rounds.restype = C.c_size_t
    lib.vs_gather_rounds.argtypes = [C.c_size_t, C.c_int, C.c_size_t]
    lib.vs_gather_round.argtypes = [C.c_size_t, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    lib.vs_shard_cut.argtypes = [C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    cases = [(262144, 8, 16384), (65536, 8, 16384), (100000, 8, 16384), (13, 3, 2), (5, 8, 16384), (1, 4, 1), (16385, 2, 16384),
             (3 * 16384 + 1, 3, 16384), (7, 7, 3), (1000, 3, 16384), (2 ** 33 + 5, 8, 16384)]
    for n, shards, chunk in cases:
        rounds = lib.vs_gather_rounds(n, shards, chunk)
        seen_total = 0
        for s in range(shards):
            lo, hi = C.c_size_t(), C.c_size_t()
            assert lib.vs_shard_cut(n, shards, s, C.byref(lo), C.byref(hi)) == 0
            assert (lo.value, hi.value) == tuple(shard_range(n, s, shards))
            own_chunks = (hi.value - lo.value + chunk - 1) // chunk
            assert own_chunks <= rounds
            nxt = lo.value
            probe = range(rounds + 2) if rounds < 64 else list(range(3)) + list(range(rounds - 2, rounds + 2))
            for k in probe:
                r0, rows = C.c_size_t(), C.c_size_t()
                assert lib.vs_gather_round(n, shards, s, chunk, k, C.byref(r0), C.byref(rows)) == 0
                if k < own_chunks:
                    assert r0.value == lo.value + k * chunk and 1 <= rows.value <= chunk and r0.value + rows.value <= hi.value
                    if rounds < 64:
                        assert r0.value == nxt
                        nxt += rows.value
                else:
                    assert rows.value == 0
            if rounds < 64:
                assert nxt == hi.value
            seen_total += hi.value - lo.value
        assert seen_total == n
        assert rounds == (shard_range(n, 0, shards)[1] + chunk - 1) // chunk
    r0, rows = C.c_size_t(), C.c_size_t()
    assert lib.vs_gather_round(10, 2, 2, 4, 0, C.byref(r0), C.byref(rows)) == _ffi.VS_ERR_ARG      # no such shard
    assert lib.vs_gather_round(10, 2, 0, 0, 0, C.byref(r0), C.byref(rows)) == _ffi.VS_ERR_ARG      # no chunk size
    assert lib.vs_gather_round(10, 2, 0, 4, 2 ** 62, C.byref(r0), C.byref(rows)) == 0 and rows.value == 0   # no wrap


def test_frame_length_multiplier_divides_every_sample_index():
    """VsDevLane.lframe_magic (csrc/vs_planhost.c: vs_lframe_magic): the output-noise kernel finds a sample's frame as
    umulhi(i, magic) >> (ceil(log2(Lframe)) - 1); that must be floor(i / Lframe) for every index a row can have
    (0 <= i < 2^31) -- checked around every multiple of the frame length near both ends of the range and on a random
    sample, for the frame lengths of every rate the vowel stage can meet (50 * an even number, vowel_new.c:361-363)"""
    lib = vs.load()
    lib.vs_lframe_magic.restype = C.c_uint32
    lib.vs_lframe_magic.argtypes = [C.c_int]
    rng = np.random.default_rng(5)
    lframes = [100, 200, 400, 500, 800, 1100, 2200, 2400, 4800, 9600, 12800, 32700, 100 * 3277, 50 * 2 * 10 ** 6]
    lframes += [int(x) * 100 for x in rng.integers(1, 20000, size=40)]
    for d in lframes:
        m = lib.vs_lframe_magic(d)
        assert 2 ** 31 < m < 2 ** 32
        sh = (d - 1).bit_length() - 1
        top = (2 ** 31 - 1) // d
        ks = np.unique(np.concatenate([np.arange(0, min(top, 300) + 1), np.arange(max(0, top - 300), top + 1), rng.integers(0, top + 1, size=2000)]))
        idx = np.unique(np.concatenate([ks * d + o for o in (-2, -1, 0, 1, 2, d // 2)]))
        idx = idx[(idx >= 0) & (idx < 2 ** 31)].astype(np.uint64)
        idx = np.concatenate([idx, rng.integers(0, 2 ** 31, size=20000).astype(np.uint64), np.array([2 ** 31 - 1], dtype=np.uint64)])
        got = ((idx * np.uint64(m)) >> np.uint64(32)) >> np.uint64(sh)
        assert np.array_equal(got, idx // np.uint64(d)), d
    assert lib.vs_lframe_magic(0) == 0 and lib.vs_lframe_magic(1) == 0
    lane, dur = vs.lane_from_cli(["-d", "1"], ["-v", "a", "-n", "20"], 1)
    rec = _ffi.DevLane()
    assert lib.vs_expand_lane(C.byref(lane), 0, C.byref(rec)) == 0
    assert rec.Lframe == 1100 and rec.lframe_magic == lib.vs_lframe_magic(1100)
