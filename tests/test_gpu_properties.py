"""Full-size properties on the GPU (BASELINE.json sizes, where the CPU oracle would take
minutes): placement independence, idempotence, sub-batch equivalence, and a checksum of
randomly chosen lanes against the oracle."""
import ctypes as C
import hashlib

import numpy as np
import pytest

import voice_synth_amd as vs
from voice_synth_amd import configs
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


def _full(engine, index, n_lanes):
    specs, fs, dur, _ = configs.config_specs(index, n_lanes)
    lanes, d = vs.lanes_from_specs(specs)
    n = vs.num_samples(fs, d)
    return lanes, n, engine.synth(lanes, n)


@pytest.fixture(scope="module")
def config3(engine):
    return _full(engine, 3, 65536)


def test_config3_full_batch_sampled_against_oracle(engine, config3):
    lanes, n, pcm = config3
    rng = np.random.default_rng(20261003)
    pick = sorted(set(rng.integers(0, 65536, size=192).tolist()) | {0, 63, 64, 65535})
    want = po.synth([lanes[i] for i in pick], n)
    assert np.array_equal(pcm[pick], want)


def test_config3_is_idempotent(engine, config3):
    lanes, n, pcm = config3
    again = engine.synth(lanes, n)
    assert hashlib.sha256(again.tobytes()).digest() == hashlib.sha256(pcm.tobytes()).digest()


def test_config3_placement_independence(engine, config3):
    """a lane's output does not depend on which wavefront / lane slot / batch it is computed in:
    8 shards of 8192 lanes (what 8 GPUs would each compute) reproduce the one-shot result"""
    lanes, n, pcm = config3
    for shard in (0, 3, 7):
        lo = shard * 8192
        sub = (vs.Lane * 8192)()
        C.memmove(sub, C.byref(lanes, lo * C.sizeof(vs.Lane)), 8192 * C.sizeof(vs.Lane))
        got = engine.synth(sub, n)
        assert np.array_equal(got, pcm[lo:lo + 8192]), shard
    # ... and ragged sub-batches that do not fill a wavefront
    for lo, cnt in ((5, 1), (100, 63), (777, 65), (4000, 130)):
        got = engine.synth([lanes[i] for i in range(lo, lo + cnt)], n)
        assert np.array_equal(got, pcm[lo:lo + cnt]), (lo, cnt)


def test_config5_f0_sweep_sampled_against_oracle(engine):
    """full BASELINE config 5 (65536 utterances, per-utterance F0 80-300 Hz, random table + gain):
    the plan sorts the lanes by period before cutting wavefronts; rows stay where they belong"""
    lanes, n, pcm = _full(engine, 5, 65536)
    pick = list(range(0, 65536, 211))
    want = po.synth([lanes[i] for i in pick], n)
    assert np.array_equal(pcm[pick], want)


def test_config4_shard_sampled_against_oracle(engine):
    """the per-GPU shard of BASELINE config 4: 32768 utterances, 22.05 kHz, 2 s (44100 samples, not
    a multiple of the 24-sample super-step or of 8) -- a half-filled chip, i.e. the wave-specialised
    kernel at scale"""
    lanes, n, pcm = _full(engine, 4, 32768)
    assert n == 44100
    pick = list(range(0, 32768, 331))
    want = po.synth([lanes[i] for i in pick], n)
    assert np.array_equal(pcm[pick], want)


def test_fma_mode_full_batch_tolerance(engine, config3):
    """VS_ARITH_FMA over the full config-3 batch: RMS error against the exact mode on the
    /32768 scale must be far inside the north star's 1e-5"""
    lanes, n, pcm = config3
    engine.set_arith(vs.VS_ARITH_FMA)
    try:
        got = engine.synth(lanes, n)
    finally:
        engine.set_arith(vs.VS_ARITH_EXACT)
    diff = got.astype(np.int32) - pcm.astype(np.int32)
    nd = int(np.count_nonzero(diff))
    rms = float(np.sqrt(np.mean((diff / 32768.0) ** 2)))
    print("fma vs exact over %d samples: %d differ, rms %.3e" % (diff.size, nd, rms))
    assert np.abs(diff).max() <= 1
    assert rms <= 1e-5
    assert nd <= diff.size // 10**6


def test_odd_sample_counts_and_pitches(engine):
    """sample counts that are odd / not multiples of 8 or 24 take the scalar store path"""
    specs, fs, dur, _ = configs.config_specs(3, 70)
    lanes, d = vs.lanes_from_specs(specs)
    for n in (1, 23, 24, 25, 999, 16001):
        got = engine.synth(lanes, n)
        want = po.synth(lanes, n)
        assert np.array_equal(got, want), n


def test_longest_period_of_the_wide_ring_and_beyond(engine):
    """fs/F0 near the LDS limit of the 64-column ring (one workgroup per CU), where the reference itself
    overflows its w[500] buffer; beyond it the narrow build of the one-wave kernel takes over (16
    utterances per wavefront, four times the slots): slow, but the samples the loops define -- 48 kHz and
    96 kHz are rates the reference accepts (flowgen_shimmer.c:535-540) and its buffer grows with the
    period (fg:569).  Beyond THAT (a period of more than ~4500 samples): VS_ERR_UNSUPPORTED, not a wrong
    answer."""
    fa = ["-r", "44100", "-d", "0.6", "-f", "50", "-g", "52", "-j", "5", "-s", "10", "-n", "15"]
    lanes = []
    for seed in range(70):
        lane, dur = vs.lane_from_cli(fa, ["-v", "u", "-g", "3"], seed)
        lanes.append(lane)
    n = vs.num_samples(44100, dur)
    got = engine.synth(lanes, n)
    assert np.array_equal(got, po.synth(lanes, n))
    for fa, nl in ((["-r", "48000", "-d", "0.6", "-f", "50", "-g", "52", "-j", "5"], 40),
                   (["-r", "48000", "-d", "0.6", "-f", "50", "-g", "52", "-j", "5", "-s", "10", "-n", "15"], 21),
                   (["-r", "96000", "-d", "0.5", "-f", "60", "-g", "70"], 5),
                   (["-r", "96000", "-d", "0.5", "-f", "50", "-g", "52", "-j", "3", "-s", "4", "-n", "30", "-l", "0.1"], 33)):
        lanes = [vs.lane_from_cli(fa, ["-v", "aiu1234567"[seed % 10], "-g", "2"], 300 + seed)[0] for seed in range(nl)]
        n = 30000
        plan = engine.plan(lanes, n)
        assert "narrow build" in plan.kernel_name(vs.VS_KIND_SYNTH), fa
        plan.close()
        assert np.array_equal(engine.synth(lanes, n), po.synth(lanes, n)), fa
        assert np.array_equal(engine.source(lanes, n), po.source(lanes, n)), fa
    lane, dur = vs.lane_from_cli(["-r", "192000", "-d", "0.6", "-f", "50", "-g", "52", "-j", "5"], ["-v", "a"], 1)
    with pytest.raises(vs.VsError) as e:
        engine.synth([lane], 1000)
    assert e.value.code == vs._ffi.VS_ERR_UNSUPPORTED


def test_which_fused_kernel_a_plan_takes(engine):
    """the roles of the wave-specialised kernel are chosen by the shape of the work, not by what happens to
    fit the LDS: three (open phase | noise | filter) wherever the rings are deep -- a full grid (BASELINE config 3; config
    5's F0 sweep too, once every group has the ring depth ITS periods need: mixed rings), half-filled chips (config 4's
    shard: the filter wavefront alone on its SIMD, open phase and noise together on the next; a 16384-utterance chunk: a SIMD
    per wavefront), with glottal noise or without (config 2's shape: since round 6, profiles/r06_roles_without_noise.txt) --
    and two where the rings hold barely a cycle (the F0 sweep over uniform rings: a filter wavefront that waits for all of
    its lanes starves there)."""
    def kernel(cfg, n):
        specs, fs, dur, _ = configs.config_specs(cfg, n)
        lanes, d = vs.lanes_from_specs(specs)
        plan = engine.plan(lanes, vs.num_samples(fs, d))
        name = plan.kernel_name(vs.VS_KIND_SYNTH)
        plan.close()
        return name
    assert kernel(3, 65536) == "vs_synth_ws_kernel<0, true, 3>"
    assert kernel(5, 65536) == "vs_synth_ws_kernel<0, true, 3>"
    engine.set_tuning(mixed_rings=-1)
    try:
        assert kernel(5, 65536) == "vs_synth_ws_kernel<0, true, 2>"
    finally:
        engine.set_tuning()
    assert kernel(2, 65536) == "vs_synth_ws_kernel<0, true, 3>"
    assert kernel(4, 32768) == "vs_synth_ws_kernel<0, true, 3>"
    assert kernel(3, 16384) == "vs_synth_ws_kernel<0, true, 3>"   # a chunk of the delivery pipelines
    assert kernel(2, 1024) == "vs_synth_ws_kernel<0, true, 3>"
    # ... over rings of 2.4 of the longest cycle where at most two groups share a CU, 1.7 (capped by the LDS) on full grids
    def slots(cfg, n):
        specs, fs, dur, _ = configs.config_specs(cfg, n)
        lanes, d = vs.lanes_from_specs(specs)
        plan = engine.plan(lanes, vs.num_samples(fs, d))
        r = plan.info()["ring_slots"]
        plan.close()
        return r
    assert slots(4, 32768) == 552 and slots(3, 16384) == 408 and slots(3, 65536) == 288


def test_mixed_batch_every_option_combination(engine):
    """one batch whose lanes differ in everything: rate, F0, options on/off, vowel, output noise"""
    specs = []
    k = 0
    for fs in ("8000", "16000", "44100"):
        for opts in ([], ["-j", "2"], ["-s", "8"], ["-n", "12"], ["-j", "1", "-s", "3", "-n", "25", "-z", "0.4"],
                     ["-n", "30", "-l", "0.2"], ["-c", "0.9", "-k", "1.0", "-j", "4"]):
            for va in (["-v", "a"], ["-v", "5", "-g", "2", "-p", "0.4"], ["-v", "i", "-n", "15"]):
                k += 1
                specs.append((["-r", fs, "-d", "0.5", "-f", "%d" % (90 + 7 * (k % 20)), "-g", "%d" % (100 + 8 * (k % 20))] + opts,
                              va, 1000 + k))
    lanes = []
    for fa, va, seed in specs:
        lane, dur = vs.lane_from_cli(fa, va, seed)
        lanes.append(lane)
    n = 7777   # one sample count for the batch (durations differ in the reference; here n is the batch's)
    got = engine.synth(lanes, n)
    want = po.synth(lanes, n)
    assert np.array_equal(got, want), int((got != want).sum())


def _fuzz_lanes(seed, count, dur="0.5", specs=None):
    """specs, when given, receives the (flowgen args, vowel args, seed) of every lane kept"""
    rng = np.random.default_rng(seed)
    lanes = []
    while len(lanes) < count:
        fs = int(rng.choice([8000, 11025, 16000, 32000, 44100]))
        f0 = float(rng.uniform(60, 400))
        fg = f0 * float(rng.uniform(1.01, 1.5)) + 0.5
        fa = ["-r", str(fs), "-d", dur, "-f", "%.2f" % f0, "-g", "%.2f" % fg]
        if rng.random() < 0.6:
            fa += ["-j", "%.2f" % rng.uniform(0, 10)]
        if rng.random() < 0.6:
            fa += ["-s", "%.2f" % rng.uniform(0, 30)]
        if rng.random() < 0.6:
            fa += ["-n", "%.1f" % rng.uniform(0, 50)]
        if rng.random() < 0.3:
            fa += ["-l", "%.3f" % rng.uniform(0, 0.29)]
        if rng.random() < 0.5:
            fa += ["-z", "%.2f" % rng.uniform(0, 1)]
        if rng.random() < 0.5:
            fa += ["-k", "%.2f" % rng.uniform(0.5, 1.2)]
        if rng.random() < 0.5:
            fa += ["-c", "%.2f" % rng.uniform(0.05, 1.0)]
        if rng.random() < 0.5:
            fa += ["-a", str(int(rng.integers(1, 32766)))]
        va = ["-v", str(rng.choice(list("aiu1234567")))]
        if rng.random() < 0.5:
            va += ["-g", "%.2f" % rng.uniform(1, 20)]
        if rng.random() < 0.5:
            va += ["-p", "%.2f" % rng.uniform(0, 1)]
        if rng.random() < 0.3:
            va += ["-n", "%.1f" % rng.uniform(1, 40)]
        try:
            lane, d = vs.lane_from_cli(fa, va, int(rng.integers(0, 2**63)))
        except vs.VsError:
            continue                      # the reference would answer usage(), e.g. F0 < 50 after rounding
        if vs.load().vs_lane_validate(C.byref(lane)) != 0:
            continue
        if int(np.float32(lane.fs) / np.float32(lane.F0)) * 1.2 > 500:
            continue                      # long periods next to 64 different cos rows exceed the LDS (tested separately)
        lanes.append(lane)
        if specs is not None:
            specs.append((fa, va, int(lane.seed)))
    return lanes


def _pick(rng, ends, lo, hi, fmt):
    """an end point / special value with probability 1/2, otherwise uniform over [lo, hi]"""
    if rng.random() < 0.5:
        return ends[int(rng.integers(0, len(ends)))]
    return fmt % rng.uniform(lo, hi)


def _corner_lanes(seed, count, specs=None):
    rng = np.random.default_rng(seed)
    lanes = []
    while len(lanes) < count:
        fs = int(rng.choice([8000, 11025, 16000, 32000, 44100, 48000]))
        f0 = float(_pick(rng, ["50", "50.01", "120", "399.99"], 50, 400, "%.2f"))
        fg = f0 * float(rng.choice([1.0001, 1.04, 1.5, 10.0])) + float(rng.choice([0.01, 1.0]))
        fa = ["-r", str(fs), "-d", "0.5", "-f", "%.2f" % f0, "-g", "%.2f" % fg]
        if rng.random() < 0.6:
            fa += ["-j", _pick(rng, ["0", "0.01", "10", "50", "200", "1000"], 0, 10, "%.2f")]
        if rng.random() < 0.6:
            fa += ["-s", _pick(rng, ["0", "0.01", "50", "100"], 0, 30, "%.2f")]
        if rng.random() < 0.7:
            fa += ["-n", _pick(rng, ["0", "0.1", "50"], 0, 50, "%.1f")]
        if rng.random() < 0.6:
            fa += ["-l", _pick(rng, ["0", "0.001", "0.3"], 0, 0.3, "%.3f")]
        if rng.random() < 0.5:
            fa += ["-z", _pick(rng, ["0", "1"], 0, 1, "%.2f")]
        if rng.random() < 0.6:
            fa += ["-k", _pick(rng, ["0.5", "1", "2", "5", "40"], 0.5, 1.2, "%.2f")]
        if rng.random() < 0.6:
            fa += ["-c", _pick(rng, ["0.01", "0.05", "1"], 0.05, 1.0, "%.2f")]
        if rng.random() < 0.7:
            fa += ["-a", _pick(rng, ["0", "1", "2", "50", "32766", "30000"], 1, 32766, "%.0f")]
        va = ["-v", str(rng.choice(list("aiu1234567")))]
        if rng.random() < 0.5:
            va += ["-g", _pick(rng, ["1", "100", "1000"], 1, 20, "%.2f")]
        if rng.random() < 0.5:
            va += ["-p", _pick(rng, ["0", "1"], 0, 1, "%.2f")]
        if rng.random() < 0.3:
            va += ["-n", _pick(rng, ["0.1", "60"], 1, 40, "%.1f")]
        try:
            lane, d = vs.lane_from_cli(fa, va, int(rng.integers(0, 2**63)))
        except vs.VsError:
            continue                      # the reference would answer usage()
        if vs.load().vs_lane_validate(C.byref(lane)) != 0:
            continue                      # e.g. a closed quotient that leaves no pulse
        if int(np.float32(lane.fs) / np.float32(lane.F0)) * 1.2 > 500:
            continue                      # long periods next to 64 different cos rows exceed the LDS
        lanes.append(lane)
        if specs is not None:
            specs.append((fa, va, int(lane.seed)))
    return lanes


def test_random_parameter_fuzz(engine):
    """600 lanes with randomly drawn command lines over the whole option space the reference
    accepts (seeded): rates, F0/Fg, closed quotient, closure speed and its variation, jitter up
    to the 10 % limit, shimmer, SNR, DC flow, amplitude, every vowel table, gain, pre-emphasis,
    output noise.  One batch, bit-exact against the oracle."""
    lanes = _fuzz_lanes(424242, 600)
    n = 6000
    got = engine.synth(lanes, n)
    want = po.synth(lanes, n)
    bad = np.flatnonzero((got != want).any(axis=1))
    assert bad.size == 0, "lanes %s differ" % bad[:10]
    flow = engine.source(lanes, n)
    assert np.array_equal(flow, po.source(lanes, n))


@pytest.mark.parametrize("kernel", [dict(kernel=vs.VS_KERNEL_AUTO), dict(kernel=vs.VS_KERNEL_SINGLE),
                                    dict(kernel=vs.VS_KERNEL_WS, ws_roles=3)])
def test_random_parameter_fuzz_large(kernel):
    """the same draw of command lines at scale: 12000 lanes (lanes that take the generator's short
    sequences next to lanes that cannot, every option on and off), every sample against the oracle,
    with the default kernel choice, with the one-wave kernel and with the three-role kernel forced"""
    lanes = _fuzz_lanes(20261004, 12000)
    n = 5000
    eng = vs.Engine(0)
    eng.set_tuning(**kernel)
    try:
        got = eng.synth(lanes, n)
    finally:
        eng.close()
    bad = 0
    for lo in range(0, len(lanes), 4000):
        want = po.synth(lanes[lo:lo + 4000], n, threads=32)
        bad += int((got[lo:lo + 4000] != want).any(axis=1).sum())
    fast = sum(1 for l in lanes[:2000] if _is_fast(l))
    print("fuzz: %d of the first 2000 lanes take the short sequences" % fast)
    assert bad == 0, "%d lanes differ" % bad


def test_corner_parameter_fuzz():
    """every option drawn from the END POINTS of the range the reference's parser accepts, mixed with
    ordinary values (zero DC flow, amplitude 0 / 1 / 32766, jitter up to the parser's real limit of
    1000 %, shimmer 100 %, closing speed 40, closed quotient 1, SNR 0 and 50 dB, gain 1000): rare
    settings in combination are the rule here.  tools/fuzz_soak.py runs the same generator over
    many seeds."""
    lanes = _corner_lanes(31337, 8000)
    n = 5000
    want = po.synth(lanes, n, threads=32)
    for kernel in (dict(kernel=vs.VS_KERNEL_AUTO), dict(kernel=vs.VS_KERNEL_SINGLE), dict(kernel=vs.VS_KERNEL_WS, ws_roles=3)):
        eng = vs.Engine(0)
        eng.set_tuning(**kernel)
        try:
            got = eng.synth(lanes, n)
        finally:
            eng.close()
        bad = int((got != want).any(axis=1).sum())
        assert bad == 0, "%d lanes differ (kernel %s)" % (bad, kernel)


def test_rows_at_the_recommended_pitch_equal_dense_rows(engine):
    """the PCM buffer's row pitch is the caller's (vs_row_pitch names the fast one): the same plan into pitched rows and
    into dense rows gives the same samples, and the bytes between the rows are left alone"""
    specs, fs, dur, _ = configs.config_specs(3, 4096 + 37)
    lanes, d = vs.lanes_from_specs(specs)
    n = 4001   # odd: the last super-step stores sample by sample
    n_lanes = len(lanes)
    pitch = vs.row_pitch(n)
    assert pitch > n and pitch % 64 == 0
    want = engine.synth(lanes, n)
    buf = engine.dev_alloc(n_lanes * pitch * 2)
    try:
        poison = np.full((n_lanes, pitch), 0x5A5A, dtype=np.int16)
        engine.dev_upload(buf, poison)
        plan = engine.plan(lanes, n)
        try:
            plan.launch(vs.VS_KIND_SYNTH, buf, out_pitch=pitch)
            plan.status()
        finally:
            plan.close()
        got = engine.dev_download(buf, (n_lanes, pitch))
        assert np.array_equal(got[:, :n], want)
        assert (got[:, n:] == 0x5A5A).all()
    finally:
        engine.dev_free(buf)


def test_a_plan_goes_up_next_to_a_running_launch(engine):
    """a caller who synthesises NEW utterances makes plan k + 1 while kernel k runs -- and the fused kernel fills every CU
    for its whole duration, so anything the runtime moves with a kernel of its own (copies of up to 16 KiB) waits for its
    END.  A plan's upload is two DMA-sized copies (records from page-locked memory; cos rows + taps, group table and error
    word in one block of >= 64 KiB): it must not take as long as what is left of the launch (profiles/r05_plan_cost.txt:
    0.25 ms next to the kernel, 1.8 ms when it waited)"""
    import time
    specs, fs, dur, _ = configs.config_specs(3, 65536)
    lanes, d = vs.lanes_from_specs(specs)
    n = vs.num_samples(fs, d)
    buf = engine.dev_alloc(65536 * n * 2)
    first = engine.plan(lanes, n)
    try:
        first.launch(vs.VS_KIND_SYNTH, buf)
        engine.synchronize()
        uploads = []
        for _ in range(3):
            t0 = time.perf_counter()
            first.launch(vs.VS_KIND_SYNTH, buf)          # ~2.6 ms of a full chip
            nxt = engine.plan(lanes, n)                   # ~1 ms of host work, then the upload: the launch is still running
            host_ms, upload_ms = nxt.timing()
            made_ms = (time.perf_counter() - t0) * 1e3
            engine.synchronize()
            launch_ms = (time.perf_counter() - t0) * 1e3
            nxt.close()
            uploads.append((upload_ms, host_ms, made_ms, launch_ms))
        first.status()
        best = min(uploads)
        assert best[3] > 2.0, uploads                      # the launch did outlast the plan ...
        assert best[2] < best[3] - 0.3, uploads            # ... which was finished well before it
        assert best[0] < 1.0, uploads                      # and its upload did not wait for the kernel's end
    finally:
        first.close()
        engine.dev_free(buf)


def _is_fast(lane):
    d = vs._ffi.DevLane()
    vs.load().vs_expand_lane(C.byref(lane), 0, C.byref(d))
    return bool(d.flags & vs._ffi.VS_DF_FAST)


def test_config4_whole_batch_on_one_device(engine):
    """ALL of BASELINE config 4 on one device: 262144 utterances x 44100 samples = 23.1 GB of PCM
    left in HBM (row offsets beyond 2^32 bytes, 4096 wavefront groups).  Sampled rows -- first,
    last, both sides of every 4 GiB boundary of the output -- against the oracle, and every row of
    the first and last per-GPU shard against the shard computed on its own."""
    n_lanes = 262144
    specs, fs, dur, _ = configs.config_specs(4, n_lanes)
    lanes, d = vs.lanes_from_specs(specs)
    n = vs.num_samples(fs, d)
    assert n == 44100
    pitch = (n + 7) & ~7
    row_bytes = pitch * 2
    buf = engine.dev_alloc(n_lanes * row_bytes)
    try:
        plan = engine.plan(lanes, n)
        try:
            plan.launch(vs.VS_KIND_SYNTH, buf, out_pitch=pitch)
            plan.status()
        finally:
            plan.close()

        def row(i):
            return engine.dev_download(buf + i * row_bytes, (pitch,))[:n]

        pick = {0, 1, 63, 64, n_lanes - 1, n_lanes - 64}
        for k in range(1, (n_lanes * row_bytes) >> 32):
            edge = (k << 32) // row_bytes
            pick |= {edge - 1, edge, edge + 1}
        pick |= set(range(777, n_lanes, 9973))
        pick = sorted(pick)
        want = po.synth([lanes[i] for i in pick], n)
        for j, i in enumerate(pick):
            assert np.array_equal(row(i), want[j]), i
        # placement: a per-GPU shard computed on its own equals its rows of the whole batch
        for shard in (0, 7):
            lo = shard * 32768
            sub = (vs.Lane * 32768)()
            C.memmove(sub, C.byref(lanes, lo * C.sizeof(vs.Lane)), 32768 * C.sizeof(vs.Lane))
            got = engine.synth(sub, n)
            whole = engine.dev_download(buf + lo * row_bytes, (32768, pitch))[:, :n]
            assert np.array_equal(got, whole), shard
    finally:
        engine.dev_free(buf)


def test_shape_fuzz_small_draw():
    """tools/shape_fuzz.py on a small draw: random batch sizes, sample counts, entry points
    (plan / vs_synth / rows callback / vs_source / vs_filter / node over logical shards) and kernels"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import shape_fuzz
    old = sys.argv
    sys.argv = ["shape_fuzz.py", "4242", "40"]
    try:
        assert shape_fuzz.main() == 0
    finally:
        sys.argv = old


def test_blocks_of_a_destroyed_plan_are_not_reused_before_its_launches_are_over():
    """vs_plan_destroy hands a plan's device blocks to the context's cache (hipFree would wait for the whole device, every
    batch: cli/vs_bench.c --fresh) and the next plan of that size takes them over -- behind an event recorded on the stream
    of the old plan's last launch.  A plan destroyed while its kernel is still running, and a successor made at once from
    its blocks with DIFFERENT utterances: both launches give what the oracle gives, again and again; vs_ctx_trim empties
    the cache"""
    eng = vs.Engine(0)
    n, ns = 16384, 16000
    outs = [eng.dev_alloc(n * ns * 2) for _ in range(2)]
    try:
        batches = []
        for b in range(6):
            specs, fs, dur, _ = configs.config_specs(3 if b % 2 == 0 else 5, n, lane0=1000 * b, seed0=77 + b)
            batches.append(vs.lanes_from_specs(specs)[0])
        for b, lanes in enumerate(batches):
            plan = eng.plan(lanes, ns)                 # (from b = 1 on: made from blocks the plan before has just given back)
            plan.launch(vs.VS_KIND_SYNTH, outs[b % 2])
            plan.close()                               # ... while its kernel runs
            if b >= 1:
                eng.synchronize()
                for k in (b - 1, b):
                    got = eng.dev_download(outs[k % 2], (n, ns), np.int16)
                    rows = np.arange(0, n, 37)
                    assert np.array_equal(got[rows], po.synth([batches[k][int(i)] for i in rows], ns, threads=32)), (b, k)
        # ... also when the caller moved the context to another stream between two launches of the plan: the new stream
        # waits for the old launches, so the block's event still stands behind every launch that reads it
        hip = C.CDLL("libamdhip64.so")               # (the runtime the library itself is linked against: already loaded)
        s1, s2 = C.c_void_p(), C.c_void_p()
        assert hip.hipStreamCreate(C.byref(s1)) == 0 and hip.hipStreamCreate(C.byref(s2)) == 0
        eng.set_stream(s1.value)
        plan = eng.plan(batches[1], ns)
        plan.launch(vs.VS_KIND_SYNTH, outs[0])       # on s1
        eng.set_stream(s2.value)
        plan.launch(vs.VS_KIND_SYNTH, outs[0])       # on s2: behind the first by the plan's own event
        plan.close()
        succ = eng.plan(batches[2], ns)               # takes the blocks over: waits for BOTH launches
        succ.launch(vs.VS_KIND_SYNTH, outs[1])
        succ.close()
        assert hip.hipDeviceSynchronize() == 0
        rows = np.arange(0, n, 41)
        for k, o in ((1, outs[0]), (2, outs[1])):
            got = eng.dev_download(o, (n, ns), np.int16)
            assert np.array_equal(got[rows], po.synth([batches[k][int(i)] for i in rows], ns, threads=32)), k
        eng.set_stream(0)
        assert hip.hipStreamDestroy(s1) == 0 and hip.hipStreamDestroy(s2) == 0
        assert vs.load().vs_ctx_trim(eng._ctx) == 0
        plan = eng.plan(batches[0], ns)
        plan.launch(vs.VS_KIND_SYNTH, outs[0])
        eng.synchronize()
        assert plan.status() == 0
        plan.close()
    finally:
        for o in outs:
            eng.dev_free(o)
        eng.close()
