"""Per-lane coefficient sets (VS_VOWEL_CUSTOM), the error path of the wave-specialised kernel,
and the regression cases of the round-1 review.

The reference's filter loop (vowel_new.c:266-289) runs over whatever A[] coefficients() loaded;
the engine keeps the 22 coefficients of every lane in that lane's registers, so a batch may carry
one coefficient set per utterance.  Bit-exact against the CPU oracle in VS_ARITH_EXACT, within
+-1 LSB in VS_ARITH_FMA, on both fused kernels."""
import ctypes as C

import numpy as np
import pytest

import voice_synth_amd as vs
from voice_synth_amd import _ffi, configs
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu

KERNELS = {"single": dict(kernel=vs.VS_KERNEL_SINGLE), "ws": dict(kernel=vs.VS_KERNEL_WS, ws_roles=2),
           "ws3": dict(kernel=vs.VS_KERNEL_WS, ws_roles=3)}


def _custom_lanes(n):
    """n lanes: distinct stable A[] (blended pole sets), per-lane gain (from config 5) and a
    per-lane pre-emphasis"""
    lanes, fs, dur, _ = configs.config5_blended_lanes(n)
    for l in range(n):
        lanes[l].pre_emphasis = [1.0, 0.0, 0.37, 0.9][l % 4]
    return lanes, vs.num_samples(fs, dur)


@pytest.mark.parametrize("kernel", ["single", "ws", "ws3"])
def test_custom_coefficient_sets_per_lane(kernel):
    lanes, ns = _custom_lanes(100)
    assert len({tuple(lanes[l].A[:]) for l in range(100)}) > 60       # really per lane
    want = po.synth(lanes, ns)
    eng = vs.Engine(0)
    eng.set_tuning(**KERNELS[kernel])
    try:
        got = eng.synth(lanes, ns)
        assert np.array_equal(got, want), "lanes %s differ" % np.flatnonzero((got != want).any(axis=1))[:10]
        eng.set_arith(vs.VS_ARITH_FMA)
        got = eng.synth(lanes, ns)
        assert np.abs(got.astype(np.int32) - want.astype(np.int32)).max() <= 1
    finally:
        eng.close()


def test_custom_set_equal_to_a_table_equals_the_table(engine):
    """A[] copied from table '4' must give what -v 4 gives (ties the custom path to the path the
    reference's golden vectors pin)"""
    lanes, ns = _custom_lanes(64)
    ref = (vs.Lane * 64)()
    C.memmove(ref, lanes, C.sizeof(ref))
    A = vs.vowel_coefficients("4")
    for l in range(64):
        for j in range(23):
            lanes[l].A[j] = float(A[j])
        ref[l].vowel = ord("4")
    assert np.array_equal(engine.synth(lanes, ns), engine.synth(ref, ns))


def test_lower_order_sets_by_zero_padding(engine):
    """orders below 22: trailing coefficients 0.0 (acc - 0*y == acc exactly), e.g. one resonator"""
    lanes, ns = _custom_lanes(64)
    for l in range(64):
        r, th = 0.90 + 0.001 * l, 0.1 + 0.02 * l
        for j in range(23):
            lanes[l].A[j] = 0.0
        lanes[l].A[0], lanes[l].A[1], lanes[l].A[2] = 1.0, -2.0 * r * np.cos(th), r * r
        lanes[l].gain = 1.0
    got = engine.synth(lanes, 4000)
    assert np.array_equal(got, po.synth(lanes, 4000))


def test_config5_blended_pole_sets_full_batch_sampled(engine):
    """BASELINE config 5, first reading of "randomised formant sets" (SURVEY.md 8d): 65536
    utterances, F0 sweep, each with its own blended pole set; sampled rows against the oracle"""
    lanes, fs, dur, _ = configs.config5_blended_lanes(65536)
    ns = vs.num_samples(fs, dur)
    pcm = engine.synth(lanes, ns)
    pick = list(range(0, 65536, 409))
    assert np.array_equal(pcm[pick], po.synth([lanes[i] for i in pick], ns))


@pytest.mark.parametrize("roles", [2, 3])
@pytest.mark.parametrize("pairs", [1, 2, 4])
def test_every_workgroup_shape_of_the_ws_kernel_on_ragged_batches(pairs, roles):
    """1, 2 and 4 groups per workgroup, two and three wavefronts per group (role-major layout), on
    batches that leave groups of the last workgroup partly or wholly without utterances"""
    eng = vs.Engine(0)
    eng.set_tuning(kernel=vs.VS_KERNEL_WS, ws_pairs=pairs, ws_roles=roles)
    try:
        for index, n in ((3, 130), (5, 321), (2, 64), (3, 1)):
            specs, fs, dur, _ = configs.config_specs(index, n)
            lanes, d = vs.lanes_from_specs(specs)
            ns = 5000
            got = eng.synth(lanes, ns)
            assert np.array_equal(got, po.synth(lanes, ns)), (pairs, index, n)
    finally:
        eng.close()


def test_config2_at_its_real_batch(engine):
    """BASELINE config 2 at its own size: 1024 utterances = 16 groups, the wave-specialised kernel"""
    specs, fs, dur, _ = configs.config_specs(2, 1024)
    lanes, d = vs.lanes_from_specs(specs)
    ns = vs.num_samples(fs, d)
    got = engine.synth(lanes, ns)
    assert np.array_equal(got, po.synth(lanes, ns))
    flow = engine.source(lanes, ns)
    assert np.array_equal(flow, po.source(lanes, ns))


@pytest.mark.parametrize("roles", [2, 3])
def test_ws_kernel_bounded_wait_reaches_the_caller(roles):
    """vs_tuning.fault withholds the generator wave's progress words: the bounded waits of the other
    wavefronts (filter; noise and filter in the three-role kernel) must run out, set the launch's
    error word and vs_synth must return VS_ERR_INTERNAL instead of hanging or returning garbage"""
    specs, fs, dur, _ = configs.config_specs(3, 200)
    lanes, d = vs.lanes_from_specs(specs)
    eng = vs.Engine(0)
    try:
        eng.set_tuning(kernel=vs.VS_KERNEL_WS, ws_roles=roles, fault=vs.VS_FAULT_WITHHOLD_PROGRESS, spin_limit=2000)
        with pytest.raises(vs.VsError) as e:
            eng.synth(lanes, 4000)
        assert e.value.code == _ffi.VS_ERR_INTERNAL
        # ... and the plan-level query reports the raw bits: bit 1 = the filter wave gave up
        plan = eng.plan(lanes, 4000)
        out = eng.dev_alloc(200 * 4000 * 2)
        plan.launch(vs.VS_KIND_SYNTH, out)
        with pytest.raises(vs.VsError):
            plan.status()
        eng.dev_free(out)
        plan.close()
        # the same context recovers once the fault is cleared
        eng.set_tuning()
        got = eng.synth(lanes, 4000)
        assert np.array_equal(got, po.synth(lanes, 4000))
    finally:
        eng.close()


@pytest.mark.parametrize("kernel,roles", [("single", 0), ("ws", 2), ("ws", 3)])
def test_cos_row_disagreement_reaches_the_caller(kernel, roles):
    """vs_tuning.fault makes the kernel believe the plan reserved no room for its cos rows: what used to be a device
    trap (a GPU fault that takes the context down) sets bit 3 of the launch's error word, the launch runs to its end,
    vs_plan_status / vs_synth answer VS_ERR_INTERNAL and the context stays usable"""
    specs, fs, dur, _ = configs.config_specs(3, 200)
    lanes, d = vs.lanes_from_specs(specs)
    eng = vs.Engine(0)
    try:
        kw = dict(kernel=vs.VS_KERNEL_SINGLE) if kernel == "single" else dict(kernel=vs.VS_KERNEL_WS, ws_roles=roles)
        eng.set_tuning(fault=vs.VS_FAULT_SHORT_COS_ROWS, **kw)
        with pytest.raises(vs.VsError) as e:
            eng.synth(lanes, 4000)
        assert e.value.code == _ffi.VS_ERR_INTERNAL
        plan = eng.plan(lanes, 4000)
        out = eng.dev_alloc(200 * 4000 * 2)
        plan.launch(vs.VS_KIND_SYNTH, out)
        flags = C.c_int(0)
        rc = vs.load().vs_plan_status(plan._plan, C.byref(flags))
        assert rc == _ffi.VS_ERR_INTERNAL and (flags.value & 8) == 8, (rc, flags.value)
        eng.dev_free(out)
        plan.close()
        eng.set_tuning(**kw)
        got = eng.synth(lanes, 4000)
        assert np.array_equal(got, po.synth(lanes, 4000))
    finally:
        eng.close()


def test_tuning_is_validated(engine):
    lib = vs.load()
    for bad in (dict(kernel=7), dict(ready_min=65), dict(ws_pairs=3), dict(gen_low=5), dict(gen_min=-1), dict(ws_roles=4),
                dict(fault=9), dict(ring_slots=-24)):
        t = vs.Tuning()
        for k, v in bad.items():
            setattr(t, k, v)
        assert lib.vs_ctx_set_tuning(engine._ctx, C.byref(t)) == _ffi.VS_ERR_ARG, bad
    assert lib.vs_ctx_set_tuning(engine._ctx, None) == 0


def test_long_period_on_a_half_filled_chip(engine):
    """review finding: 16385..32768 lanes select two generator/filter pairs per workgroup; with a
    long period (44.1 kHz, F0 60, jitter) two pairs do not fit the 160 KiB of LDS and the launch
    failed.  The plan now falls back to one pair (or to the one-wave kernel)."""
    fa = ["-r", "44100", "-d", "0.5", "-f", "60", "-g", "62", "-j", "3", "-s", "5", "-n", "20"]
    n = 16448
    lane, dur = vs.lane_from_cli(fa, ["-v", "a", "-g", "2"], 0)
    lanes = (vs.Lane * n)()
    for l in range(n):
        C.memmove(C.byref(lanes[l]), C.byref(lane), C.sizeof(vs.Lane))
        lanes[l].seed = 900 + l
        lanes[l].out_seed = 900 + l
    ns = 3000
    got = engine.synth(lanes, ns)
    pick = list(range(0, n, 257)) + [n - 1]
    assert np.array_equal(got[pick], po.synth([lanes[i] for i in pick], ns))


def test_output_noise_only_on_the_high_rate_lanes(engine):
    """review finding: the frame-power rows were sized from the lanes WITH vowel -n only; a lane
    without -n and a lower rate has more frames and overran its row into its neighbour's"""
    lanes = []
    for l in range(70):
        if l % 2 == 0:
            lane, dur = vs.lane_from_cli(["-r", "16000", "-d", "2", "-j", "1"], ["-v", "a", "-g", "2"], 40 + l)
        else:
            lane, dur = vs.lane_from_cli(["-r", "32000", "-d", "1", "-j", "1"], ["-v", "a", "-g", "2", "-n", "20"], 40 + l)
        lanes.append(lane)
    ns = 32000
    got = engine.synth(lanes, ns)
    want = po.synth(lanes, ns)
    assert np.array_equal(got, want), np.flatnonzero((got != want).any(axis=1))[:10]


def test_filter_only_accepts_any_sample_rate(engine):
    """review finding: vs_filter validated the SOURCE fields, so a 192 kHz flow (P = fs/F0 beyond
    the ring) was refused although the filter-only kind has no ring; the reference filters any rate"""
    lanes = []
    for l in range(64):
        lane = vs.default_lane()
        lane.fs = 192000
        lane.vowel = ord("aiu1234567"[l % 10])
        lane.gain = 1.0 + l % 5
        lane.out_snr = 100.0 if l % 3 == 0 else 0.0
        lane.out_seed = 77 + l
        lanes.append(lane)
    rng = np.random.default_rng(11)
    flow = rng.integers(-12000, 12000, size=(64, 20000), dtype=np.int16)
    got = engine.filter(lanes, flow)
    assert np.array_equal(got, po.filter(lanes, flow))


def test_small_host_calls_run_without_a_single_copy():
    """vs_source / vs_filter of a handful of utterances -- what the two drop-in programs call -- on a context that has
    not copied anything yet: the plan's records, the flow, the PCM, the cycle log and the counts live in pinned,
    device-mapped host memory (VS_PLAN_ZERO_COPY; a process that only does this never pays the runtime's 27 ms
    copy-path set-up).  Same samples, same log as the oracle; a device-side check that fails still reaches the caller
    (the error word is host memory the kernel writes with an atomic over PCIe)."""
    specs, fs, dur, _ = configs.config_specs(3, 5)
    lanes, d = vs.lanes_from_specs(specs)
    ns = vs.num_samples(fs, d)
    eng = vs.Engine(0)                      # fresh context: nothing has been copied on it
    try:
        flow, recs, ncyc = eng.source(lanes, ns, log_cycles=160)
        want_flow = po.source(lanes, ns)
        assert np.array_equal(flow, want_flow)
        for l in range(5):
            _, want_recs, want_ncyc, _ = po.source_one(lanes[l], ns, max_recs=160)
            assert ncyc[l] == want_ncyc
            for field in ("T", "S", "x_pow", "w_pow"):
                assert np.array_equal(recs[l, :ncyc[l]][field], want_recs[field])
            assert not recs[l, ncyc[l]:]["T"].any()      # rows behind the last cycle were zeroed (by the CPU here)
        pcm = eng.filter(lanes, flow)
        assert np.array_equal(pcm, po.synth(lanes, ns))
        assert np.array_equal(eng.source(lanes, ns), want_flow)       # the pooled block is reused
        eng.set_tuning(fault=vs.VS_FAULT_SHORT_COS_ROWS)
        with pytest.raises(vs.VsError) as e:
            eng.source(lanes, ns)
        assert e.value.code == _ffi.VS_ERR_INTERNAL
        eng.set_tuning()
        # ... and once the context has made a plan that copies, small calls take the copy path: same answer
        big, _ = vs.lanes_from_specs(configs.config_specs(3, 200)[0])
        assert np.array_equal(eng.synth(big, 2000)[:5], po.synth(big, 2000)[:5])
        assert np.array_equal(eng.source(lanes, ns), want_flow)
    finally:
        eng.close()


@pytest.mark.parametrize("index,n", [(3, 700), (5, 70000), (2, 300)])
def test_plan_reseed_is_a_new_plan_with_those_seeds(index, n):
    """vs_plan_reseed: the same utterances with new draws (what running the reference again does: it seeds from the clock,
    flowgen_shimmer.c:241) -- 16 bytes per lane go up instead of a new plan.  The reseeded plan must give, sample for sample,
    what a plan made from lanes with those seeds gives (config 5 at full-grid size: records in kernel order over mixed rings,
    seeds[i] still belongs to lane i), launches enqueued BEFORE the reseed keep the old seeds, and a second reseed right
    behind the first does not disturb it."""
    specs, fs, dur, _ = configs.config_specs(index, n)
    lanes, d = vs.lanes_from_specs(specs)
    ns = 3000
    eng = vs.Engine(0)
    try:
        plan = eng.plan(lanes, ns)
        out0, out1, out2 = (eng.dev_alloc(n * ns * 2) for _ in range(3))
        rng = np.random.default_rng(index)
        s1 = rng.integers(0, 2**63, size=n, dtype=np.uint64)
        s2 = rng.integers(0, 2**63, size=n, dtype=np.uint64)
        plan.launch(vs.VS_KIND_SYNTH, out0)          # old seeds
        plan.reseed(s1)
        plan.launch(vs.VS_KIND_SYNTH, out1)          # s1
        plan.reseed(s2, out_seeds=s1)
        plan.launch(vs.VS_KIND_SYNTH, out2)          # s2
        eng.synchronize()
        assert plan.status() == 0
        got = [eng.dev_download(o, (n, ns), np.int16) for o in (out0, out1, out2)]
        for o in (out0, out1, out2):
            eng.dev_free(o)
        plan.close()
        view = np.frombuffer(lanes, dtype=np.dtype(vs.Lane))
        want0 = po.synth(lanes, ns)
        view["seed"] = s1
        want1 = po.synth(lanes, ns)
        view["seed"] = s2
        want2 = po.synth(lanes, ns)
        assert np.array_equal(got[0], want0) and np.array_equal(got[1], want1) and np.array_equal(got[2], want2)
        assert not np.array_equal(got[0], got[1])
    finally:
        eng.close()


def test_plan_reseed_reaches_the_source_only_kind():
    """the one-wave kernel (source-only launches) reads the same records: a reseeded plan's flow is the oracle's for the new seeds"""
    specs, fs, dur, _ = configs.config_specs(3, 3)
    lanes, d = vs.lanes_from_specs(specs)
    ns = 2500
    eng = vs.Engine(0)
    try:
        plan = eng.plan(lanes, ns)
        out = eng.dev_alloc(3 * ns * 2)
        s1 = np.array([11, 22, 33], dtype=np.uint64)
        plan.reseed(s1)
        plan.launch(vs.VS_KIND_SOURCE, out)
        eng.synchronize()
        got = eng.dev_download(out, (3, ns), np.int16)
        eng.dev_free(out)
        plan.close()
        view = np.frombuffer(lanes, dtype=np.dtype(vs.Lane))
        view["seed"] = s1
        assert np.array_equal(got, po.source(lanes, ns))
    finally:
        eng.close()
