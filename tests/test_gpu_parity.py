"""GPU parity: the gfx950 engine (through the C ABI) against the CPU oracle, bit for bit.

Small seeded batches at sizes the oracle finishes in seconds.  VS_ARITH_EXACT must be
bit-identical; VS_ARITH_FMA must stay within the north-star tolerance (1e-5 RMS on the
/32768 scale) and is additionally expected to differ in at most a handful of samples.
"""
import numpy as np
import pytest

import voice_synth_amd as vs
from voice_synth_amd import configs
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu

TOL_RMS_NORMALISED = 1e-5  # BASELINE.json north_star


def _lanes(index, n, lane0=0):
    specs, fs, dur, _ = configs.config_specs(index, n, lane0)
    lanes, d = vs.lanes_from_specs(specs)
    return lanes, vs.num_samples(fs, d)


def _rms_norm(a, b):
    d = (a.astype(np.float64) - b.astype(np.float64)) / 32768.0
    return float(np.sqrt(np.mean(d * d)))


@pytest.mark.parametrize("index,n", [(1, 1), (2, 96), (3, 200), (4, 70), (5, 130)])
def test_source_bit_exact(engine, index, n):
    lanes, ns = _lanes(index, n)
    got = engine.source(lanes, ns)
    want = po.source(lanes, ns)
    assert np.array_equal(got, want), "flow differs: %d samples" % int((got != want).sum())


@pytest.mark.parametrize("index,n", [(1, 1), (2, 96), (3, 200), (4, 70), (5, 130)])
def test_synth_exact_bit_exact(engine, index, n):
    lanes, ns = _lanes(index, n)
    engine.set_arith(vs.VS_ARITH_EXACT)
    got = engine.synth(lanes, ns)
    want = po.synth(lanes, ns)
    assert np.array_equal(got, want), "pcm differs: %d samples" % int((got != want).sum())


@pytest.mark.parametrize("index,n", [(2, 96), (3, 200), (5, 130)])
def test_synth_fma_within_tolerance(engine, index, n):
    lanes, ns = _lanes(index, n)
    engine.set_arith(vs.VS_ARITH_FMA)
    try:
        got = engine.synth(lanes, ns)
    finally:
        engine.set_arith(vs.VS_ARITH_EXACT)
    want = po.synth(lanes, ns)
    ndiff = int((got != want).sum())
    assert _rms_norm(got, want) <= TOL_RMS_NORMALISED
    assert np.abs(got.astype(np.int32) - want.astype(np.int32)).max() <= 1
    assert ndiff <= max(2, got.size // 1000000), ndiff


def test_filter_only_bit_exact(engine):
    lanes, ns = _lanes(3, 64)
    rng = np.random.default_rng(7)
    flow = rng.integers(-20000, 20000, size=(64, ns), dtype=np.int16)
    got = engine.filter(lanes, flow)
    want = po.filter(lanes, flow)
    assert np.array_equal(got, want)


def test_filter_decay_into_denormals(engine):
    """vowel_new.c:413-427 returns 1 -- not 0 -- for a result in (-2^-54, 0): dec = x - floor(x)
    rounds to 1.0, x + 1 rounds to 1.0.  After an impulse the recurrence decays through the
    denormal range and keeps oscillating there, so the reference's output is a long pattern of
    0s and 1s that only an identical double state (denormals included) reproduces."""
    lanes = []
    for v in "aiu1234567":
        for g, p in (("1", "1"), ("10", "0"), ("3", "0.5")):
            lane, _ = vs.lane_from_cli(["-r", "16000", "-d", "1"], ["-v", v, "-g", g, "-p", p], 1)
            lanes.append(lane)
    n = 400000
    flow = np.zeros((len(lanes), n), dtype=np.int16)
    flow[:, 0] = 20000
    flow[:, 1000] = -3
    want = po.filter(lanes, flow)
    assert int((want[:, 100000:] == 1).sum()) > 100000       # the regime is reached
    got = engine.filter(lanes, flow)
    assert np.array_equal(got, want)
    engine.set_arith(vs.VS_ARITH_FMA)
    try:
        fma = engine.filter(lanes, flow)
    finally:
        engine.set_arith(vs.VS_ARITH_EXACT)
    assert np.abs(fma.astype(np.int32) - want.astype(np.int32)).max() <= 1


def test_cycle_log_matches_oracle(engine):
    lanes, ns = _lanes(3, 5)
    flow, recs, ncyc = engine.source(lanes, ns, log_cycles=400)
    for l in range(5):
        f, r, n, _ = po.source_one(lanes[l], ns, 400)
        assert n == ncyc[l]
        assert np.array_equal(f, flow[l])
        for name in ("S", "x_pow", "w_pow", "T"):
            assert np.array_equal(recs[l][:n][name], r[name]), name


KERNELS = {"single": dict(kernel=vs.VS_KERNEL_SINGLE), "ws": dict(kernel=vs.VS_KERNEL_WS, ws_roles=2),
           "ws3": dict(kernel=vs.VS_KERNEL_WS, ws_roles=3)}


@pytest.mark.parametrize("kernel", ["single", "ws", "ws3"])
def test_both_fused_kernels_bit_exact(kernel):
    """The plan picks the one-wave kernel for full grids and the wave-specialised kernel
    (generator wave + filter wave per 64 utterances, LDS progress words) for grids that leave
    half of the SIMDs empty; vs_ctx_set_tuning() forces either.  Both must be bit-exact on
    every shape."""
    eng = vs.Engine(0)
    eng.set_tuning(**KERNELS[kernel])
    try:
        for index, n in ((2, 96), (3, 200), (5, 130), (4, 70), (1, 1)):
            lanes, ns = _lanes(index, n)
            got = eng.synth(lanes, ns)          # vs_synth checks the kernel's spin-limit word
            assert np.array_equal(got, po.synth(lanes, ns)), (kernel, index)
        for arith in (vs.VS_ARITH_FMA,):
            eng.set_arith(arith)
            lanes, ns = _lanes(3, 200)
            got = eng.synth(lanes, ns)
            want = po.synth(lanes, ns)
            assert np.abs(got.astype(np.int32) - want.astype(np.int32)).max() <= 1
    finally:
        eng.close()


@pytest.mark.parametrize("kernel", ["single", "ws", "ws3"])
def test_noise_below_t4_is_never_consumed_early(kernel):
    """-l gives a DC flow > 0, so the noise also covers [0, T4) of every cycle and is added to
    samples that were written earlier in the same cycle.  The wave-specialised kernel publishes
    its progress inside a cycle; it must not hand those samples to the filter wave before the
    noise is in (a race that showed as a rare mismatch before it was fenced off)."""
    fa = ["-r", "16000", "-d", "0.5", "-j", "3", "-s", "10", "-n", "5", "-z", "0.5", "-l", "0.1"]
    lanes = []
    for seed in range(700):
        lane, dur = vs.lane_from_cli(fa, ["-v", "u", "-g", "1"], 5000 + seed)
        lanes.append(lane)
    ns = vs.num_samples(16000, dur)
    eng = vs.Engine(0)
    eng.set_tuning(**KERNELS[kernel])
    try:
        for _ in range(3):
            got = eng.synth(lanes, ns)
            assert np.array_equal(got, po.synth(lanes, ns))
    finally:
        eng.close()


def test_device_selftest(engine):
    """exhaustive on the device: the 3-instruction division shortcut equals IEEE division for
    all 2^31 draws; Philox known answer; integer square root; round2int against the literal form; the output-noise
    sample of vowel -n (one conversion instruction instead of round2int) over every float"""
    rc, fails = engine.selftest()
    assert rc == 0 and fails == [0, 0, 0, 0, 0, 0, 0, 0], fails


def test_wavefronts_are_dealt_to_the_simds_cyclically_and_plans_fall_back_when_not():
    """what the three-role layouts are built on: wavefront w of a workgroup runs on SIMD w % 4 -- read
    from HW_ID by a probe launch (vs_ctx_simd_dealing), for 12- and for 8-wavefront workgroups, on MI355X.  A plan on a
    device where it does not hold (provoked through vs_tuning.fault) takes two roles instead of three in the wrong order,
    says so (vs_plan_roles) -- and still synthesises the same samples."""
    eng = vs.Engine(0)
    try:
        assert eng.simd_dealing() == (True, True)
        for index, n, shape in ((3, 65536, ("role-major", "vs_synth_ws_kernel<0, true, 3>")), (4, 32768, ("spread", "vs_synth_ws_kernel<0, true, 3>"))):
            lanes, ns = _lanes(index, n)
            eng.set_tuning()
            plan = eng.plan(lanes, ns)
            assert plan.roles() == {"roles": 3, "layout": shape[0], "simd_fallback": False}
            assert plan.kernel_name() == shape[1]
            plan.close()
            eng.set_tuning(fault=vs.VS_FAULT_SIMD_DEALING)
            plan = eng.plan(lanes, ns)
            assert plan.roles() == {"roles": 2, "layout": "role-major", "simd_fallback": True}
            assert plan.kernel_name() == "vs_synth_ws_kernel<0, true, 2>"
            plan.close()
        # the fallback's samples are the oracle's
        lanes, ns = _lanes(3, 65536)
        sub = (vs.Lane * 192).from_buffer(lanes)
        eng.set_tuning(fault=vs.VS_FAULT_SIMD_DEALING, kernel=vs.VS_KERNEL_WS, ws_pairs=4)
        got = eng.synth(sub, ns)
        assert np.array_equal(got, po.synth(sub, ns))
    finally:
        eng.set_tuning()
        eng.close()


@pytest.mark.parametrize("index,n", [(2, 64), (3, 130), (4, 40), (5, 100)])
def test_every_config_at_unit_gain(engine, index, n):
    """the default gain 10 clips heavily (the rounding clamp hides differences in
    the clipped samples); every configuration is therefore also checked at the minimum gain 1,
    where far fewer samples clip, in both arithmetic modes."""
    lanes, ns = _lanes(index, n)
    for l in range(n):
        lanes[l].gain = 1.0
    want = po.synth(lanes, ns)
    engine.set_arith(vs.VS_ARITH_EXACT)
    assert np.array_equal(engine.synth(lanes, ns), want)
    engine.set_arith(vs.VS_ARITH_FMA)
    try:
        got = engine.synth(lanes, ns)
    finally:
        engine.set_arith(vs.VS_ARITH_EXACT)
    assert np.abs(got.astype(np.int32) - want.astype(np.int32)).max() <= 1
    assert _rms_norm(got, want) <= TOL_RMS_NORMALISED
