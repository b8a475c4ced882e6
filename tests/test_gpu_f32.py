"""VS_ARITH_F32 -- the opt-in packed single-precision filter of the wave-specialised kernels (SURVEY.md 8 f4, F19;
/root/reference/vowel_new.c:279-281 is the recurrence whose precision is traded) -- held to its MEASURED distance from
the exact oracle: tests/golden/f32_bounds.json is what tools/f32_survey.py --write measured on an MI355X (64 utterances x
16000 samples per vowel table / gain / pre-emphasis, fixed seeds); the kernels must stay within it."""
import json
import os
import sys

import numpy as np
import pytest

import voice_synth_amd as vs
from voice_synth_amd import configs
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def f32_engine():
    eng = vs.Engine(0, arith=vs.VS_ARITH_F32)
    yield eng
    eng.close()


def test_f32_mode_stays_within_its_measured_bounds(f32_engine):
    """per table at gain 10 / pre-emphasis 1 (the reference's defaults), gain 1 / pre-emphasis 1 and gain 10 / pre-emphasis 0:
    RMS <= the committed figure + 10 %, largest |difference| <= the committed one; and the table itself says what the
    header says: single precision is OUTSIDE 1e-5 RMS for some tables at the default gain (why the mode is opt-in)"""
    import f32_survey
    bounds = json.load(open(os.path.join(ROOT, "tests", "golden", "f32_bounds.json")))["cases"]
    worst = 0.0
    for v in f32_survey.TABLES:
        for g, p in (("10", "1"), ("1", "1"), ("10", "0")):
            lanes = f32_survey.case_lanes(v, g, p)
            got, name = f32_survey.launch_once(f32_engine, lanes, 16000)
            assert name.startswith("vs_synth_ws_kernel<2,"), name
            pc, mx, rms = f32_survey.stats(got, po.synth(lanes, 16000))
            b = bounds["%s/%s/%s" % (v, g, p)]
            assert rms <= 1.1 * b["rms"] + 1e-9, (v, g, p, rms, b)
            assert mx <= b["max"], (v, g, p, mx, b)
            worst = max(worst, rms)
    assert 1e-5 < max(b["rms"] for b in bounds.values()) < 4e-5
    assert worst < 4e-5


def test_f32_mode_on_the_baseline_shapes(f32_engine):
    """one launch of config 3 (three roles, full grid), config 4's shard (three roles, spread layout), config 2's shape (no
    glottal noise) and config 5 (mixed rings), 4096..65536 utterances: the single-precision kernels are the ones that run, the
    output stays within 32 LSB of the exact oracle everywhere and within 4e-5 RMS"""
    for index, n, check in ((3, 65536, 2048), (4, 32768, 1024), (2, 1024, 1024), (5, 65536, 2048)):
        specs, fs, dur, label = configs.config_specs(index, n)
        lanes, d = vs.lanes_from_specs(specs)
        ns = vs.num_samples(fs, d)
        plan = f32_engine.plan(lanes, ns)
        out = f32_engine.dev_alloc(n * ns * 2)
        try:
            assert plan.kernel_name(vs.VS_KIND_SYNTH).startswith("vs_synth_ws_kernel<2,"), plan.kernel_name(vs.VS_KIND_SYNTH)
            plan.launch(vs.VS_KIND_SYNTH, out)
            f32_engine.synchronize()
            assert plan.status() == 0
            got = f32_engine.dev_download(out, (n, ns), np.int16)
        finally:
            f32_engine.dev_free(out)
            plan.close()
        rows = np.linspace(0, n - 1, check).astype(int)
        want = po.synth([lanes[i] for i in rows], ns, threads=32)
        diff = got[rows].astype(np.int32) - want.astype(np.int32)
        rms = float(np.sqrt(np.mean((diff / 32768.0) ** 2)))
        print("%s: f32 max %d LSB, rms %.2e" % (label, np.abs(diff).max(), rms))
        assert np.abs(diff).max() <= 32 and rms < 4e-5, (label, np.abs(diff).max(), rms)


def test_f32_mode_elsewhere_is_the_fma_mode(f32_engine):
    """only the fused wave-specialised kernels have the single-precision filter: the source-only and filter-only kinds, the
    per-cycle log and coefficient sets of 23..40 taps run what VS_ARITH_FMA runs (and say so), vowel -n rides along"""
    specs, fs, dur, _ = configs.config_specs(3, 200)
    lanes, d = vs.lanes_from_specs(specs)
    ns = vs.num_samples(fs, d)
    plan = f32_engine.plan(lanes, ns)
    try:
        assert plan.kernel_name(vs.VS_KIND_SOURCE) == "vs_synth_kernel<0, 1, false, false>"
        assert plan.kernel_name(vs.VS_KIND_FILTER).startswith("vs_synth_kernel<1, 2,")
    finally:
        plan.close()
    flow = f32_engine.source(lanes, ns)
    assert np.array_equal(flow, po.source(lanes, ns))                 # the source is exact in every arithmetic
    ref = vs.Engine(0, arith=vs.VS_ARITH_FMA)
    try:
        assert np.array_equal(f32_engine.filter(lanes, flow), ref.filter(lanes, flow))
        wide, wfs, wdur = configs.wide_order_lanes([23, 31, 40, 40], lane0=3)
        wns = vs.num_samples(wfs, wdur)
        assert np.array_equal(f32_engine.synth(wide, wns), ref.synth(wide, wns))
    finally:
        ref.close()
    specs, fs, dur, _ = configs.config_specs(3, 4096, out_noise_db=20)
    lanes, d = vs.lanes_from_specs(specs)
    plan = f32_engine.plan(lanes, ns)
    out = f32_engine.dev_alloc(4096 * ns * 2)
    try:
        assert plan.kernel_name(vs.VS_KIND_SYNTH) == "vs_synth_ws_pow_kernel<2, true, 3> + vs_out_power_fill_kernel + vs_out_noise_kernel"
        plan.launch(vs.VS_KIND_SYNTH, out)
        f32_engine.synchronize()
        assert plan.status() == 0
        got = f32_engine.dev_download(out, (4096, ns), np.int16)
    finally:
        f32_engine.dev_free(out)
        plan.close()
    want = po.synth(lanes, ns, threads=32)
    diff = got.astype(np.int32) - want.astype(np.int32)
    # (the noise width follows the frame's power, so a filter that differs by a few LSB moves the noise by a few more)
    assert np.abs(diff).max() <= 64 and float(np.sqrt(np.mean((diff / 32768.0) ** 2))) < 6e-5
