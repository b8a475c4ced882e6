"""Every sample of every BASELINE batch against the CPU oracle -- not a sample of rows.

The oracle does ~50 Msamples/s on 16 host threads, so the full batches (1.05e9 .. 1.45e9
samples) are a matter of seconds each on the GPU box; the result is the strongest statement of the
parity contract: the int16 output of the shipped kernels equals the CPU restatement of the
reference (which the reference's own outputs pin) in EVERY sample of BASELINE configurations 2
(65536-utterance shape), 3, 4 (per-GPU shard) and 5 (both readings of "randomised formant sets")."""
import numpy as np
import pytest

import voice_synth_amd as vs
from voice_synth_amd import configs
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu

THREADS = 32


def _compare(engine, lanes, ns, block=8192):
    """GPU result through vs_synth (the delivery pipeline), oracle block by block"""
    got = engine.synth(lanes, ns)
    n = len(lanes)
    bad = 0
    for lo in range(0, n, block):
        hi = min(n, lo + block)
        want = po.synth([lanes[i] for i in range(lo, hi)], ns, threads=THREADS)
        bad += int((got[lo:hi] != want).sum())
    return bad, got.size


@pytest.mark.parametrize("index,n", [(3, 65536), (5, 65536), (2, 65536), (4, 32768)])
def test_full_batch_every_sample(engine, index, n):
    specs, fs, dur, label = configs.config_specs(index, n)
    lanes, d = vs.lanes_from_specs(specs)
    ns = vs.num_samples(fs, d)
    bad, total = _compare(engine, lanes, ns)
    print("%s: %d of %d samples differ" % (label, bad, total))
    assert bad == 0


def test_full_batch_blended_pole_sets_every_sample(engine):
    lanes, fs, dur, label = configs.config5_blended_lanes(65536)
    ns = vs.num_samples(fs, dur)
    bad, total = _compare(engine, lanes, ns)
    print("%s: %d of %d samples differ" % (label, bad, total))
    assert bad == 0


@pytest.mark.parametrize("index,n,kernel,mixed", [(3, 65536, "vs_synth_ws_kernel<0, true, 3>", 0), (3, 65536 - 219, "vs_synth_ws_kernel<0, true, 3>", 0),
                                                  (5, 65536 - 219, "vs_synth_ws_kernel<0, true, 3>", 0), (5, 65536, "vs_synth_ws_kernel<0, true, 3>", 0),
                                                  (5, 65536 - 219, "vs_synth_ws_kernel<0, true, 2>", -1)])
def test_one_launch_of_the_whole_batch_every_sample(engine, index, n, kernel, mixed):
    """what bench.py times -- ONE launch of the plan of the whole batch (the tests above go through the
    delivery pipeline, i.e. plans of 16384 utterances, which never take the full-grid kernels): the
    three-role kernel on config 3, with a last group of 37 utterances and three empty groups behind it in
    its workgroup; config 5's F0 sweep over MIXED rings (every workgroup holds groups from across the period range, each
    with the ring depth its periods need: three roles everywhere), with a ragged last group and whole; and the same sweep
    over uniform rings (vs_tuning.mixed_rings = -1): the two-role kernel with the divergent filter loop"""
    specs, fs, dur, label = configs.config_specs(index, n)
    lanes, d = vs.lanes_from_specs(specs)
    ns = vs.num_samples(fs, d)
    engine.set_tuning(mixed_rings=mixed) if mixed else engine.set_tuning()
    plan = engine.plan(lanes, ns)
    engine.set_tuning()
    out = engine.dev_alloc(n * ns * 2)
    try:
        assert plan.kernel_name(vs.VS_KIND_SYNTH) == kernel
        plan.launch(vs.VS_KIND_SYNTH, out)
        engine.synchronize()
        assert plan.status() == 0
        got = engine.dev_download(out, (n, ns), np.int16)
    finally:
        engine.dev_free(out)
        plan.close()
    bad = 0
    for lo in range(0, n, 8192):
        hi = min(n, lo + 8192)
        bad += int((got[lo:hi] != po.synth([lanes[i] for i in range(lo, hi)], ns, threads=THREADS)).sum())
    print("%s, one launch of %d utterances: %d of %d samples differ" % (label, n, bad, got.size))
    assert bad == 0


@pytest.mark.parametrize("index,n,roles", [(3, 65536, 3), (5, 65536 - 219, 3), (4, 32768, 3), (2, 1024, 3)])
def test_one_launch_with_output_noise_every_sample(engine, index, n, roles):
    """vowel -n (vowel_new.c:302-324) behind the kernels bench.py times: the plan of a batch in which every utterance asks
    for output noise still takes the wave-specialised kernel of its shape -- its filter wavefronts take the frame powers
    along (one frame length for the whole batch: vs_synth_ws_pow_kernel), a scan fills in what they left
    (vs_out_power_fill_kernel), the noise is one streaming pass over the finished PCM (vs_out_noise_kernel) -- and every
    sample equals the oracle's: config 3 (16 kHz: frames of 800 samples), config 5's F0 sweep over mixed
    rings with a ragged last group, config 4's shard (22.05 kHz, 2 s: frames of 1100 samples, rows that are no multiple of
    8 samples long, a last frame of 100) and config 2's shape (no glottal noise)"""
    specs, fs, dur, label = configs.config_specs(index, n, out_noise_db=20)
    lanes, d = vs.lanes_from_specs(specs)
    ns = vs.num_samples(fs, d)
    plan = engine.plan(lanes, ns)
    out = engine.dev_alloc(n * ns * 2)
    try:
        name = plan.kernel_name(vs.VS_KIND_SYNTH)
        assert name == "vs_synth_ws_pow_kernel<0, true, %d> + vs_out_power_fill_kernel + vs_out_noise_kernel" % roles, name
        plan.launch(vs.VS_KIND_SYNTH, out)
        engine.synchronize()
        assert plan.status() == 0
        got = engine.dev_download(out, (n, ns), np.int16)
    finally:
        engine.dev_free(out)
        plan.close()
    bad = 0
    for lo in range(0, n, 8192):
        hi = min(n, lo + 8192)
        bad += int((got[lo:hi] != po.synth([lanes[i] for i in range(lo, hi)], ns, threads=THREADS)).sum())
    print("%s, one launch of %d utterances: %d of %d samples differ" % (label, n, bad, got.size))
    assert bad == 0


@pytest.mark.parametrize("index,n", [(3, 4096), (4, 2048), (3, 65536)])
def test_output_noise_when_the_fused_kernel_rounds_twice(index, n):
    """the filter wavefronts that take vowel -n's frame powers along cannot vouch for a frame during which round2int()'s
    quirk path rounded a super-step again (one chance in 2^32 per sample, or a signal below 2^-54): they mark it and the
    scan behind the launch (vs_out_power_fill_kernel) sums it from the finished PCM.  Provoked in every seventh super-step
    (VS_FAULT_REROUND) -- frames that end inside such a super-step, frames that start in one, on both launch shapes and
    both frame lengths: every sample still equals the oracle's"""
    specs, fs, dur, label = configs.config_specs(index, n, out_noise_db=17)
    lanes, d = vs.lanes_from_specs(specs)
    ns = vs.num_samples(fs, d)
    eng = vs.Engine(0)
    eng.set_tuning(fault=vs.VS_FAULT_REROUND)
    out = None
    try:
        plan = eng.plan(lanes, ns)
        out = eng.dev_alloc(n * ns * 2)
        assert plan.kernel_name(vs.VS_KIND_SYNTH).startswith("vs_synth_ws_pow_kernel<0, true, 3> + vs_out_power_fill_kernel")
        plan.launch(vs.VS_KIND_SYNTH, out)
        eng.synchronize()
        assert plan.status() == 0
        got = eng.dev_download(out, (n, ns), np.int16)
        plan.close()
    finally:
        if out:
            eng.dev_free(out)
        eng.close()
    bad = 0
    for lo in range(0, n, 8192):
        hi = min(n, lo + 8192)
        bad += int((got[lo:hi] != po.synth([lanes[i] for i in range(lo, hi)], ns, threads=THREADS)).sum())
    assert bad == 0


def test_output_noise_over_two_frame_lengths_in_one_launch(engine):
    """a batch whose utterances differ in rate has no frame length the filter wavefronts could share: the plain
    wave-specialised kernel, and every frame's power from the streaming pass (vs_out_power_kernel)"""
    lanes = []
    for l in range(4096):
        rate = "16000" if l % 3 else "32000"
        lane, dur = vs.lane_from_cli(["-r", rate, "-d", "1", "-j", "1", "-s", "3", "-n", "25"], ["-v", "aiu1234567"[l % 10], "-g", "3", "-n", "14"], 900 + l)
        lanes.append(lane)
    ns = 16000
    plan = engine.plan(lanes, ns)
    out = engine.dev_alloc(len(lanes) * ns * 2)
    try:
        assert plan.kernel_name(vs.VS_KIND_SYNTH) == "vs_synth_ws_kernel<0, true, 3> + vs_out_power_kernel + vs_out_noise_kernel"
        plan.launch(vs.VS_KIND_SYNTH, out)
        engine.synchronize()
        assert plan.status() == 0
        got = engine.dev_download(out, (len(lanes), ns), np.int16)
    finally:
        engine.dev_free(out)
        plan.close()
    assert np.array_equal(got, po.synth(lanes, ns, threads=THREADS))


def test_mixed_rings_plan_on_the_one_wave_kernel(engine):
    """a mixed-rings plan (config 5's F0 sweep, full grid) launched as what does NOT take the wave-specialised kernel: the
    source-only kind and the per-cycle log run the one-wave kernel with ONE group's LDS -- a ring of the plan's deepest
    depth + its cos rows, not the four-group sum of a mixed-rings workgroup (which left one wavefront per CU) -- and give
    the oracle's flow, cycle counts and log"""
    n = 65536 - 219
    specs, fs, dur, label = configs.config_specs(5, n)
    lanes, d = vs.lanes_from_specs(specs)
    ns = vs.num_samples(fs, d)
    plan = engine.plan(lanes, ns)
    recs_per_lane = 420
    rec_dt = np.dtype([("S", "<f4"), ("x_pow", "<f4"), ("w_pow", "<f4"), ("T", "<i4")])
    out = engine.dev_alloc(n * ns * 2)
    log = engine.dev_alloc(n * recs_per_lane * rec_dt.itemsize)
    ncyc = engine.dev_alloc(n * 4)
    try:
        assert plan.info()["lds_bytes"] > 64 * 1024          # mixed rings: several rings per workgroup
        assert plan.kernel_name(vs.VS_KIND_SOURCE).startswith("vs_synth_kernel<0, 1,")
        plan.launch(vs.VS_KIND_SOURCE, out, log_ptr=log, log_pitch=recs_per_lane, ncyc_ptr=ncyc)
        engine.synchronize()
        assert plan.status() == 0
        flow = engine.dev_download(out, (n, ns), np.int16)
        recs = engine.dev_download(log, (n, recs_per_lane), rec_dt)
        counts = engine.dev_download(ncyc, (n,), np.int32)
    finally:
        engine.dev_free(out)
        engine.dev_free(log)
        engine.dev_free(ncyc)
        plan.close()
    bad = 0
    for lo in range(0, n, 8192):
        hi = min(n, lo + 8192)
        bad += int((flow[lo:hi] != po.source([lanes[i] for i in range(lo, hi)], ns, threads=THREADS)).sum())
    assert bad == 0
    for l in list(range(0, n, 4099)) + [n - 1]:
        _, want_recs, want_n, _ = po.source_one(lanes[l], ns, max_recs=recs_per_lane)
        assert counts[l] == want_n
        for field in ("T", "S", "x_pow", "w_pow"):
            assert np.array_equal(recs[l, :want_n][field], want_recs[field]), (l, field)


def test_full_grid_of_random_utterances_over_mixed_rings():
    """one launch of ~40000 random utterances (F0 60-400 Hz, every option drawn at random, most with glottal noise): a
    full grid whose groups differ in period, i.e. the mixed-rings plan -- groups from across the period range share a
    workgroup, each with the ring depth its periods need -- and whatever roles it picks; every sample against the oracle"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_fullgrid
    k, bad, st, name, info = fuzz_fullgrid.run(4242, 42000, 3000)
    print("%d lanes, %s, %s" % (k, name, info))
    assert k > 33000 and st == 0 and bad == 0
    # a third of the utterances ask for vowel -n: the fused kernel takes the frame powers along (pre-emphasis differs per
    # utterance: the general instantiation), a scan and the noise pass follow
    assert name.startswith("vs_synth_ws_pow_kernel<0, false, 3> + vs_out_power_fill_kernel") and info["lds_bytes"] > 64 * 1024  # mixed rings
    # ... and the same utterances without output noise: the plain kernel
    k, bad, st, name, info = fuzz_fullgrid.run(4242, 42000, 3000, onoise=False)
    assert k > 33000 and st == 0 and bad == 0 and name == "vs_synth_ws_kernel<0, false, 3>"
