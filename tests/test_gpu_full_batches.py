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
    assert name.startswith("vs_synth_ws_kernel") and info["lds_bytes"] > 64 * 1024     # mixed rings: several rings per workgroup
