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
