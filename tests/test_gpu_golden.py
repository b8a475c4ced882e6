"""GPU engine against the committed golden fixtures directly (outputs of the reference itself,
tests/golden) -- no oracle in the loop."""
import hashlib
import json
import os

import numpy as np
import pytest

import voice_synth_amd as vs

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
MANIFEST = json.load(open(os.path.join(GOLD, "manifest.json")))["cases"]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype="<i2").tobytes()).hexdigest()


def _groups():
    """cases grouped by sample count, so each group is one batched launch"""
    g = {}
    for c in MANIFEST:
        g.setdefault(c["n_samples"], []).append(c)
    return sorted(g.items())


@pytest.mark.parametrize("n,cases", _groups(), ids=["n%d" % n for n, _ in _groups()])
def test_every_golden_case_bit_exact(engine, n, cases):
    lanes = []
    for c in cases:
        lane, dur = vs.lane_from_cli(c["flowgen_args"], c["vowel_args"], c["seed"])
        assert vs.num_samples(lane.fs, dur) == n
        lanes.append(lane)
    engine.set_arith(vs.VS_ARITH_EXACT)
    flow = engine.source(lanes, n)
    pcm = engine.synth(lanes, n)
    pcm2 = engine.filter(lanes, flow)
    for i, c in enumerate(cases):
        assert sha(flow[i]) == c["sha256_flow"], c["name"]
        assert sha(pcm[i]) == c["sha256_pcm"], c["name"]
        assert sha(pcm2[i]) == c["sha256_pcm"], c["name"]
