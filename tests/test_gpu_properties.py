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
    lanes, n, pcm = _full(engine, 5, 16384)
    pick = list(range(0, 16384, 97))
    want = po.synth([lanes[i] for i in pick], n)
    assert np.array_equal(pcm[pick], want)


def test_config4_shape_sampled_against_oracle(engine):
    """22.05 kHz, 2 s (44100 samples, not a multiple of the 24-sample super-step or of 8)"""
    lanes, n, pcm = _full(engine, 4, 4096)
    assert n == 44100
    pick = list(range(0, 4096, 131))
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
