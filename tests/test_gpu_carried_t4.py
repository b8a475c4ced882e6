"""T4 of flowgen_shimmer.c:114 is declared outside the cycle loop and only ever assigned inside
`if(x[i] < par.DC)` (fg:319-323): a cycle that never goes below the DC flow inherits the T4 of an
earlier cycle.  With zero DC flow (`-l 0`) the only way below it is an amplitude above 32767
wrapping the (short) conversion, so T4 is set by a rare cycle and then CARRIED: every later cycle
adds noise to [0, T4) (fg:385-391) and takes its open-phase power over [T4, T3) (fg:375-378).

Found by tools/fuzz_soak.py (1 lane in 30000 of the option fuzz: it needs `-l 0.000`, noise and an
amplitude that shimmer can push past 32767); the three command lines it reported are kept here
next to a batch built to live in that regime.  Golden vectors of the same regime, produced by the
compiled reference: tests/golden carry_t4_*."""
import numpy as np
import pytest

import voice_synth_amd as vs
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu

FOUND = [
    (["-r", "32000", "-d", "0.5", "-f", "347.94", "-g", "517.48", "-s", "6.48", "-n", "25.7", "-l", "0.000",
      "-k", "0.64", "-a", "27613"], ["-v", "1", "-p", "0.28"], 640316100324252851),
    (["-r", "11025", "-d", "0.5", "-f", "267.26", "-g", "367.88", "-s", "16.67", "-n", "39.2", "-l", "0.000",
      "-k", "1.19", "-c", "0.40", "-a", "27080"], ["-v", "5", "-g", "18.52"], 6074900611788198163),
    (["-r", "16000", "-d", "0.5", "-f", "312.89", "-g", "420.97", "-s", "2.90", "-n", "4.2", "-l", "0.000",
      "-z", "0.14", "-k", "0.81", "-a", "25625"], ["-v", "a", "-g", "12.50", "-p", "0.38"], 1652936060977174108),
]


def _regime_lanes(n_lanes):
    rng = np.random.default_rng(20261004)
    lanes = []
    for k in range(n_lanes):
        fs = int(rng.choice([11025, 16000, 44100]))
        fa = ["-r", str(fs), "-d", "0.5",
              "-a", str(int(rng.integers(24000, 32767))), "-s", "%.2f" % rng.uniform(2, 25),
              "-n", "%.1f" % rng.uniform(3, 45), "-l", "0"]
        f0 = float(rng.uniform(max(90.0, fs / 300.0), 330))   # periods of at most 300 samples
        fa += ["-f", "%.2f" % f0, "-g", "%.2f" % (f0 * 1.2 + 1)]
        if k % 3 == 0:
            fa += ["-j", "%.2f" % rng.uniform(0, 5)]
        if k % 4 == 0:
            fa += ["-k", "%.2f" % rng.uniform(0.5, 1.2), "-z", "%.2f" % rng.uniform(0, 0.5)]
        lane, _ = vs.lane_from_cli(fa, ["-v", str(rng.choice(list("aiu1234567")))], 1 + k)
        lanes.append(lane)
    return lanes


def _carrying_cycles(lane, n):
    """cycles of the oracle's flow that start with noise (x[0] is the DC flow, 0, otherwise) although
    their own peak stays below the wrap"""
    flow, recs, ncyc, _ = po.source_one(lane, n, 2000)
    starts = np.concatenate([[0], np.cumsum(recs["T"][:ncyc])[:-1]])
    count = 0
    for c in range(ncyc):
        a, b = int(starts[c]), int(starts[c] + recs["T"][c])
        if b <= n and flow[a] != 0 and flow[a:b].max() < 32000:
            count += 1
    return count


@pytest.mark.parametrize("kernel", [dict(kernel=vs.VS_KERNEL_SINGLE), dict(kernel=vs.VS_KERNEL_WS, ws_roles=2),
                                    dict(kernel=vs.VS_KERNEL_WS, ws_roles=3)])
def test_power_sum_starts_at_the_carried_t4(kernel):
    lanes = [vs.lane_from_cli(fa, va, seed)[0] for fa, va, seed in FOUND] + _regime_lanes(509)
    n = 5000
    assert sum(_carrying_cycles(l, n) for l in lanes[:40]) > 100   # the batch is in the regime
    eng = vs.Engine(0)
    eng.set_tuning(**kernel)
    try:
        flow = eng.source(lanes, n)
        pcm = eng.synth(lanes, n)
    finally:
        eng.close()
    assert np.array_equal(flow, po.source(lanes, n))
    assert np.array_equal(pcm, po.synth(lanes, n))


def test_carried_t4_next_to_lanes_on_the_short_sequences(engine):
    """the same lanes interleaved with lanes that take the generator's short sequences: a wavefront
    with both runs the general sequence for all of them"""
    from voice_synth_amd import configs
    specs, fs, dur, _ = configs.config_specs(3, 256)
    plain, d = vs.lanes_from_specs(specs)
    mixed = []
    odd = _regime_lanes(256)
    for k in range(256):
        mixed += [plain[k], odd[k]]
    n = 4000
    for l in mixed:
        assert vs.load().vs_lane_validate(vs.C.byref(l)) == 0
    got = engine.synth(mixed, n)
    assert np.array_equal(got, po.synth(mixed, n))
