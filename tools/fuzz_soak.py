"""Soak run of the parameter fuzz (tests/test_gpu_properties.py::_fuzz_lanes) over many seeds:
every sample of every lane against the CPU oracle, default kernel choice and one-wave kernel,
plus the source-only kind.  Needs the GPU; the oracle is the checker.

    python tools/fuzz_soak.py [first_seed] [n_seeds] [lanes] [samples] [uniform|corners|sets22|sets40]

"uniform" draws every option uniformly over its usual range (the generator of the test suite);
"corners" draws every option from the END POINTS of the range the reference's parser accepts
(flowgen_shimmer.c:470-546, vowel_new.c:123-187) mixed with ordinary values -- zero DC flow, amplitudes 0, 1
and 32766, jitter up to the parser's real limit of 1000 %, shimmer 100 %, closing speeds far above 1,
closed quotient 1, SNR 0 and 50 dB -- so that combinations of rare settings are the rule.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import voice_synth_amd as vs  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from test_gpu_properties import _corner_lanes, _fuzz_lanes  # noqa: E402


LOG_LANES = 1500   # lanes per seed whose cycle records are compared
LOG_CAP = 700      # records kept per lane (0.5 s at 400 Hz is 200 cycles; jitter can shorten them)


def main():
    seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    n_seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    n_lanes = int(sys.argv[3]) if len(sys.argv) > 3 else 12000
    n = int(sys.argv[4]) if len(sys.argv) > 4 else 5000
    mode = sys.argv[5] if len(sys.argv) > 5 else "uniform"
    gen = _corner_lanes if mode == "corners" else _fuzz_lanes
    if mode in ("sets22", "sets40"):
        # a third of the lanes get an explicit coefficient set of random order: up to 22 taps (fused
        # kernels) or up to MAX_ORDER 40 (the whole plan takes the wide path)
        from voice_synth_amd import configs

        def gen(seed, count, top=22 if mode == "sets22" else 40):
            lanes = _fuzz_lanes(seed, count)
            rng = np.random.default_rng(seed + 5)
            for lane in lanes[::3]:
                order = int(rng.integers(1, top + 1))
                A = configs.random_pole_set(order, rng, rmax=0.95)
                lane.vowel = 0
                lane.order = order
                for j in range(len(lane.A)):
                    lane.A[j] = float(A[j]) if j <= order else 0.0
            return lanes
    bad_total = 0
    lanes_total = 0
    t0 = time.time()
    for seed in range(seed0, seed0 + n_seeds):
        lanes = gen(seed, n_lanes)
        want = po.synth(lanes, n, threads=32)
        want_flow = po.source(lanes, n, threads=32)
        for kernel in (vs.VS_KERNEL_AUTO, vs.VS_KERNEL_SINGLE):
            eng = vs.Engine(0)
            eng.set_tuning(kernel=kernel)
            try:
                got = eng.synth(lanes, n)
                flow = eng.source(lanes, n) if kernel == vs.VS_KERNEL_AUTO else None
            finally:
                eng.close()
            bad = int((got != want).any(axis=1).sum())
            if flow is not None:
                bad += int((flow != want_flow).any(axis=1).sum())
            bad_total += bad
            if bad:
                rows = np.flatnonzero((got != want).any(axis=1))[:5]
                print("seed %d kernel %d: %d lanes differ, first rows %s" % (seed, kernel, bad, rows), flush=True)
        # the per-cycle records of the single-utterance programs (S, x_pow, w_pow, T: what the
        # reference prints, flowgen_shimmer.c:307,409) come from a different instantiation of the
        # generator (no short sequences): a slice of the lanes through it
        sub = lanes[:LOG_LANES]
        eng = vs.Engine(0)
        try:
            flow, recs, ncyc = eng.source(sub, n, log_cycles=LOG_CAP)
        finally:
            eng.close()
        bad = int((flow != want_flow[:len(sub)]).any(axis=1).sum())
        for i, lane in enumerate(sub):
            _, wrecs, wn, _ = po.source_one(lane, n, LOG_CAP)
            k = min(wn, LOG_CAP)
            same = int(ncyc[i]) == wn
            for f in ("S", "x_pow", "w_pow", "T"):
                same = same and np.array_equal(recs[i][f][:k], wrecs[f][:k], equal_nan=(f != "T"))
            if not same:
                bad += 1
                print("seed %d lane %d: cycle records differ" % (seed, i), flush=True)
        bad_total += bad
        lanes_total += len(lanes)
        print("seed %d: %d lanes x %d samples, both kernels + source: %s  (%.0f s)"
              % (seed, len(lanes), n, "ok" if bad_total == 0 else "DIFFERENCES", time.time() - t0), flush=True)
    print("soak: %d lanes over %d seeds, %d differing lanes" % (lanes_total, n_seeds, bad_total))
    return 1 if bad_total else 0


if __name__ == "__main__":
    sys.exit(main())
