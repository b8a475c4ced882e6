"""Debug helper for a lane the fuzz soak reports: runs it alone on the GPU, compares flow and
per-cycle records with the oracle and prints the surroundings of the first difference.

    python tools/fuzz_debug.py seed:index [seed:index ...]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import voice_synth_amd as vs  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from test_gpu_properties import _fuzz_lanes  # noqa: E402


def main():
    n = 5000
    eng = vs.Engine(0)
    cache = {}
    for arg in sys.argv[1:]:
        seed, idx = (int(x) for x in arg.split(":"))
        if seed not in cache:
            cache[seed] = _fuzz_lanes(seed, 12000)
        lane = cache[seed][idx]
        want, wrecs, ncyc, nd = po.source_one(lane, n, 400)
        got, grecs, gncyc = eng.source([lane], n, log_cycles=400)
        got = got[0]
        plain = eng.source([lane], n)[0]
        print("   (without the cycle log: %d samples differ from the oracle, %d from the logged run)"
              % (int((plain != want).sum()), int((plain != got).sum())))
        got = plain
        diff = np.flatnonzero(got != want)
        print("== %s: %d samples differ, oracle cycles %d gpu cycles %d" % (arg, diff.size, ncyc, int(gncyc[0])))
        if diff.size == 0:
            continue
        i0 = int(diff[0])
        ends = np.cumsum(wrecs["T"])
        cyc = int(np.searchsorted(ends, i0, side="right"))
        start = int(ends[cyc - 1]) if cyc else 0
        print("first difference at sample %d = cycle %d (starts %d, T %d), offset %d; last diff %d"
              % (i0, cyc, start, int(wrecs["T"][cyc]), i0 - start, int(diff[-1])))
        for c in range(max(0, cyc - 1), min(ncyc, cyc + 2)):
            print("  cycle %d oracle %s gpu %s" % (c, wrecs[c], grecs[0][c]))
        lo = max(start, i0 - 6)
        print("  want", want[lo:i0 + 12].tolist())
        print("  got ", got[lo:i0 + 12].tolist())
        cs = int(wrecs["T"][cyc])
        print("  cycle want", want[start:start + cs].tolist())
        print("  cycle got ", got[start:start + cs].tolist())
    eng.close()


main()
