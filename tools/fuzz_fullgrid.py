"""Parameter fuzz of the FULL-GRID kernels: one plan, one launch of tens of thousands of random utterances (the
suite's fuzz goes through the delivery pipeline, whose 16384-utterance chunks never take the full-grid shapes --
in particular never the mixed rings an F0 sweep gets on a full grid).  Every sample of every lane against the
CPU oracle.  Needs the GPU; the oracle is the checker.

    python tools/fuzz_fullgrid.py [first_seed] [n_seeds] [lanes] [samples]
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
from test_gpu_properties import _fuzz_lanes  # noqa: E402


def lanes_for(seed, count, noisy_share=0.85, one_cq=True, onoise=True):
    """the suite's random utterances, most of them with glottal noise (three roles want a noisy majority), one sample rate
    per batch (a plan has one length -- and, since round 6, one frame length: the third of the utterances that ask for
    `vowel -n` get their frame powers from the filter wavefronts of the fused kernel, vs_synth_ws_pow_kernel; onoise=False
    drops the output noise as rounds 4 and 5 did, when it sent the whole plan to the one-wave kernel).  one_cq: every
    utterance keeps the default closed quotient, so that a group of 64 neighbouring periods needs a handful of cos rows
    and the mixed rings fit; with random quotients a group stages up to 64 rows and the plan stays on uniform rings"""
    rng = np.random.default_rng(seed + 77)
    base = _fuzz_lanes(seed, count)
    fs = int(rng.choice([16000, 22050, 11025]))
    for lane in base:
        if not onoise:
            lane.out_snr = 0.0
        lane.fs = fs
        if one_cq:
            lane.cq = 0.55
        if rng.random() < noisy_share:
            lane.flags |= vs.VS_FLAG_NOISE
            if lane.noise <= 0:
                lane.noise = float(10 ** rng.uniform(0, 5))
            if lane.DC == 0 and not (lane.flags & vs.VS_FLAG_NOISE and lane.DC):
                lane.DC = 0.25      # what -n sets (fg:182)
    ok = [l for l in base if vs.load().vs_lane_validate(vs.C.byref(l)) == 0
          and 50 <= l.F0 < l.Fg and int(np.float32(l.fs) / np.float32(l.F0)) * 1.2 <= 500]
    return ok


def run(seed, count, n, one_cq=True, onoise=True):
    lanes = lanes_for(seed, count, one_cq=one_cq, onoise=onoise)
    arr = (vs.Lane * len(lanes))(*lanes)
    eng = vs.Engine(0)
    try:
        plan = eng.plan(arr, n)
        out = eng.dev_alloc(len(lanes) * n * 2)
        plan.launch(vs.VS_KIND_SYNTH, out)
        eng.synchronize()
        st = plan.status()
        got = eng.dev_download(out, (len(lanes), n), np.int16)
        info = dict(plan.info(), **plan.roles())
        name = plan.kernel_name()
        eng.dev_free(out)
        plan.close()
    finally:
        eng.close()
    bad = 0
    for lo in range(0, len(lanes), 8192):
        hi = min(len(lanes), lo + 8192)
        want = po.synth(lanes[lo:hi], n, threads=32)
        bad += int((got[lo:hi] != want).any(axis=1).sum())
    return len(lanes), bad, st, name, info


def main():
    seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    n_seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    count = int(sys.argv[3]) if len(sys.argv) > 3 else 45000
    n = int(sys.argv[4]) if len(sys.argv) > 4 else 4000
    total = bad_total = 0
    t0 = time.time()
    for seed in range(seed0, seed0 + n_seeds):
        k, bad, st, name, info = run(seed, count, n, one_cq=(seed % 2 == 0))   # odd seeds: random closed quotients, uniform rings
        total += k
        bad_total += bad
        print("seed %d: %d lanes x %d samples, %s, roles %d %s, ring slots up to %d, LDS per workgroup %d B, status %d: %s  (%.0f s)"
              % (seed, k, n, name, info["roles"], info["layout"], info["ring_slots"], info["lds_bytes"], st,
                 "ok" if bad == 0 else "%d LANES DIFFER" % bad, time.time() - t0), flush=True)
    print("full-grid fuzz: %d lanes over %d seeds, %d differing lanes" % (total, n_seeds, bad_total))
    return 1 if bad_total else 0


if __name__ == "__main__":
    sys.exit(main())
