"""VS_ARITH_FMA over the parameter fuzz: how far the fused-multiply-add recurrence gets from the
exact one (the oracle) -- maximum |difference| in LSB, differing samples, RMS on the /32768 scale.

    python tools/fuzz_fma.py [first_seed] [n_seeds] [lanes] [samples] [fma|f32]

"f32": the same survey for VS_ARITH_F32 (report only: its contract is the per-table distance of tests/golden/f32_bounds.json,
and utterances the wave-specialised kernels do not serve run VS_ARITH_FMA).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import voice_synth_amd as vs  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from test_gpu_properties import _corner_lanes, _fuzz_lanes  # noqa: E402


def main():
    seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n_seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    n_lanes = int(sys.argv[3]) if len(sys.argv) > 3 else 12000
    n = int(sys.argv[4]) if len(sys.argv) > 4 else 5000
    mode = sys.argv[5] if len(sys.argv) > 5 else "fma"
    eng = vs.Engine(0, arith=vs.VS_ARITH_F32 if mode == "f32" else vs.VS_ARITH_FMA)
    rms_all = []
    worst = 0
    worst_plain = 0   # ... over utterances WITHOUT vowel -n
    for name, gen in (("uniform", _fuzz_lanes), ("corners", _corner_lanes)):
        for seed in range(seed0, seed0 + n_seeds):
            lanes = gen(seed, n_lanes)
            want = po.synth(lanes, n, threads=32).astype(np.int32)
            got = eng.synth(lanes, n).astype(np.int32)
            d = got - want
            nd = int(np.count_nonzero(d))
            mx = int(np.abs(d).max())
            worst = max(worst, mx)
            rms_all.append(float(np.sqrt(np.mean((d / 32768.0) ** 2))))
            rms = float(np.sqrt(np.mean((d / 32768.0) ** 2)))
            print("%s seed %d: %d of %d samples differ, max |d| %d LSB, rms %.2e" % (name, seed, nd, d.size, mx, rms), flush=True)
            if mx > 1:   # who: utterances with the vowel stage's own noise (its width follows the frame's power) or without
                rows = np.flatnonzero(np.abs(d).max(axis=1) > 1)
                with_n = [int(r) for r in rows if lanes[r].out_snr > 0]
                worst_plain = max(worst_plain, max([int(np.abs(d[r]).max()) for r in rows if not lanes[r].out_snr > 0], default=0))
                print("    > 1 LSB in %d utterances, %d of them with vowel -n; e.g. row %d: gain %g, out_snr %g" %
                      (rows.size, len(with_n), rows[0], lanes[rows[0]].gain, lanes[rows[0]].out_snr), flush=True)
    eng.close()
    print("worst |difference| %d LSB; without vowel -n %d LSB" % (worst, max(worst_plain, min(worst, 1))))
    print("%s: RMS of full scale per seed: median %.2e, max %.2e" % (mode, float(np.median(rms_all)), max(rms_all)))
    if mode == "f32":
        return 0
    return 0 if worst_plain <= 1 and worst <= 2 else 1


if __name__ == "__main__":
    sys.exit(main())
