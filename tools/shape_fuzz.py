"""Fuzz of SHAPES rather than parameters: random batch sizes (1 .. 40000 utterances, ragged against
the 64-lane groups, the 16384-utterance chunks and the workgroup shapes), random sample counts
(1 .. 20000, ragged against the 24-sample super-step and the 8-sample stores), random entry point
(plan launch with a padded pitch, vs_synth, vs_synth_rows, vs_source, vs_filter, the node entry over
1..8 logical shards), both fused kernels -- every sample against the CPU oracle.

    python tools/shape_fuzz.py [seed] [cases]
"""
import ctypes as C
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


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    cases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    rng = np.random.default_rng(seed)
    pool = _fuzz_lanes(seed, 6000) + _corner_lanes(seed + 1, 3000)
    bad = 0
    t0 = time.time()
    for case in range(cases):
        n_lanes = int(rng.choice([1, 2, 63, 64, 65, 127, 129, 1000, 4097, 16383, 16384, 16385, 16449, 20000, 32769, 40000,
                                  int(rng.integers(1, 40000))]))
        n = int(rng.choice([1, 7, 8, 23, 24, 25, 47, 48, 49, 100, 999, 4000, int(rng.integers(1, 20000))]))
        if n_lanes * n > 250_000_000:
            n = max(1, 250_000_000 // n_lanes)
        idx = rng.integers(0, len(pool), size=n_lanes)
        lanes = (vs.Lane * n_lanes)()
        for k, i in enumerate(idx):
            C.memmove(C.byref(lanes, k * C.sizeof(vs.Lane)), C.byref(pool[int(i)]), C.sizeof(vs.Lane))
            lanes[k].seed = int(rng.integers(0, 2**62))
            lanes[k].out_seed = lanes[k].seed
        entry = str(rng.choice(["plan", "synth", "rows", "source", "filter", "node"]))
        kernel = int(rng.choice([vs.VS_KERNEL_AUTO, vs.VS_KERNEL_SINGLE, vs.VS_KERNEL_WS]))
        eng = vs.Engine(0)
        eng.set_tuning(kernel=kernel)
        what = "%s lanes=%d n=%d kernel=%d" % (entry, n_lanes, n, kernel)
        try:
            if entry == "source":
                got = eng.source(lanes, n)
                want = po.source(lanes, n, threads=32)
            elif entry == "filter":
                flow = rng.integers(-32768, 32767, size=(n_lanes, n), dtype=np.int16)
                got = eng.filter(lanes, flow)
                want = po.filter(lanes, flow, threads=32)
            else:
                want = po.synth(lanes, n, threads=32)
                if entry == "synth":
                    got = eng.synth(lanes, n)
                elif entry == "rows":
                    got = np.zeros((n_lanes, n), dtype=np.int16)
                    seen = np.zeros(n_lanes, dtype=np.int32)

                    def sink(row0, block):
                        got[row0:row0 + len(block)] = block
                        seen[row0:row0 + len(block)] += 1
                        return 0
                    eng.synth_rows(lanes, n, sink)
                    assert (seen == 1).all(), "rows delivered %s times" % np.unique(seen)
                elif entry == "plan":
                    pitch = n + int(rng.integers(0, 9))
                    buf = eng.dev_alloc(n_lanes * pitch * 2 + 64)
                    plan = eng.plan(lanes, n)
                    plan.launch(vs.VS_KIND_SYNTH, buf, out_pitch=pitch)
                    plan.status()
                    plan.close()
                    got = eng.dev_download(buf, (n_lanes, pitch))[:, :n]
                    eng.dev_free(buf)
                else:
                    shards = int(rng.integers(1, 9))
                    what += " shards=%d" % shards
                    node = vs.Node([0] * shards)
                    buf = eng.dev_alloc(n_lanes * n * 2)
                    flags = int(rng.choice([vs.Node.OVERLAP, vs.Node.OVERLAP | vs.Node.STAGE_ALL, vs.Node.STAGE_ALL]))
                    node.synth_gather(lanes, n, buf, n, flags)
                    got = eng.dev_download(buf, (n_lanes, n))
                    eng.dev_free(buf)
                    node.close()
            nd = int((got != want).any(axis=1).sum())
        finally:
            eng.close()
        if nd:
            bad += 1
            print("DIFFERENT: %s: %d lanes differ" % (what, nd), flush=True)
        else:
            print("ok  %s  (%.0f s)" % (what, time.time() - t0), flush=True)
    print("shape fuzz: %d cases, %d with differences" % (cases, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
