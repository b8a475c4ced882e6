"""Drop-in check of the two programs over random command lines: voice_synth_amd/bin/flowgen_shimmer
and voice_synth_amd/bin/vowel against oracle/_ref (the reference itself, compiled with the Philox
random() shim) -- the .wav files byte for byte (72-byte LP64 header, which is what the reference
build here writes) and the complete stdout of both programs.  Needs the GPU and oracle/_ref.

    python tools/cli_fuzz.py [seed] [count]
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import voice_synth_amd as vs  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from test_gpu_properties import _corner_lanes, _fuzz_lanes  # noqa: E402

BIN = os.path.join(os.path.dirname(vs.__file__), "bin")


def reference_defined(lane):
    """the reference's buffers hold 2*fs/Fg samples (flowgen_shimmer.c:569) and 500 noise values
    (fg:115); and its T4 is an UNINITIALISED stack variable (fg:114) that only a sample
    below the DC flow assigns: with noise on and a DC flow of zero it is read before it is ever
    written, and what the compiled reference then does depends on the stack of the day"""
    tmax = int(1.2 * int(np.float32(lane.fs) / np.float32(lane.F0))) + 2
    if (lane.flags & vs.VS_FLAG_NOISE) and lane.DC <= 0:
        return False
    return tmax <= int(lane.fs / lane.Fg * 2) and tmax <= 500


def run_pair(dirname, exe_dir, fa, va, seed, ours):
    env = dict(os.environ, VS_SEED=str(seed))
    if ours:
        env["VS_WAV_HEADER"] = "72"
    else:
        env["VS_DRAWLOG"] = os.path.join(dirname, "draws.txt")
    fg = subprocess.run([os.path.join(exe_dir, "flowgen_shimmer"), "-o", "g.wav"] + fa, cwd=dirname, env=env,
                        capture_output=True, timeout=120)
    vw = subprocess.run([os.path.join(exe_dir, "vowel"), "-i", "g.wav", "-o", "o.wav"] + va, cwd=dirname, env=env,
                        capture_output=True, timeout=120)
    def slurp(name):
        try:
            return open(os.path.join(dirname, name), "rb").read()
        except OSError:
            return None
    return fg, vw, slurp("g.wav"), slurp("o.wav")


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    if not po.have_reference():
        sys.exit("oracle/_ref is not built")
    bad = done = 0
    batch = []
    for name, gen in (("uniform", _fuzz_lanes), ("corners", _corner_lanes)):
        specs = []
        lanes = gen(seed, 3 * count, specs=specs)
        k = 0
        for lane, (fa, va, s) in zip(lanes, specs):
            if k >= count:
                break
            if not reference_defined(lane):
                continue
            k += 1
            s &= (1 << 62) - 1
            with tempfile.TemporaryDirectory(prefix="vsr") as dr, tempfile.TemporaryDirectory(prefix="vso") as do:
                rfg, rvw, rg, ro = run_pair(dr, po.REF_DIR, fa, va, s, False)
                ofg, ovw, og, oo = run_pair(do, BIN, fa, va, s, True)
            if rg is None or ro is None or rfg.returncode != 0 or rvw.returncode != 0:
                # the reference itself fell over (undefined behaviour of its own): nothing to compare
                print("reference failed, skipped: %s | %s" % (" ".join(fa), " ".join(va)), flush=True)
                k -= 1
                continue
            same = (rg == og, ro == oo, rfg.stdout == ofg.stdout, rvw.stdout == ovw.stdout,
                    rfg.returncode == ofg.returncode, rvw.returncode == ovw.returncode)
            done += 1
            batch.append((fa, va, s, ro))
            if not all(same):
                bad += 1
                print("DIFFERENT (flow file, speech file, flowgen stdout, vowel stdout, rc, rc) = %s\n   %s | %s  seed %d"
                      % (same, " ".join(fa), " ".join(va), s), flush=True)
        print("%s: %d command lines compared" % (name, k), flush=True)
        # the same lines as ONE manifest through vs_batch (one launch per distinct sample count)
        with tempfile.TemporaryDirectory(prefix="vsb") as db:
            with open(os.path.join(db, "m.txt"), "w") as f:
                for i, (fa, va, s, _) in enumerate(batch):
                    f.write("seed=%d -o out%d.wav %s | %s\n" % (s, i, " ".join(fa), " ".join(va)))
            r = subprocess.run([os.path.join(BIN, "vs_batch"), "m.txt"], cwd=db, capture_output=True,
                               env=dict(os.environ, VS_WAV_HEADER="72"), timeout=300)
            nb = 0
            for i, (fa, va, s, ro) in enumerate(batch):
                try:
                    got = open(os.path.join(db, "out%d.wav" % i), "rb").read()
                except OSError:
                    got = b""
                if got != ro:
                    nb += 1
                    print("vs_batch output %d differs: %s | %s  seed %d" % (i, " ".join(fa), " ".join(va), s), flush=True)
            print("%s: vs_batch rc %d, %d of %d files differ from the reference's  (%s)"
                  % (name, r.returncode, nb, len(batch), r.stdout.decode(errors="replace").strip().splitlines()[-1] if r.stdout else ""), flush=True)
            bad += nb + (1 if r.returncode else 0)
        batch = []
    print("cli fuzz: %d command lines, %d with differences" % (done, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
