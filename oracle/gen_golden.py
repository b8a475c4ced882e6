#!/usr/bin/env python3
"""Generates tests/golden/ from the REFERENCE ITSELF (oracle/_ref, built by oracle/Makefile from
/root/reference with the Philox random() shim).  TEST INFRASTRUCTURE ONLY.

Run in the build container (the reference does not exist on the GPU box):

    make -C oracle && python3 oracle/gen_golden.py

Outputs (data only -- inputs as command lines + seeds, outputs as int16 payloads):
    tests/golden/manifest.json   one entry per case: flowgen argv, vowel argv, seed, sample
                                 count, draw count, sha256 of both payloads
    tests/golden/cases.npz       flow_<name>, pcm_<name> int16 arrays
    tests/golden/stdout_*.txt    the reference's stdout for two cases (per-cycle S / SNRdb lines)
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import pyoracle as po  # noqa: E402
from voice_synth_amd import configs  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype="<i2").tobytes()).hexdigest()


def main():
    os.makedirs(OUT, exist_ok=True)
    cases = []

    def add(name, fa, va, seed=0, keep=True):
        cases.append((name, list(fa), list(va), int(seed), keep))

    # --- RNG-independent known answers of SURVEY.md section 4 (hash pinned there) ---
    g16 = ["-r", "16000", "-d", "1"]
    for v in "aiu1234567":
        add("ka_g16_v%s" % v, g16, ["-v", v], keep=(v in "a1"))
    add("ka_g16_va_g1", g16, ["-v", "a", "-g", "1"])
    add("ka_g22_va_g2_p09", ["-d", "2"], ["-v", "a", "-g", "2", "-p", "0.9"])
    add("ka_f80", g16 + ["-f", "80", "-g", "84"], ["-v", "a"], keep=False)
    add("ka_f150", g16 + ["-f", "150", "-g", "157"], ["-v", "a"], keep=False)
    add("ka_f300", g16 + ["-f", "300", "-g", "313"], ["-v", "a"], keep=False)

    # --- RNG-on: lanes of the BASELINE configurations ---
    for idx, lanes in ((2, [0, 1, 2, 1023]), (3, [0, 1, 2, 3, 4, 65535]), (4, [0, 1, 2, 3, 4]),
                       (5, [0, 1, 2, 3, 4, 5, 6, 7])):
        for l in lanes:
            specs, fs, dur, _ = configs.config_specs(idx, 1, lane0=l)
            fa, va, seed = specs[0]
            add("cfg%d_lane%d" % (idx, l), fa, va, seed)

    # --- edge cases of the source (every option, rejection loops under stress) ---
    add("edge_dc_kvar", g16 + ["-j", "3", "-s", "10", "-n", "5", "-z", "0.5", "-l", "0.1"],
        ["-v", "u", "-g", "1"], 11)
    add("edge_jit10_shim50", g16 + ["-j", "10", "-s", "50", "-n", "0"], ["-v", "2"], 12)
    add("edge_cq07_k08", g16 + ["-f", "300", "-g", "313", "-j", "1", "-s", "3", "-n", "10", "-k", "0.8",
                                "-c", "0.7"], ["-v", "i"], 13)
    add("edge_amp_hi", g16 + ["-a", "30000", "-n", "50", "-s", "2"], ["-v", "3", "-g", "1", "-p", "0"], 14)
    add("edge_amp0", g16 + ["-a", "0", "-n", "10"], ["-v", "a"], 15)
    add("edge_noise_only", g16 + ["-n", "20"], ["-v", "6", "-p", "0.5"], 16)
    add("edge_dur_frac", ["-r", "11025", "-d", "0.77", "-j", "2", "-n", "15"], ["-v", "5"], 17)
    add("edge_dc_max", g16 + ["-l", "0.25", "-n", "25", "-j", "0.5"], ["-v", "7", "-g", "3"], 18)
    add("edge_cq1", g16 + ["-c", "1", "-j", "5", "-n", "20"], ["-v", "a", "-g", "1"], 19)
    # --- vowel -n: white noise added to the filtered signal frame by frame (vowel_new.c:302-324) ---
    add("onoise_cfg3", g16 + ["-j", "1", "-s", "5.76", "-n", "20"], ["-v", "1", "-n", "20"], 20)
    add("onoise_22k", ["-d", "2", "-j", "1", "-n", "10"], ["-v", "4", "-g", "2", "-n", "5"], 21)
    add("onoise_frac", ["-r", "11025", "-d", "0.77"], ["-v", "a", "-n", "35", "-p", "0.3"], 22)
    add("edge_seed64", g16 + ["-j", "1", "-s", "5.76", "-n", "20"], ["-v", "4"], 0xFEDCBA9876543210)
    # --- T4 carried from cycle to cycle (flowgen_shimmer.c:114 is never reset): zero DC flow, so only
    #     a cycle whose amplitude passes 32767 and wraps the (short) conversion sets T4 (fg:319-323);
    #     every later cycle adds noise to [0, T4) and takes its power over [T4, T3) (fg:375-400) ---
    add("carry_t4_k119", ["-r", "11025", "-d", "0.6", "-f", "267.26", "-g", "367.88", "-c", "0.40", "-k", "1.19",
                          "-a", "27080", "-s", "16.67", "-n", "39.2", "-l", "0"], ["-v", "5"], 3)
    add("carry_t4_jit", g16 + ["-a", "30000", "-s", "10", "-n", "20", "-l", "0", "-j", "1"], ["-v", "a", "-g", "1"], 1)
    add("carry_t4_22k", ["-d", "0.5", "-a", "32000", "-s", "3", "-n", "10", "-l", "0", "-z", "0.3"],
        ["-v", "3", "-g", "2", "-p", "0.5"], 1, keep=False)

    # --- periods beyond the engine's 64-column ring (the narrow build of the one-wave kernel): rates the
    #     reference accepts (fg:535-540); without glottal noise, where its w[500] buffer is not in play ---
    add("long_48k_f50_j5", ["-r", "48000", "-d", "0.6", "-f", "50", "-g", "52", "-j", "5"], ["-v", "a"], 31, keep=False)
    add("long_96k_f60", ["-r", "96000", "-d", "0.5", "-f", "60", "-g", "70", "-s", "4"], ["-v", "2", "-g", "2"], 32, keep=False)

    manifest = []
    arrays = {}
    for name, fa, va, seed, keep in cases:
        res = po.run_reference(fa, va, seed)
        entry = {
            "name": name,
            "flowgen_args": fa,
            "vowel_args": va,
            "seed": seed,
            "n_samples": int(len(res["flow"])),
            "ndraws": int(res["ndraws"]),
            "sha256_flow": sha(res["flow"]),
            "sha256_pcm": sha(res["pcm"]),
            "stored": bool(keep),
        }
        manifest.append(entry)
        if keep:
            arrays["flow_" + name] = res["flow"]
            arrays["pcm_" + name] = res["pcm"]
        if name in ("cfg3_lane0", "edge_dc_kvar"):
            with open(os.path.join(OUT, "stdout_%s.txt" % name), "wb") as f:
                f.write(res["flow_stdout"])
        print("%-24s n=%6d draws=%6d flow %s pcm %s" %
              (name, entry["n_samples"], entry["ndraws"], entry["sha256_flow"][:16], entry["sha256_pcm"][:16]))

    # -O2 build of the reference must agree with the -O0 build
    for name, fa, va, seed, keep in cases[:3] + cases[-6:]:
        r2 = po.run_reference(fa, va, seed, opt="O2")
        e = [m for m in manifest if m["name"] == name][0]
        assert sha(r2["flow"]) == e["sha256_flow"] and sha(r2["pcm"]) == e["sha256_pcm"], name

    np.savez_compressed(os.path.join(OUT, "cases.npz"), **arrays)
    with open(os.path.join(OUT, "manifest.json"), "w") as f:
        json.dump({"generator": "oracle/gen_golden.py", "reference_build": "gcc -O0 -w + oracle/rng_shim.c (LP64)",
                   "cases": manifest}, f, indent=1)
    print("wrote %d cases, %d stored payload pairs" % (len(manifest), len(arrays) // 2))


if __name__ == "__main__":
    main()
