"""The packed single-precision filter (measurement variant: `make variant NAME=f32 DEFS=-DVS_EXP_F32`, where
VS_ARITH_FMA runs it in the wave-specialised kernel; select with VS_LIB=libvoicesynth_f32.so) against the
exact oracle, per vowel table / gain / pre-emphasis: differing samples, maximum |difference| in LSB, RMS
of full scale; and the launch times of BASELINE config 3 -> profiles/r02_f32_mode_measured.txt"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import voice_synth_amd as vs
from voice_synth_amd import configs
from oracle import pyoracle as po

eng = vs.Engine(0, arith=vs.VS_ARITH_FMA)


def stats(got, want):
    d = got.astype(np.int32) - want.astype(np.int32)
    return 100.0 * np.count_nonzero(d) / d.size, int(np.abs(d).max()), float(np.sqrt(np.mean((d / 32768.0) ** 2)))


fa = ["-r", "16000", "-d", "1", "-j", "1", "-s", "5.76", "-n", "20"]
print("per table / gain / pre-emphasis (64 utterances each, 16000 samples):")
worst = 0.0
for v in "aiu1234567":
    row = []
    for g, p in (("1", "1"), ("10", "1"), ("10", "0"), ("20", "0.5")):
        lanes = [vs.lane_from_cli(fa, ["-v", v, "-g", g, "-p", p], 100 + k)[0] for k in range(64)]
        got = eng.synth(lanes, 16000)
        want = po.synth(lanes, 16000)
        pc, mx, rms = stats(got, want)
        worst = max(worst, rms)
        row.append("g=%s p=%s: %4.1f %% max %d rms %.1e" % (g, p, pc, mx, rms))
    print("  -v %s   %s" % (v, " | ".join(row)))
print("worst rms over the tables: %.2e" % worst)
specs, fs, dur, label = configs.config_specs(3, 65536)
lanes, dd = vs.lanes_from_specs(specs)
plan = eng.plan(lanes, 16000)
out = eng.dev_alloc(65536 * 16000 * 2)
for arith, nm in ((vs.VS_ARITH_EXACT, "exact"), (vs.VS_ARITH_FMA, "fma (= the fp32 sequence in the variant library)")):
    eng.set_arith(arith)
    plan.launch(vs.VS_KIND_SYNTH, out); eng.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(5):
            plan.launch(vs.VS_KIND_SYNTH, out)
        eng.synchronize()
        ts.append((time.perf_counter() - t0) / 5 * 1e3)
    print("config 3, %s: %.3f ms per launch (%s)" % (nm, min(ts), plan.kernel_name(vs.VS_KIND_SYNTH)))
