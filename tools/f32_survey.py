"""VS_ARITH_F32 (the packed single-precision filter of the wave-specialised kernels) against the exact CPU oracle, per
vowel table / gain / pre-emphasis: differing samples, largest |difference| in LSB, RMS of full scale.
    tools/f32_survey.py            prints the table (-> profiles/r06_f32_mode_measured.txt) and the config-3 launch times
    tools/f32_survey.py --write[=path]   also rewrites tests/golden/f32_bounds.json (or path: only gpurun_out/ travels back
                                   from the GPU box), the bounds tests/test_gpu_f32.py holds the kernels to (RMS <= table
                                   + 10 %, max |difference| <= table)
The cases are fixed (64 utterances each: config 3's source options, seeds 100..163, one launch of the two-role or
three-role kernel whichever the plan takes), so the figures are reproducible from HEAD on any MI355X."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import voice_synth_amd as vs
from voice_synth_amd import configs
from oracle import pyoracle as po

TABLES = "aiu1234567"
SETTINGS = (("1", "1"), ("10", "1"), ("10", "0"), ("20", "0.5"))
FLOWGEN = ["-r", "16000", "-d", "1", "-j", "1", "-s", "5.76", "-n", "20"]
BOUNDS = os.path.join(ROOT, "tests", "golden", "f32_bounds.json")


def case_lanes(v, g, p, n=64):
    return [vs.lane_from_cli(FLOWGEN, ["-v", v, "-g", g, "-p", p], 100 + k)[0] for k in range(n)]


def stats(got, want):
    d = got.astype(np.int32) - want.astype(np.int32)
    return 100.0 * np.count_nonzero(d) / d.size, int(np.abs(d).max()), float(np.sqrt(np.mean((d / 32768.0) ** 2)))


def launch_once(eng, lanes, ns):
    """one launch of the plan (the fused wave-specialised kernel: vs_synth would cut the batch into pipeline chunks)"""
    plan = eng.plan(lanes, ns)
    out = eng.dev_alloc(len(lanes) * ns * 2)
    try:
        name = plan.kernel_name(vs.VS_KIND_SYNTH)
        plan.launch(vs.VS_KIND_SYNTH, out)
        eng.synchronize()
        assert plan.status() == 0
        return eng.dev_download(out, (len(lanes), ns), np.int16), name
    finally:
        eng.dev_free(out)
        plan.close()


def survey(eng):
    """{ "v/g/p": {"pct": .., "max": .., "rms": ..} } for every table and setting"""
    res = {}
    for v in TABLES:
        for g, p in SETTINGS:
            lanes = case_lanes(v, g, p)
            got, name = launch_once(eng, lanes, 16000)
            assert name.startswith("vs_synth_ws_kernel<2,"), name
            pc, mx, rms = stats(got, po.synth(lanes, 16000))
            res["%s/%s/%s" % (v, g, p)] = {"pct": round(pc, 2), "max": mx, "rms": float("%.3e" % rms)}
    return res


def main():
    eng = vs.Engine(0, arith=vs.VS_ARITH_F32)
    res = survey(eng)
    print("# tools/f32_survey.py: VS_ARITH_F32 against the exact oracle, 64 utterances x 16000 samples per case")
    print("# (differing samples %, largest |difference| in LSB, RMS of full scale); kernel: the fused wave-specialised one")
    worst = 0.0
    for v in TABLES:
        row = []
        for g, p in SETTINGS:
            r = res["%s/%s/%s" % (v, g, p)]
            worst = max(worst, r["rms"])
            row.append("g=%s p=%s: %4.1f %% max %2d rms %.1e" % (g, p, r["pct"], r["max"], r["rms"]))
        print("  -v %s   %s" % (v, " | ".join(row)))
    print("worst rms over the tables: %.2e" % worst)
    target = next((a.split("=", 1)[1] if "=" in a else BOUNDS for a in sys.argv[1:] if a.startswith("--write")), None)
    if target:
        json.dump({"made_by": "tools/f32_survey.py --write", "flowgen": FLOWGEN, "utterances": 64, "samples": 16000, "seeds": "100..163",
                   "cases": res}, open(target, "w"), indent=1, sort_keys=True)
        print("wrote", target)
    # launch times of BASELINE config 3 in the three arithmetics, same plan
    specs, fs, dur, label = configs.config_specs(3, 65536)
    lanes, dd = vs.lanes_from_specs(specs)
    ns = vs.num_samples(fs, dd)
    pitch = vs.row_pitch(ns)
    plan = eng.plan(lanes, ns)
    out = eng.dev_alloc(65536 * pitch * 2)
    for arith, nm in ((vs.VS_ARITH_EXACT, "exact"), (vs.VS_ARITH_FMA, "fma"), (vs.VS_ARITH_F32, "f32")):
        eng.set_arith(arith)
        ts = []
        for r in range(12):
            t0 = time.perf_counter()
            plan.launch(vs.VS_KIND_SYNTH, out, out_pitch=pitch)
            eng.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        ts = sorted(ts[2:])
        print("config 3, %s: median %.3f ms, min %.3f ms per launch (%s)" % (nm, ts[len(ts) // 2], ts[0], plan.kernel_name(vs.VS_KIND_SYNTH)))
    got = eng.dev_download(out, (65536, pitch), np.int16)[:4096, :ns]
    pc, mx, rms = stats(got, po.synth([lanes[i] for i in range(4096)], ns, threads=32))
    print("config 3, f32, first 4096 utterances against the oracle: %.1f %% differ, max %d LSB, rms %.2e" % (pc, mx, rms))


if __name__ == "__main__":
    main()
