"""Turns rocprofv3 --pmc passes of `bench.py --no-cpu-baseline` into the two small JSON files
bench.py reads:

  profiles/pmc_traffic.json  (roofline.traffic)  from two SEPARATE passes, FETCH_SIZE and
      WRITE_SIZE, as /opt/skills/guides/MI355X_MICROARCH.md section HBM prescribes.  Corrections
      applied (same guide): counters are in KiB (x1024); on gfx950 FETCH_SIZE reports half of the
      bytes of a read stream -> doubled; WRITE_SIZE is exact for 16-byte-per-lane stores, which is
      what the filter super-step issues.
  profiles/pmc_valu.json     (roofline.valu)     from one or two SQ passes: wave-instructions
      by class, active / wait quad-cycles, per launch of the dominant kernel.

usage: summarize_pmc.py traffic <fetch.csv> <write.csv> <key> [kernel substring]
       summarize_pmc.py valu <sq1.csv> [<sq2.csv> ...] <key> [--kernel substring]"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.provenance import stamp  # noqa: E402  (which tree the pass ran on: profile_head + kernel_sources_sha16)


def per_kernel(path):
    """{kernel: {counter: [values per dispatch]}} and {kernel: [durations ns]}"""
    vals, dur = {}, {}
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        vals.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        try:
            dur.setdefault(k, {})[r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        except (KeyError, ValueError):
            pass
    return vals, dur


def pick(vals, counter, want):
    ks = [k for k in vals if counter in vals[k] and (want in k if want else k.startswith("void vs_synth"))]
    ks.sort(key=lambda k: -len(vals[k][counter]))
    return ks[0]


def main():
    mode = sys.argv[1]
    if mode == "traffic":
        fetch_csv, write_csv, key = sys.argv[2], sys.argv[3], sys.argv[4]
        want = sys.argv[5] if len(sys.argv) > 5 else ""
        out_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        data = json.load(open(out_path)) if os.path.exists(out_path) else {}
        f, _ = per_kernel(fetch_csv)
        w, _ = per_kernel(write_csv)
        k = pick(w, "WRITE_SIZE", want)
        fv, wv = f[k]["FETCH_SIZE"], w[k]["WRITE_SIZE"]
        fetch = sum(fv) / len(fv) * 1024 * 2
        write = sum(wv) / len(wv) * 1024
        data[key] = {
            "kernel": k,
            "launches_averaged": len(wv),
            "FETCH_SIZE_KiB_raw": sum(fv) / len(fv),
            "WRITE_SIZE_KiB_raw": sum(wv) / len(wv),
            "fetch_bytes_corrected_x2": fetch,
            "write_bytes": write,
            "hbm_bytes_per_launch": fetch + write,
        }
        data[key].update(stamp())
    elif mode == "valu":
        args = sys.argv[2:]
        want = ""
        if "--kernel" in args:
            i = args.index("--kernel")
            want = args[i + 1]
            del args[i:i + 2]
        csvs, key = args[:-1], args[-1]
        out_path = os.path.join(ROOT, "profiles", "pmc_valu.json")
        data = json.load(open(out_path)) if os.path.exists(out_path) else {}
        rec = {}
        for path in csvs:
            v, d = per_kernel(path)
            cands = [x for x in v if (want in x if want else x.startswith("void vs_synth"))]
            cands.sort(key=lambda x: -max(len(c) for c in v[x].values()))
            k = cands[0]
            rec["kernel"] = k
            for counter, xs in v[k].items():
                rec[counter + "_per_launch"] = sum(xs) / len(xs)
            if k in d and d[k]:
                ds = sorted(d[k].values())
                rec["kernel_ns_under_profiler_median"] = ds[len(ds) // 2]
            rec["launches_averaged"] = len(next(iter(v[k].values())))
        # effective clock (MI355X_MICROARCH.md, DVFS): GRBM_GUI_ACTIVE is summed over the 8 XCDs
        if "GRBM_GUI_ACTIVE_per_launch" in rec and "kernel_ns_under_profiler_median" in rec:
            rec["clock_GHz_under_profiler"] = rec["GRBM_GUI_ACTIVE_per_launch"] / 8.0 / rec["kernel_ns_under_profiler_median"]
        rec.update(stamp())
        data[key] = rec
    else:
        sys.exit(__doc__)
    json.dump(data, open(out_path, "w"), indent=1)
    print(key, json.dumps(data[key]))


main()
