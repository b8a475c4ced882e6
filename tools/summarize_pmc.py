"""Turns the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as
/opt/skills/guides/MI355X_MICROARCH.md section HBM prescribes) of `bench.py --no-cpu-baseline`
into profiles/pmc_traffic.json, which bench.py reads for roofline.traffic.

Corrections applied (same guide): counters are in KiB (x1024); on gfx950 FETCH_SIZE reports
half of the bytes of a read stream -> doubled; WRITE_SIZE is exact for 16-byte-per-lane stores,
which is what the filter super-step issues."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel(path, counter):
    vals = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        vals.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return vals


def main():
    fetch_csv, write_csv, key = sys.argv[1], sys.argv[2], sys.argv[3]
    out_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    data = json.load(open(out_path)) if os.path.exists(out_path) else {}
    f = per_kernel(fetch_csv, "FETCH_SIZE")
    w = per_kernel(write_csv, "WRITE_SIZE")
    kern = [k for k in w if "vs_synth_kernel<0, 0" in k or "vs_synth_kernel<1, 0" in k]
    kern.sort(key=lambda k: -len(w[k]))
    k = kern[0]
    fetch = sum(f[k]) / len(f[k]) * 1024 * 2
    write = sum(w[k]) / len(w[k]) * 1024
    data[key] = {
        "kernel": k,
        "launches_averaged": len(w[k]),
        "FETCH_SIZE_KiB_raw": sum(f[k]) / len(f[k]),
        "WRITE_SIZE_KiB_raw": sum(w[k]) / len(w[k]),
        "fetch_bytes_corrected_x2": fetch,
        "write_bytes": write,
        "hbm_bytes_per_launch": fetch + write,
    }
    json.dump(data, open(out_path, "w"), indent=1)
    print(key, json.dumps(data[key]))


main()
