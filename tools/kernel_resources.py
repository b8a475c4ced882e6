#!/usr/bin/env python3
"""VGPRs / SGPRs / scratch / occupancy of every kernel of csrc/vs_kernels.hip, from the compiler's own remarks
(-Rpass-analysis=kernel-resource-usage with the SHIPPED flags: `make resources`).  One line per kernel; used by
tests/test_kernel_resources.py (the guard on the three-role kernel's 168 registers) and to write
profiles/rNN_kernel_resources.txt.

    python tools/kernel_resources.py [extra hipcc flags ...]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def hipflags():
    """the HIPFLAGS line of the Makefile: what the shipped object is compiled with"""
    for line in open(os.path.join(ROOT, "Makefile")):
        if line.startswith("HIPFLAGS"):
            return line.split(":=", 1)[1].replace("$(ARCH)", "gfx950").split()
    raise RuntimeError("HIPFLAGS not found in the Makefile")


def demangle(names):
    try:
        out = subprocess.run(["c++filt"] + names, capture_output=True, text=True, check=True).stdout
        return [n.replace("(VsKernelArgs)", "").replace("void ", "").strip() for n in out.splitlines()]
    except Exception:
        return names


def resources(extra=()):
    """[{name, vgprs, agprs, sgprs, scratch, occupancy}] in the order the compiler reports them"""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    src = os.path.join(ROOT, "voice_synth_amd", "csrc", "vs_kernels.hip")
    cmd = [hipcc] + hipflags() + list(extra) + ["-Rpass-analysis=kernel-resource-usage", "-c", "-o", os.devnull, src]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed: " + r.stderr[-2000:])
    recs, cur = [], None
    for line in r.stderr.splitlines():
        m = re.search(r"remark:\s+(.*?)\s*\[-Rpass-analysis", line)
        if not m:
            continue
        body = m.group(1)
        if body.startswith("Function Name:"):
            cur = {"mangled": body.split(":", 1)[1].strip()}
            recs.append(cur)
        elif cur is not None and ":" in body:
            k, v = (x.strip() for x in body.split(":", 1))
            key = {"VGPRs": "vgprs", "AGPRs": "agprs", "TotalSGPRs": "sgprs", "ScratchSize [bytes/lane]": "scratch",
                   "Occupancy [waves/SIMD]": "occupancy", "LDS Size [bytes/block]": "lds_static",
                   "VGPRs Spill": "vgpr_spill", "SGPRs Spill": "sgpr_spill"}.get(k)
            if key:
                cur[key] = int(v)
    for rec, name in zip(recs, demangle([r_["mangled"] for r_ in recs])):
        rec["name"] = name
    return recs


def main():
    recs = resources(sys.argv[1:])
    print("# hipcc %s %s" % (" ".join(hipflags()), " ".join(sys.argv[1:])))
    print("%-58s %6s %6s %6s %8s %10s" % ("kernel", "VGPRs", "AGPRs", "SGPRs", "scratch", "occupancy"))
    for r in recs:
        print("%-58s %6d %6d %6d %8d %10d" % (r["name"], r.get("vgprs", -1), r.get("agprs", 0), r.get("sgprs", -1),
                                               r.get("scratch", -1), r.get("occupancy", -1)))


if __name__ == "__main__":
    main()
