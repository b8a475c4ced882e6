"""Which sources a measurement belongs to.

PMC passes (profiles/pmc_traffic.json, profiles/pmc_valu.json) are collected in their own runs, so the
figures bench.py copies out of them are only true of the kernel that was shipped when they were taken.
Two stamps tie them to the tree:

  kernel_sources_sha16  sha256 (first 16 hex digits) over what determines the device code AND how it is launched:
                        csrc/vs_kernels.hip, vs_dev_*.h, vs_device.h, vs_tables.h (the kernels), vs_api.c,
                        vs_planhost.[ch], vs_internal.h (the plan: ring depth, roles, thresholds), include/voice_synth.h
                        and the Makefile's HIPFLAGS -- with comments and white space stripped, so that a
                        comment-only edit does not orphan the PMC passes (round 4 had to re-take them for one).
                        Computable on the GPU box (the snapshot carries no .git), so it is what bench.py
                        compares at run time.  all_sources_sha16: the raw bytes of everything under csrc/ and
                        include/, informational.
  profile_head          `git log -1 --format=%h -- voice_synth_amd/csrc include` (+ "-dirty") of the tree
                        the pass ran on.  Taken HERE, before the snapshot leaves (tools/gpurun.sh writes
                        build/git_head.txt, which travels), because the box has no git history.
"""
import hashlib
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOURCE_DIRS = ("voice_synth_amd/csrc", "include")
SOURCE_EXT = (".hip", ".h", ".c")


KERNEL_FILES = ("voice_synth_amd/csrc/vs_kernels.hip", "voice_synth_amd/csrc/vs_dev_primitives.h",
                "voice_synth_amd/csrc/vs_dev_generator.h", "voice_synth_amd/csrc/vs_dev_filter.h",
                "voice_synth_amd/csrc/vs_device.h", "voice_synth_amd/csrc/vs_tables.h", "voice_synth_amd/csrc/vs_api.c",
                "voice_synth_amd/csrc/vs_planhost.c", "voice_synth_amd/csrc/vs_planhost.h",
                "voice_synth_amd/csrc/vs_internal.h", "include/voice_synth.h")


def strip_comments(text):
    """C / C++ source without comments and with runs of white space collapsed (string and character
    literals are left alone)"""
    pat = re.compile(r'//[^\n]*|/\*.*?\*/|"(?:\\.|[^"\\])*"|\'(?:\\.|[^\'\\])*\'', re.S)
    text = pat.sub(lambda m: " " if m.group(0).startswith("/") else m.group(0), text)
    return " ".join(text.split())


def hipflags(root=ROOT):
    try:
        for line in open(os.path.join(root, "Makefile")):
            if line.startswith("HIPFLAGS"):
                return " ".join(line.split(":=", 1)[1].split())
    except OSError:
        pass
    return ""


def kernel_sources_sha16(root=ROOT):
    h = hashlib.sha256()
    for rel in KERNEL_FILES:
        path = os.path.join(root, rel)
        if not os.path.exists(path):
            continue
        h.update(rel.encode() + b"\0")
        h.update(strip_comments(open(path, encoding="utf-8", errors="replace").read()).encode())
        h.update(b"\0")
    h.update(hipflags(root).encode())
    return h.hexdigest()[:16]


def all_sources_sha16(root=ROOT):
    h = hashlib.sha256()
    for d in SOURCE_DIRS:
        base = os.path.join(root, d)
        for name in sorted(os.listdir(base)):
            if name.endswith(SOURCE_EXT):
                h.update(name.encode() + b"\0")
                h.update(open(os.path.join(base, name), "rb").read())
                h.update(b"\0")
    return h.hexdigest()[:16]


def git_head(root=ROOT):
    """hash of the last commit that touched the kernel sources, "-dirty" appended when they differ from
    it; from git where there is a history, else from the stamp tools/gpurun.sh left; None if neither"""
    if os.path.isdir(os.path.join(root, ".git")):
        try:
            h = subprocess.run(["git", "log", "-1", "--format=%h", "--"] + list(SOURCE_DIRS), cwd=root, capture_output=True,
                               text=True, timeout=20).stdout.strip()
            dirty = subprocess.run(["git", "status", "--porcelain", "--"] + list(SOURCE_DIRS), cwd=root, capture_output=True,
                                   text=True, timeout=20).stdout.strip()
            if h:
                return h + ("-dirty" if dirty else "")
        except Exception:
            pass
    try:
        rec = open(os.path.join(root, "build", "git_head.txt")).read().split()
        # only believe the stamp if it was taken of THESE sources
        if len(rec) >= 2 and rec[1] == kernel_sources_sha16(root):
            return rec[0]
    except OSError:
        pass
    return None


def stamp():
    return {"profile_head": git_head(), "kernel_sources_sha16": kernel_sources_sha16(), "all_sources_sha16": all_sources_sha16()}


if __name__ == "__main__":
    # tools/gpurun.sh: leave the stamp where the snapshot takes it along
    os.makedirs(os.path.join(ROOT, "build"), exist_ok=True)
    head = git_head() or "unknown"
    with open(os.path.join(ROOT, "build", "git_head.txt"), "w") as f:
        f.write("%s %s\n" % (head, kernel_sources_sha16()))
    print(head, kernel_sources_sha16())
