#!/usr/bin/env python3
"""Wall clock of the two drop-in programs against the reference's own, one utterance each:
   bin/flowgen_shimmer -o f.wav -r 16000 -d 1    vs    oracle/_ref/flowgen_shimmer (same argv)
   bin/vowel -i f.wav -o v.wav -v a              vs    oracle/_ref/vowel
20 runs each (first run reported separately: cold caches), then tools/startup_probe.c: the library calls behind the
programs, each timed.  Reference surface: flowgen_shimmer.c:222-241, vowel_new.c:195-234.
    python tools/cli_startup.py [runs]"""
import os
import statistics
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUNS = int(sys.argv[1]) if len(sys.argv) > 1 else 20


def timed(cmd, cwd, env):
    ts = []
    for _ in range(RUNS + 1):
        t0 = time.perf_counter()
        r = subprocess.run(cmd, cwd=cwd, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
        ts.append((time.perf_counter() - t0) * 1e3)
        if r.returncode != 0:
            raise RuntimeError("%s: rc %d %s" % (cmd, r.returncode, r.stderr.decode()[-300:]))
    return ts[0], ts[1:]


def main():
    env = dict(os.environ, VS_SEED="1", VS_WAV_HEADER="72")
    b = os.path.join(ROOT, "voice_synth_amd", "bin")
    ref = os.path.join(ROOT, "oracle", "_ref")
    rows = []
    with tempfile.TemporaryDirectory(prefix="vscli") as d:
        for name, exe_dir in (("ours", b), ("reference -O0", ref)):
            fg = os.path.join(exe_dir, "flowgen_shimmer")
            vw = os.path.join(exe_dir, "vowel")
            if not os.path.exists(fg):
                print("%s: %s is not built" % (name, fg))
                continue
            first, ts = timed([fg, "-o", "f.wav", "-r", "16000", "-d", "1"], d, env)
            rows.append((name, "flowgen_shimmer -o f.wav -r 16000 -d 1", first, ts))
            first, ts = timed([vw, "-i", "f.wav", "-o", "v.wav", "-v", "a"], d, env)
            rows.append((name, "vowel -i f.wav -o v.wav -v a", first, ts))
    print("%-14s %-42s %9s %9s %9s %9s   (ms, %d runs after the first)" % ("program", "command", "first", "min", "median", "max", RUNS))
    for name, cmd, first, ts in rows:
        print("%-14s %-42s %9.1f %9.1f %9.1f %9.1f" % (name, cmd, first, min(ts), statistics.median(ts), max(ts)))
    probe = os.path.join(tempfile.gettempdir(), "vs_startup_probe")
    lib = os.path.join(ROOT, "voice_synth_amd", "lib")
    subprocess.run(["gcc", "-O2", "-I" + os.path.join(ROOT, "include"), "-o", probe, os.path.join(ROOT, "tools", "startup_probe.c"),
                    "-L" + lib, "-lvoicesynth", "-lm", "-Wl,-rpath," + lib], check=True)
    hprobe = os.path.join(tempfile.gettempdir(), "vs_hip_startup_probe")
    subprocess.run(["gcc", "-O2", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-o", hprobe,
                    os.path.join(ROOT, "tools", "hip_startup_probe.c"), "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    for var in (None, None, "PROBE_SMALL_FIRST", "PROBE_PINNED_FIRST"):
        print("--- tools/hip_startup_probe.c (the HIP runtime alone, call by call)%s" % (" with %s=1" % var if var else ""))
        t0 = time.perf_counter()
        out = subprocess.run([hprobe], capture_output=True, text=True, env=dict(os.environ, **({var: "1"} if var else {})))
        print(out.stdout.rstrip())
        print("whole process                %8.2f ms" % ((time.perf_counter() - t0) * 1e3))
    t0 = time.perf_counter()
    subprocess.run(["/bin/true"])
    print("--- /bin/true (process start on this box) %8.2f ms" % ((time.perf_counter() - t0) * 1e3))
    for _ in range(2):
        print("--- tools/startup_probe.c (one process: the calls behind the programs)")
        t0 = time.perf_counter()
        out = subprocess.run([probe, "filter"], capture_output=True, text=True)
        print(out.stdout.rstrip())
        print("whole process            %8.2f ms" % ((time.perf_counter() - t0) * 1e3))


main()
