#!/usr/bin/env python3
"""Where the hot loops of a kernel sit in the code object: address, length in bytes and the phase of the loop head
relative to 64-byte lines (a wavefront alone on its SIMD is sensitive to it: profiles/r05_loop_alignment.txt).
    python tools/isa_align.py <device ELF or bundled object> [kernel symbol substring] [min bytes]"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def device_elf(path):
    head = open(path, "rb").read(4)
    if head == b"\x7fELF":
        probe = subprocess.run([LLVM + "/llvm-readelf", "-h", path], capture_output=True, text=True).stdout
        if "AMDGPU" in probe or "0xe0" in probe.lower():
            return path
        # a host object with the device code inside
    out = os.path.join(tempfile.gettempdir(), os.path.basename(path) + ".gfx950.elf")
    if head == b"\x7fELF":
        # a host object or shared library: the device code is a clang offload bundle in its .hip_fatbin section
        fat = os.path.join(tempfile.gettempdir(), os.path.basename(path) + ".hip_fatbin")
        subprocess.run([LLVM + "/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", path, fat], check=True)
        path = fat
    for typ in ("o", "a"):
        r = subprocess.run([LLVM + "/clang-offload-bundler", "--type=" + typ, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                            "--input=" + path, "--output=" + out, "--unbundle"], capture_output=True, text=True)
        if r.returncode == 0 and os.path.exists(out) and os.path.getsize(out) > 0:
            return out
    raise RuntimeError("no gfx950 code object in " + path)


def loops(elf, sym, min_bytes):
    dis = subprocess.run([LLVM + "/llvm-objdump", "-d", elf], capture_output=True, text=True).stdout.split("\n")
    cur, res, ins = None, [], {}
    for l in dis:
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", l)
        if m:
            cur = m.group(1)
            continue
        if cur is None or sym not in cur:
            continue
        m = re.match(r"\s+(\S.*?)\s+//\s+([0-9A-F]+):\s+([0-9A-F ]+)", l)
        if not m:
            continue
        addr, text = int(m.group(2), 16), m.group(1)
        mb = re.match(r"s_c?branch\w*\s+(\d+)", text)
        if mb:
            off = int(mb.group(1))
            if off >= 32768:
                off -= 65536
            tgt = addr + 4 + 4 * off
            if tgt < addr and addr - tgt >= min_bytes:
                res.append((cur, tgt, addr + 4 - tgt))
    return res


def superstep_alignment(path, sym):
    """(instructions, 8-byte instructions on an 8-byte boundary, 8-byte instructions off it, s_nop) of the filter's
    super-step loop of kernel `sym` (exact mangled name): the smallest loop that holds 24 samples' worth of fp64 arithmetic"""
    elf = device_elf(path)
    dis = subprocess.run([LLVM + "/llvm-objdump", "-d", elf, "--disassemble-symbols=" + sym], capture_output=True, text=True).stdout
    ins = []
    for l in dis.split("\n"):
        m = re.match(r"\s+(\S.*?)\s+//\s+([0-9A-F]+):\s+([0-9A-F ]+)", l)
        if m:
            ins.append((int(m.group(2), 16), m.group(1).split()[0], 4 * len(m.group(3).split())))
    best = None
    for _, head, size in loops(elf, sym, 4000):
        body = [x for x in ins if head <= x[0] < head + size]
        f64 = sum(1 for _, op, _ in body if op in ("v_mul_f64", "v_add_f64", "v_fma_f64", "v_fmac_f64_e32"))
        if f64 >= 24 * 22 and (best is None or size < best[0]):
            best = (size, body)
    if best is None:
        raise RuntimeError("no super-step loop found in " + sym)
    body = best[1]
    on = sum(1 for a, _, n in body if n == 8 and a % 8 == 0)
    off = sum(1 for a, _, n in body if n == 8 and a % 8)
    return len(body), on, off, sum(1 for _, op, _ in body if op == "s_nop")


def main():
    path = sys.argv[1]
    sym = sys.argv[2] if len(sys.argv) > 2 else "vs_synth_ws_kernel"
    min_bytes = int(sys.argv[3]) if len(sys.argv) > 3 else 4000
    for name, head, size in loops(device_elf(path), sym, min_bytes):
        print("%-62s loop head 0x%06x  %6d bytes  head %% 64 = %2d  head %% 32 = %2d" % (name, head, size, head % 64, head % 32))


if __name__ == "__main__":
    main()
