"""Basic-block census of one kernel in a hipcc -S listing: VALU / SALU / LDS / VMEM counts per block and
the branch targets, to read hot-loop instruction counts off the ISA.

  hipcc $(HIPFLAGS) -S --cuda-device-only -o /tmp/vs.s voice_synth_amd/csrc/vs_kernels.hip
  python tools/isa_blocks.py /tmp/vs.s _Z18vs_synth_ws_kernelILi0ELb1EEv12VsKernelArgs [min_valu]
"""
import re
import sys


def main():
    path, sym = sys.argv[1], sys.argv[2]
    min_valu = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(sym + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith("\t.end_amdhsa_kernel") or lines[i].startswith(".Lfunc_end"))
    blocks, cur = [], {"name": "entry", "ins": []}
    for l in lines[start + 1:end]:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append(cur)
            cur = {"name": m.group(1), "ins": []}
            continue
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."):
            continue
        cur["ins"].append(t.split(";")[0].strip())
    blocks.append(cur)
    order = {b["name"]: i for i, b in enumerate(blocks)}
    for i, b in enumerate(blocks):
        ops = [x.split()[0] for x in b["ins"]]
        valu = [o for o in ops if o.startswith("v_")]
        if len(valu) < min_valu:
            continue
        tg = [x.split()[-1] for x in b["ins"] if x.startswith("s_cbranch") or x.startswith("s_branch")]
        back = [t for t in tg if t in order and order[t] <= i]
        mix = {}
        for o in valu:
            k = ("f64" if "f64" in o else "mad_u64" if "mad_u64" in o else "bitop" if "bitop" in o else
                 "cndmask" if "cndmask" in o else "cmp" if "cmp" in o else "mov" if "mov" in o or "accvgpr" in o else "other")
            mix[k] = mix.get(k, 0) + 1
        lds = sum(1 for o in ops if o.startswith("ds_"))
        vm = sum(1 for o in ops if o.startswith("global_") or o.startswith("buffer_"))
        sal = sum(1 for o in ops if o.startswith("s_"))
        print("%-12s valu %4d salu %4d lds %3d vmem %2d  %s  -> %s%s" % (
            b["name"], len(valu), sal, lds, vm, mix, ",".join(tg), "  LOOP" if back else ""))


main()
