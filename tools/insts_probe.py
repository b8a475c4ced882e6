"""Dynamic instruction counts of the fused kernels: wave-instructions per sample of a 64-utterance group (vector / scalar /
LDS) from one rocprofv3 --pmc pass over tools/quick_bench.py -- the figure every kernel change is judged by (the launch
time of a chip whose vector pipe is 95 % busy follows it).
    tools/insts_probe.py <counter_collection.csv> [lanes] [samples]     prints one line per kernel
Run on the GPU box as:
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d <dir> -o p -- python3 tools/quick_bench.py 3 65536 3"""
import csv, sys, collections
path = sys.argv[1]
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
ns = int(sys.argv[3]) if len(sys.argv) > 3 else 16000
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(path)):
    acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
units = lanes * ns / 64.0
for k in sorted(acc):
    if "vs_synth" not in k and "vs_out" not in k:
        continue
    c = acc[k]
    def per(name):
        v = c.get(name, [])
        return sum(v) / len(v) / units if v else float("nan")
    print("%-60s launches %3d  VALU %7.2f  SALU %6.2f  LDS %5.2f  per sample" % (k.replace("void ", "").replace("(VsKernelArgs)", "")[:60],
          len(c.get("SQ_INSTS_VALU", [])), per("SQ_INSTS_VALU"), per("SQ_INSTS_SALU"), per("SQ_INSTS_LDS")))
