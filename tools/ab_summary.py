import re,collections
d=collections.defaultdict(lambda: collections.defaultdict(list))
cur=None
for l in open("gpurun_out/ab.log"):
    m=re.match(r"== config (\d+) rep (\d+) (\S+)",l)
    if m: cur=(m.group(1),m.group(3)); continue
    m=re.match(r"(exact|fma)/synth: ([\d.]+) ms",l)
    if m: d[cur][m.group(1)].append(float(m.group(2)))
for v in sorted(d):
    print("config %s %-8s"%v, "  ".join("%s min %.3f med %.3f"%(a,min(d[v][a]),sorted(d[v][a])[len(d[v][a])//2]) for a in ("exact","fma")))
