#!/usr/bin/env python3
"""Instruction mix of the loops of one kernel in a hipcc -S listing.
   tools/isa_loops.py <listing.s> <kernel symbol substring> [min instructions]
A loop = label ... backward branch to that label; for each: VALU / SALU / LDS / VMEM / other counts and the
most frequent opcodes.  Straight-line counts only (inner branches are not weighed)."""
import re, sys, collections
path, sym = sys.argv[1], sys.argv[2]
minlen = int(sys.argv[3]) if len(sys.argv) > 3 else 20
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\w+:', l) and sym in l)
body = []
for l in lines[start + 1:]:
    t = l.strip()
    if t.startswith('.end_amdhsa_kernel') or t.startswith('.Lfunc_end'): break
    if not t or t.startswith(';'): continue
    body.append(t.split(';')[0].strip())
labels = {}
ins = []
for t in body:
    m = re.match(r'^(\.LBB\w+):', t)
    if m: labels[m.group(1)] = len(ins); continue
    if t.startswith('.'): continue
    ins.append(t)
def cls(op):
    if op.startswith('v_'): return 'VALU'
    if op.startswith('ds_'): return 'LDS'
    if op.startswith('s_'): return 'SALU'
    if op.startswith('global_') or op.startswith('buffer_') or op.startswith('flat_') or op.startswith('scratch_'): return 'VMEM'
    return 'other'
loops = []
for i, t in enumerate(ins):
    m = re.match(r'^s_cbranch_\w+\s+(\.LBB\w+)|^s_branch\s+(\.LBB\w+)', t)
    if m:
        lab = m.group(1) or m.group(2)
        if lab in labels and labels[lab] <= i: loops.append((labels[lab], i, lab))
print("kernel:", lines[start].rstrip(':'), " instructions:", len(ins))
for a, b, lab in sorted(loops):
    if b - a + 1 < minlen: continue
    c = collections.Counter(cls(t.split()[0]) for t in ins[a:b + 1])
    ops = collections.Counter(re.sub(r'_e(32|64)$', '', t.split()[0]) for t in ins[a:b + 1])
    print("%-12s [%5d..%5d] %5d instr  VALU %4d SALU %4d LDS %3d VMEM %3d | %s" % (lab, a, b, b - a + 1, c['VALU'], c['SALU'], c['LDS'], c['VMEM'],
          ' '.join('%s:%d' % kv for kv in ops.most_common(9))))
