"""Very long utterances (millions of samples per row): the fused kernels, the staging limit of the host path."""
import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, voice_synth_amd as vs
from oracle import pyoracle as po
from voice_synth_amd import configs
specs, fs, dur, _ = configs.config_specs(3, 70)
lanes, d = vs.lanes_from_specs(specs)
eng = vs.Engine(0)
for n in (3_000_000, 8_388_607):
    sub = [lanes[i] for i in range(3 if n > 4_000_000 else 70)]
    got = eng.synth(sub, n)
    want = po.synth(sub, n, threads=32)
    print(n, len(sub), "lanes: equal", bool(np.array_equal(got, want)))
# one row longer than a staging block into pageable memory must be refused, into pinned memory not
lane=[lanes[0]]
got = eng.synth(lane, 9_000_000)   # one row is longer than a staging block: DMA by the runtime into pageable memory
print("pageable 9M equal:", bool(np.array_equal(got[0], po.synth(lane, 9_000_000)[0])))
v = eng.synth_pinned(lane, 9_000_000)
print("pinned 9M equal:", bool(np.array_equal(v[0], po.synth(lane, 9_000_000)[0])))
eng.host_free(v)
