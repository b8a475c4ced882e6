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
# the same with the vowel stage's own noise (vowel -n 20): 3750 frames per row -- on every row (the fused kernel takes the frame
# powers along, vs_synth_ws_pow_kernel) and on every other row (the streaming pass does the rest)
for every in (1, 2):
    specs_n, _, _, _ = configs.config_specs(3, 70, out_noise_db=20.0)
    ln, _ = vs.lanes_from_specs(specs_n)
    for i in range(70):
        if i % every:
            ln[i].out_snr = 0.0
    n = 3_000_000
    plan = eng.plan(ln, n)
    name = plan.kernel_name(vs.VS_KIND_SYNTH)
    plan.close()
    got = eng.synth(ln, n)
    want = po.synth(ln, n, threads=32)
    print(n, "vowel -n on every %d. row, %s: equal" % (every, name), bool(np.array_equal(got, want)))
# one row longer than a staging block into pageable memory must be refused, into pinned memory not
lane=[lanes[0]]
got = eng.synth(lane, 9_000_000)   # one row is longer than a staging block: DMA by the runtime into pageable memory
print("pageable 9M equal:", bool(np.array_equal(got[0], po.synth(lane, 9_000_000)[0])))
v = eng.synth_pinned(lane, 9_000_000)
print("pinned 9M equal:", bool(np.array_equal(v[0], po.synth(lane, 9_000_000)[0])))
eng.host_free(v)
