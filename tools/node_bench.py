"""vs_node_synth_gather on ONE device with logical shards (the same device listed S times): what
the node-level C entry costs next to a plain launch, with the transfer path forced
(VS_NODE_STAGE_ALL: every shard synthesises into its chunk buffers and copies the chunks to their
rows of the root buffer -- device-to-device here, peer DMA over xGMI on a real node) and with /
without the overlap of copies and kernels.  Functional evidence for the pipeline, not an xGMI number."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import voice_synth_amd as vs
from voice_synth_amd import configs
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
specs, fs, dur, label = configs.config_specs(3, n)
lanes, d = vs.lanes_from_specs(specs)
ns = vs.num_samples(fs, d)
eng = vs.Engine(0)
root = eng.dev_alloc(n * ns * 2)
plan = eng.plan(lanes, ns)
for _ in range(2):
    plan.launch(vs.VS_KIND_SYNTH, root); eng.synchronize()
t0 = time.perf_counter(); plan.launch(vs.VS_KIND_SYNTH, root); eng.synchronize(); t1 = time.perf_counter()
print("%s: one plan, one launch: %.2f ms" % (label, (t1 - t0) * 1e3), flush=True)
ref = eng.dev_download(root, (n, ns))[::1021].copy()
for shards in (1, 2, 4, 8):
    node = vs.Node([0] * shards)
    for flags, name in ((vs.Node.OVERLAP, "in place (root device)"), (vs.Node.STAGE_ALL, "staged, copies after kernels"),
                        (vs.Node.STAGE_ALL | vs.Node.OVERLAP, "staged, copies behind kernels")):
        best = None
        for rep in range(3):
            tot, comp = node.synth_gather(lanes, ns, root, ns, flags)
            best = (tot, comp) if best is None or tot < best[0] else best
        ok = bool(np.array_equal(eng.dev_download(root, (n, ns))[::1021], ref))
        print("  %d shard(s), %-30s total %.2f ms (slowest shard's kernels %.2f ms)  %.1f Gsamples/s  equal=%s"
              % (shards, name, best[0], best[1], n * ns / best[0] / 1e6, ok), flush=True)
    node.close()
