#!/usr/bin/env python3
"""bench.py -- throughput of the fused source->filter hot path on N MI355X of one node.

    python bench.py --gpus N --steps K --warmup W          (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one launch of the fused kernel over one batch of synthetic utterances already
described in HBM (lane records + cos rows; the int16 PCM is written to HBM).  Workload at every
N: BASELINE.json configs[2] per GPU -- 65536 utterances, mixed vowels 1/2/4/6/7, 16 kHz, 1 s,
jitter 1 % + shimmer 0.5 dB (-s 5.76) + glottal noise 20 dB -- i.e. weak scaling, lane keys
derived from the GLOBAL lane index.  (configs[1], batch 1024, is 16 wavefronts on a 1024-SIMD
chip and is a parity case, not a throughput case; the north star's target is stated for
>= 1e4 concurrent utterances.)

Rank 0 prints ONE JSON line.  Besides the driver's keys it carries
  roofline     : algorithmic 2 B/sample over the kernel's mean duration (HIP events on the
                 launch stream, inside the timed region) against the 8 TB/s HBM peak; the
                 fp64-VALU view of the same kernel (the path is VALU-issue-bound);
                 `valu`: wave-instructions per sample and issue rate from the SQ counters
                 committed under profiles/ for this kernel;
  plan         : host cost of vs_plan_create for the batch (outside every timed region);
  cpu_baseline : the CPU oracle (kind "port", OpenMP over lanes) on this box's host cores, on a
                 bounded sample of the same workload, and next to it the REFERENCE as shipped
                 (oracle/_ref, -O0 as its Makefile builds it, and -O2; one process pair per
                 utterance through .wav files, all host cores, driven by oracle/ref_pipelines) --
                 rank 0, N = 1 only;
  sustained    : the same plan launched back to back for >= 2 s (outside the timed region), HIP events on the
                 first and the last launch: the figure a compute-bound kernel holds once the chip has settled
                 on its clock, next to the 20-step burst above (`ratio_to_timed_region`);
  fresh_batches: N = 1 only -- a plan per batch of NEW utterances (new seeds) + its launch, the next plan made on the host
                 while this batch's kernel runs: what the product path around the kernel costs a caller who does not
                 launch the same plan twice;
  other_arith  : the other arithmetic contract (fma when the run is exact), HIP events around each of
                 10 launches after 3 warm-ups, outside the timed region;
  row_pitch_ab : the PCM buffer is the caller's, [utterance][row pitch]; the timed region writes rows at the pitch
                 vs_row_pitch() names (16064 samples for rows of 16000; --dense-rows: rows only rounded up to 16 bytes),
                 and this block launches the same plan with both pitches in turn -- both arithmetic modes, 10 launches
                 with HIP events each, outside the timed region: what the caller's choice is worth
                 (profiles/r05_row_pitch.txt);
  config4      : N > 1 only -- BASELINE.json's configuration for the node: 262144 utterances x 44100
                 samples (22.05 kHz, 2 s) cut over the N ranks; `value` (PCM left sharded),
                 `roofline_per_gpu`, and `value_with_gather` / `gather`: the same synthesis in chunks of
                 16384 utterances with every finished chunk travelling to rank 0 over RCCL while the next
                 one is being synthesised (voice_synth_amd/dist.py::PipelinedGather), timed end to end.
                 The top-level `value` stays config 3 per GPU (weak scaling), so that the N = 1 point of a
                 scaling curve is the single-GPU bench line.
VS_BENCH_REHEARSAL=1|2 (tests only): all ranks share device 0 and talk over gloo, so that the N > 1
control flow -- and with 2 also the pipelined gather leg, end to end on device tensors -- can be run on
a one-GPU box (RCCL refuses two ranks on one device); the numbers of such a run mean nothing.
The only use of oracle/ is inside cpu_baseline(): the CPU port and the compiled reference are
timed there, and the port's rows are compared with the rows the GPU produced in the timed region --
the whole batch (rms_vs_c_ref.rows_checked), since the CPU sample starts at lane 0 and outlasts it.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s
FP64_VALU_PEAK_TFLOPS = 78.6    # 256 CU x 4 SIMD x 16 lanes/clk x 2 flop x 2.4 GHz
ALGO_BYTES_PER_SAMPLE = 2       # one int16 store; the flow never reaches HBM
FLOP_PER_SAMPLE = 48            # 22 mul + 22 sub + gain + pre-emphasis mul/sub
GATHER_CHUNK = 16384            # utterances per chunk of the pipelined gather
SUSTAINED_S = 2.0                # back-to-back launches of the same plan for at least this long (`sustained`)
# the config-4 block runs under a watchdog that knows its phases: a phase that shows no progress for this long is a
# stalled exchange (bytes moved scale the allowance: 60 s + 1 s per GB into rank 0)
PHASE_DEADLINE_S = int(os.environ.get("VS_BENCH_PHASE_DEADLINE_S", "60"))   # (the variable: tests of the watchdog only)
TEARDOWN_DEADLINE_S = 60        # nor may the closing barrier hold the process once the line is printed


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=3, help="BASELINE.json configuration (1-based), default 3")
    ap.add_argument("--lanes", type=int, default=0, help="utterances per GPU (default: the configuration's batch)")
    ap.add_argument("--arith", choices=["exact", "fma", "f32"], default="exact",
                    help="exact (default: the reference's rounding sequence); fma and f32 are the opt-in tolerance modes")
    ap.add_argument("--vowel-n", type=float, default=None, metavar="DB",
                    help="every utterance also asks the vowel stage for its own noise (vowel -n DB): not a BASELINE workload, for profiles")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dense-rows", action="store_true", help="PCM rows only rounded up to 16 bytes instead of the pitch vs_row_pitch() names")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="N = 1: skip the configs 2 / 4-shard / 5 launches behind the timed region")
    ap.add_argument("--no-config4", action="store_true", help="N > 1: skip the config-4 block (262144 x 44100 over the ranks)")
    ap.add_argument("--config4-lanes", type=int, default=0, help="N > 1: utterances of the config-4 block (default 262144; tests)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="target CPU-baseline run time")
    return ap.parse_args()


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _cpu_quota():
    """CPU share of this container in cores (cgroup v2 cpu.max), None when unlimited or unknown"""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if quota == "max" else round(float(quota) / float(period), 2)
    except Exception:
        return None


def reference_as_shipped(specs_fn, n_samples, max_workers, target_s=6.0):
    """oracle/_ref (the reference's own two programs, built -O0 as its Makefile does and -O2, Philox
    random() shim) over the bench workload: one process pair per utterance through .wav files,
    `workers` worker processes -- what `xargs -P $(nproc)` does with the reference as it ships.
    Run by oracle/ref_pipelines (C, posix_spawn; no interpreter in the loop): a calibration run, then
    at least 4096 pipelines sized for about target_s seconds."""
    import subprocess
    import tempfile

    helper = os.path.join(ROOT, "oracle", "ref_pipelines")
    refdir = os.path.join(ROOT, "oracle", "_ref")
    out = {"cpu_model": _cpu_model(), "cpu_quota_cores": _cpu_quota(),
           "how": "oracle/ref_pipelines: per utterance `flowgen_shimmer -o f.wav ...; vowel -i f.wav -o v.wav ...` "
                  "(posix_spawn, one scratch directory per worker), process start and file I/O included"}
    if not os.path.exists(helper):
        out["error"] = "oracle/ref_pipelines is not built"
        return out

    def run(n, suffix, scratch, workers):
        specs = specs_fn(n)
        mf = os.path.join(scratch, "manifest.txt")
        with open(mf, "w") as f:
            for fa, va, seed in specs:
                f.write("%d|%s|%s\n" % (seed, " ".join(fa), " ".join(va)))
        r = subprocess.run([helper, os.path.join(refdir, "flowgen_shimmer" + suffix), os.path.join(refdir, "vowel" + suffix),
                            mf, str(workers), scratch], capture_output=True, text=True, timeout=600)
        rec = json.loads(r.stdout.strip().splitlines()[-1])
        if r.returncode != 0 or rec["failed"]:
            raise RuntimeError("ref_pipelines: rc %d, %s %s" % (r.returncode, r.stdout.strip(), r.stderr.strip()[:200]))
        return rec

    for suffix, key in (("", "O0_as_shipped"), ("_O2", "O2")):
        if not os.path.exists(os.path.join(refdir, "vowel" + suffix)):
            out[key] = {"error": "oracle/_ref/vowel%s is not built" % suffix}
            continue
        try:
            with tempfile.TemporaryDirectory(prefix="vsrp") as scratch:
                # how many workers this box can feed: the affinity mask of a container says little about
                # its CPU share (a one-GPU box reports 256 CPUs and peaks at 16 workers), so take the
                # best of a short sweep
                sweep = {}
                w = max_workers
                while w >= 4:
                    # a first look sizes the calibration: every worker count is then measured over at least
                    # 2000 pipelines AND about a second (tens of milliseconds of 8 x w pipelines chose the count
                    # from noise: 3478 / 2443 / 7335 per second for 8 / 16 / 32 workers in one round-3 run)
                    quick = run(8 * w, suffix, scratch, w)
                    n_cal = int(max(2000, min(60000, 1.0 * quick["pipelines"] / quick["seconds"])))
                    cal = run(n_cal, suffix, scratch, w)
                    sweep[w] = round(cal["pipelines"] / cal["seconds"], 1)
                    w //= 2
                workers = max(sweep, key=sweep.get)
                # the measurement itself in three equal segments, so that the line carries its own spread
                n = int(max(4096, min(400000, target_s * sweep[workers])))
                seg = [run((n + 2) // 3, suffix, scratch, workers) for _ in range(3)]
            rec = {k_: sum(r_[k_] for r_ in seg) for k_ in ("pipelines", "seconds", "process_seconds_flowgen", "process_seconds_vowel")}
            seg_rates = sorted(r_["pipelines"] / r_["seconds"] * n_samples / 1e6 for r_ in seg)
            per = rec["pipelines"] / rec["seconds"]
            out[key] = {"value": round(per * n_samples / 1e6, 2), "unit": "Msamples/s", "workers": workers,
                        "pipelines": rec["pipelines"], "seconds": round(rec["seconds"], 2),
                        "pipelines_per_s": round(per, 1),
                        "segments_Msamples/s": {"min": round(seg_rates[0], 2), "median": round(seg_rates[1], 2), "max": round(seg_rates[2], 2)},
                        "ms_per_pipeline_per_worker": round(1e3 * workers / per, 3),
                        "ms_in_flowgen": round(1e3 * rec["process_seconds_flowgen"] / rec["pipelines"], 3),
                        "ms_in_vowel": round(1e3 * rec["process_seconds_vowel"] / rec["pipelines"], 3),
                        "pipelines_per_s_by_workers": {str(k): v for k, v in sorted(sweep.items())},
                        "calibration": ">= 2000 pipelines and >= ~1 s per worker count"}
        except Exception as exc:  # pragma: no cover - e.g. a process limit of the box
            out[key] = {"error": "%s: %s" % (type(exc).__name__, exc)}
    return out


def cpu_baseline(specs_fn, n_samples, target_s, gpu_first_lanes=None, tolerance_rows=None):
    """The CPU oracle on a bounded sample of the same workload, all host cores.  The sample
    starts at lane 0, so its first rows double as a parity spot check of what the GPU just
    produced (the only place bench.py touches oracle/)."""
    import numpy as np
    import voice_synth_amd as vs
    from oracle import pyoracle as po

    host_cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(host_cores, po.max_threads()))
    # warm-up (library load, start of the thread team), then calibrate on growing samples until one
    # takes long enough to be a fair estimate
    lanes, _ = vs.lanes_from_specs(specs_fn(cores))
    po.synth(lanes, n_samples, threads=cores)
    lanes_cal = 4 * cores
    rate = 0.0
    for _ in range(5):
        lanes, _ = vs.lanes_from_specs(specs_fn(lanes_cal))
        t0 = time.perf_counter()
        po.synth(lanes, n_samples, threads=cores)
        t_cal = time.perf_counter() - t0
        rate = lanes_cal * n_samples / t_cal
        if t_cal >= 0.3:
            break
        lanes_cal *= 8
    n_lanes = int(min(262144, max(lanes_cal, target_s * rate / n_samples)))
    n_lanes = (n_lanes // cores) * cores or cores
    lanes, _ = vs.lanes_from_specs(specs_fn(n_lanes))
    t0 = time.perf_counter()
    pcm = po.synth(lanes, n_samples, threads=cores)
    t = time.perf_counter() - t0
    out = {
        "value": round(n_lanes * n_samples / t / 1e6, 2),
        "unit": "Msamples/s",
        "cores": cores,
        "cpu_quota_cores": _cpu_quota(),
        "kind": "port",
        "cpu_model": _cpu_model(),
        "sample": "%d utterances x %d samples of the same workload (first lanes), %.1f s, OpenMP over lanes"
                  % (n_lanes, n_samples, t),
    }
    if po.have_reference():
        # the reference AS SHIPPED: its Makefile passes no -O flag (-O0); -O2 next to it.  The GPU box
        # allows 1024 processes per command: a worker and its one child each.
        workers = max(1, min(host_cores, 256))
        shipped = reference_as_shipped(specs_fn, n_samples, workers)
        # the compiled reference and the port agree (a handful of utterances through the Python harness)
        try:
            few = specs_fn(4)
            shipped["matches_port"] = bool(all(
                np.array_equal(po.run_reference(fa, va, seed)["pcm"], pcm[i]) for i, (fa, va, seed) in enumerate(few)))
        except Exception as exc:  # pragma: no cover
            shipped["matches_port"] = "%s: %s" % (type(exc).__name__, exc)
        out["reference_as_shipped"] = shipped
    if gpu_first_lanes is not None:
        # every GPU row the CPU sample covers (the whole batch when the sample is at least as large)
        k = min(len(gpu_first_lanes), n_lanes)
        out["gpu_rows_checked"] = k
        mism = 0
        sq = 0.0
        for lo in range(0, k, 4096):   # in blocks: a float64 copy of the whole batch would be 17 GB
            a = gpu_first_lanes[lo:min(k, lo + 4096)]
            b = pcm[lo:min(k, lo + 4096)]
            ne = a != b
            if ne.any():
                mism += int(ne.sum())
                dd = a[ne].astype(np.float64) - b[ne].astype(np.float64)
                sq += float((dd * dd).sum())
        out["gpu_mismatched_samples"] = mism
        # BASELINE.json's second figure: RMS error against the C reference path for identical
        # seeds, in int16 LSB and on the /32768 scale (north star: <= 1e-5 normalised)
        rms = (sq / (float(k) * n_samples)) ** 0.5 if k else 0.0
        out["gpu_rms_error_lsb"] = rms
        out["gpu_rms_error_normalised"] = rms / 32768.0
    if tolerance_rows:
        # the opt-in arithmetics (fma, f32): first rows of their launches against the same CPU rows
        out["tolerance_modes"] = {}
        for name, rows in tolerance_rows.items():
            k = min(len(rows), n_lanes)
            d = rows[:k].astype(np.int32) - pcm[:k].astype(np.int32)
            r = float(np.sqrt(np.mean(d.astype(np.float64) ** 2))) if k else 0.0
            out["tolerance_modes"][name] = {"rows_checked": k, "differing_samples": int(np.count_nonzero(d)),
                                            "max_abs_lsb": int(np.abs(d).max()) if k else 0, "lsb": r, "normalised": r / 32768.0}
    return out


def fresh_batches(eng, dev, stream, lanes, n_samples, out, pitch, per_gpu, batches=12):
    """`batches` batches of NEW utterances (the timed workload with other seeds): vs_plan_create + launch per batch, the
    next batch's plan made on the host while this batch's kernel runs; wall clock over all of them, HIP events around it."""
    import numpy as np
    import torch

    import voice_synth_amd as vs

    # the batches' utterance descriptions are the CALLER's input and exist before the clock starts: `batches` + 2 copies
    # of the lane array, each with seeds nobody has synthesised yet (27 MB each)
    descr = []
    for b in range(batches + 2):
        mine = (vs.Lane * per_gpu).from_buffer_copy(lanes)
        view = np.frombuffer(mine, dtype=np.dtype(vs.Lane))
        view["seed"] += np.uint64((b + 1) * per_gpu)
        view["out_seed"] += np.uint64((b + 1) * per_gpu)
        descr.append(mine)
    plans, host_ms, upload_ms = [], [], []

    def make(k):
        p_ = eng.plan(descr[k], n_samples)
        h_, u_ = p_.timing()
        host_ms.append(h_)
        upload_ms.append(u_)
        return p_

    for k in range(2):                                               # warm-up: two batches, untimed
        p_ = make(batches + k)
        p_.launch(vs.VS_KIND_SYNTH, out.data_ptr(), out_pitch=pitch)
        torch.cuda.synchronize(dev)
        p_.status()
        p_.close()
    del host_ms[:], upload_ms[:]
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    nxt = make(0)
    for k in range(batches):
        cur = nxt
        cur.launch(vs.VS_KIND_SYNTH, out.data_ptr(), out_pitch=pitch)   # enqueues; the kernel runs while ...
        if k + 1 < batches:
            nxt = make(k + 1)                                            # ... the next batch's plan is made
        plans.append(cur)
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    for p_ in plans:
        p_.status()
        p_.close()
    # ... and the cheaper way to "new utterances" when only the draws change (what running the reference again does:
    # it seeds from the clock): vs_plan_reseed -- 16 bytes per lane go up instead of a new plan
    base = eng.plan(descr[0], n_samples)
    seeds = [np.arange(per_gpu, dtype=np.uint64) + np.uint64((batches + 5 + b) * per_gpu) for b in range(batches)]
    for b in range(2):
        base.reseed(seeds[b])
        base.launch(vs.VS_KIND_SYNTH, out.data_ptr(), out_pitch=pitch)
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    for b in range(batches):
        base.reseed(seeds[b])
        base.launch(vs.VS_KIND_SYNTH, out.data_ptr(), out_pitch=pitch)
    torch.cuda.synchronize(dev)
    reseed_elapsed = time.perf_counter() - t1
    base.status()
    base.close()
    host_ms.sort()
    return {"batches": batches,
            "reseed_ms_per_batch": round(reseed_elapsed / batches * 1e3, 4),
            "reseed_Msamples/s": round(per_gpu * n_samples * batches / reseed_elapsed / 1e6, 1), "utterances_per_batch": per_gpu, "ms_per_batch": round(elapsed / batches * 1e3, 4),
            "Msamples/s": round(per_gpu * n_samples * batches / elapsed / 1e6, 1),
            "plan_host_ms_median": round(host_ms[len(host_ms) // 2], 3), "plan_upload_ms_median": round(sorted(upload_ms)[len(upload_ms) // 2], 3),
            "how": "vs_plan_create (utterances with new seeds, described beforehand) + launch per batch, plan k + 1 made while kernel k runs; "
                   "reseed_*: ONE plan, vs_plan_reseed + launch per batch (same utterances, new draws); wall clock, python in the loop"}


def measure_config(eng, dev, stream, cfg_i, n_lanes, arith_first, launches=10, warm=5, out_noise_db=None):
    """One BASELINE configuration outside the timed region: plan, `warm` untimed launches (the chip has idled through the
    plan's host work and comes back on a low clock), then HIP events on the launch stream around each of `launches`
    launches, in the three arithmetics (exact, then the opt-in fma and f32).  Returns one record per arithmetic:
    {workload, kernel, arith, kernel_ms_avg, kernel_ms_min, roofline_frac, ...}.
    out_noise_db: the same utterances with "vowel -n <dB>" as well (vowel_new.c:302-324): a launch is then
    the fused kernel (its filter wavefronts take the frame powers along), a scan and the streaming noise pass; the events
    bracket all three and the record says so (`vowel_n_db`, `bytes_per_sample` 6: 2 written by the fused kernel, 2 read and
    2 written by the noise pass -- `roofline_frac` stays on the 2 B of the finished sample, like every other row)."""
    import torch

    import voice_synth_amd as vs
    from voice_synth_amd import configs

    specs, fs, dur, label = configs.config_specs(cfg_i, n_lanes, lane0=0, out_noise_db=out_noise_db)
    lanes, d = vs.lanes_from_specs(specs)
    ns = vs.num_samples(fs, d)
    pitch = vs.row_pitch(ns)
    plan = eng.plan(lanes, ns)
    out = torch.empty((n_lanes, pitch), dtype=torch.int16, device=dev)
    recs = []
    try:
        order = [arith_first] + [a_ for a_ in (vs.VS_ARITH_EXACT, vs.VS_ARITH_FMA, vs.VS_ARITH_F32) if a_ != arith_first]
        for ar in order:
            eng.set_arith(ar)
            for _ in range(warm):
                plan.launch(vs.VS_KIND_SYNTH, out.data_ptr(), out_pitch=pitch)
            torch.cuda.synchronize(dev)
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(launches)]
            for a, b in ev:
                a.record(stream)
                plan.launch(vs.VS_KIND_SYNTH, out.data_ptr(), out_pitch=pitch)
                b.record(stream)
            torch.cuda.synchronize(dev)
            plan.status()
            ms = sorted(a.elapsed_time(b) for a, b in ev)
            avg = sum(ms) / len(ms)
            recs.append({"workload": label, "baseline_config_index": cfg_i - 1, "utterances": n_lanes, "samples_per_utterance": ns,
                         "row_pitch_samples": pitch,
                         "arith": {vs.VS_ARITH_EXACT: "exact", vs.VS_ARITH_FMA: "fma", vs.VS_ARITH_F32: "f32"}[ar],
                         "kernel": plan.kernel_name(vs.VS_KIND_SYNTH), "launches": launches,
                         "kernel_ms_avg": round(avg, 4), "kernel_ms_min": round(ms[0], 4),
                         "Msamples/s": round(n_lanes * ns / (avg * 1e-3) / 1e6, 1),
                         "roofline_frac": round(ALGO_BYTES_PER_SAMPLE * n_lanes * ns / (avg * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)})
            try:
                cus_ = eng.device_info()[1]
                recs[-1]["profile"] = _config_profile(cfg_i, recs[-1]["arith"], n_lanes, out_noise_db, recs[-1]["kernel"], avg, cus_, ns)
            except Exception as exc:  # pragma: no cover
                recs[-1]["profile"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
            if out_noise_db is not None:
                recs[-1].update({"vowel_n_db": out_noise_db, "bytes_per_sample": 6,
                                 "hbm_frac_at_6_bytes": round(6 * n_lanes * ns / (avg * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)})
    finally:
        del out
        plan.close()
        torch.cuda.empty_cache()
    return recs


def _config_profile(cfg_i, arith_name, n_lanes, out_noise_db, kernel_name, kernel_ms, cus, n_samples):
    """What profiles/pmc_*.json hold for one of the other configurations: HBM bytes per launch and the vector-pipe bound,
    each only if the record was taken on THIS tree's kernel sources (else null, like the headline's)."""
    key = "config%d%s%s_%s_%d" % (cfg_i, "_shard" if cfg_i == 4 else "", "_onoise" if out_noise_db is not None else "", arith_name, n_lanes)
    rec = _profile_record("pmc_traffic.json", key)
    sq = _profile_record("pmc_valu.json", key)
    prov = _profile_provenance(rec, sq)
    first = kernel_name.split(" + ")[0]
    if rec and not (prov["traffic"]["matches_tree"] and first in str(rec.get("kernel", ""))):
        rec = None
    if sq and not (prov["valu"]["matches_tree"] and first in str(sq.get("kernel", ""))):
        sq = None
    out = {"profile_key": key, "of_kernel": first + (" (the first of the launch's kernels only; kernel_ms is all of them)" if " + " in kernel_name else ""),
           "traffic": rec["hbm_bytes_per_launch"] if rec else None,
           "traffic_matches_tree": bool(prov["traffic"] and prov["traffic"]["matches_tree"]),
           "valu_matches_tree": bool(prov["valu"] and prov["valu"]["matches_tree"])}
    if rec:
        out["traffic_over_algorithmic"] = round(rec["hbm_bytes_per_launch"] / (ALGO_BYTES_PER_SAMPLE * n_lanes * n_samples), 4)
    if sq:
        wi, clk = sq["SQ_INSTS_VALU_per_launch"], sq.get("clock_GHz_under_profiler")
        out["valu_wave_instructions_per_sample"] = round(wi / (n_lanes * n_samples / 64.0), 2)
        if clk:
            pipe_ms = wi * 4.0 / (4.0 * cus) / (clk * 1e9) * 1e3
            out["pipe_bound_ms"] = round(pipe_ms, 4)
            out["frac_of_pipe_bound"] = round(pipe_ms / kernel_ms, 4)
    return out


def _profile_record(name, key):
    path = os.path.join(ROOT, "profiles", name)
    if os.path.exists(path):
        try:
            return json.load(open(path)).get(key)
        except Exception:
            return None
    return None


def _profile_provenance(traffic_rec, valu_rec):
    """The PMC figures are copied from profiles/ (a --pmc pass cannot share a run with the timed region): say which
    tree they were taken on and whether that is THIS tree -- by content hash of the kernel sources, which the GPU box
    can compute (it sees no .git), with the git hash the pass recorded next to it."""
    from tools.provenance import git_head, kernel_sources_sha16

    tree = kernel_sources_sha16()
    out = {"tree_kernel_sources_sha16": tree, "tree_head": git_head()}
    for name, rec in (("traffic", traffic_rec), ("valu", valu_rec)):
        out[name] = None if not rec else {"profile_head": rec.get("profile_head"),
                                          "kernel_sources_sha16": rec.get("kernel_sources_sha16"),
                                          "matches_tree": rec.get("kernel_sources_sha16") == tree}
    return out


LINE_OUT = None   # where the JSON line goes (None: sys.stdout); see reserve_stdout


def reserve_stdout():
    """Multi-rank runs: the process's standard output carries ONE line, rank 0's.  RCCL writes a five-line banner (version,
    host, library path) to STDOUT when the first communicator is made -- seen in tools/rccl_self_probe.py's output -- and
    whatever else a library prints would land there too: file descriptor 1 is pointed at standard error for the rest of the
    run, and the line goes to a copy of the original made here."""
    global LINE_OUT
    if LINE_OUT is None:
        sys.stdout.flush()
        LINE_OUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)


def print_line(obj):
    out = LINE_OUT or sys.stdout
    out.write(json.dumps(obj) + "\n")
    out.flush()


def preflight_errors(ranks_seen, world):
    """What is wrong with a node run before anything is timed: [] or the findings -- N ranks must drive N DIFFERENT devices
    (host + PCI bus id), over RCCL, each in a communicator of all N."""
    wrong = []
    distinct = len({(r_["host"], r_["pci_bus_id"]) for r_ in ranks_seen})
    if len(ranks_seen) != world:
        wrong.append("%d ranks reported, expected %d" % (len(ranks_seen), world))
    if distinct != world:
        wrong.append("%d ranks drive %d distinct devices" % (world, distinct))
    if any(r_.get("comm_world_size") != world for r_ in ranks_seen):
        wrong.append("communicator sizes %s, expected %d" % ([r_.get("comm_world_size") for r_ in ranks_seen], world))
    if any(r_.get("backend") != "nccl" for r_ in ranks_seen):
        wrong.append("backends %s, expected nccl (= RCCL)" % sorted({str(r_.get("backend")) for r_ in ranks_seen}))
    return wrong


def _free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves, as a CHILD process
    (`python -m torch.distributed.run`, one rank per GPU), relay its output unchanged and leave with its exit code.
    Called before this process has imported torch or touched the GPU; nothing is exec'ed over a running program."""
    import subprocess

    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL between processes needs it on this image
    env["VS_BENCH_SELF_LAUNCHED"] = "1"
    sys.stderr.write("bench.py: --gpus %d without a launcher: starting %s\n" % (n, " ".join(cmd)))
    sys.stderr.flush()
    child = subprocess.Popen(cmd, env=env)      # inherits stdout / stderr: rank 0's JSON line arrives as it is
    try:
        rc = child.wait()
    except KeyboardInterrupt:  # pragma: no cover
        child.terminate()
        rc = child.wait()
    return rc if rc >= 0 else 128 - rc


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if "WORLD_SIZE" not in os.environ and "RANK" not in os.environ and args.gpus > 1:
            sys.exit(launch_ranks(args.gpus))
        sys.exit("bench.py: WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))

    # dmabuf IPC: on this image RCCL between processes needs it (hipIpcGetMemHandle: invalid argument without); the driver's
    # environment exports it, this is for a launcher that does not -- read when the HIP runtime starts, i.e. before torch
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist

    import voice_synth_amd as vs
    from voice_synth_amd import configs

    # VS_BENCH_REHEARSAL=1 (tests only): all ranks share device 0 and talk over gloo -- the N > 1 control
    # flow (sharded lane keys, barriers, max-over-ranks timing, rank 0's line) on a one-GPU box.  RCCL
    # refuses two ranks on one device, so the gather leg is left out there.
    rehearsal = os.environ.get("VS_BENCH_REHEARSAL") in ("1", "2")
    if rehearsal:
        local_rank = 0
        if os.environ.get("VS_BENCH_REHEARSAL") == "1":
            args.no_gather = True   # gloo on device tensors: only with VS_BENCH_REHEARSAL=2
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)
    if use_dist:  # launched by torch.distributed.run: one rank per GPU over RCCL
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        reserve_stdout()
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    # ---- workload: per-GPU batch, lane keys from the global lane index ----
    full = {1: 1, 2: 1024, 3: 65536, 4: 262144 // 8, 5: 65536}[args.config]
    per_gpu = args.lanes or full
    lane0 = rank * per_gpu
    specs, fs, dur, label = configs.config_specs(args.config, per_gpu, lane0=lane0, out_noise_db=args.vowel_n)
    lanes, d = vs.lanes_from_specs(specs)
    n_samples = vs.num_samples(fs, d)
    # the PCM buffer is the caller's: rows of n_samples at the pitch vs_row_pitch() names (include/voice_synth.h: a whole
    # number of 128-byte lines, 3 mod 4 -- 16064 samples for 16000), or, --dense-rows, only rounded up to 16 bytes.  The
    # line says which (config.row_pitch_samples) and carries the other one's launch time (`dense_rows` / `pitched_rows`).
    dense_pitch = (n_samples + 7) & ~7
    pitch = dense_pitch if args.dense_rows else vs.row_pitch(n_samples)

    arith = {"exact": vs.VS_ARITH_EXACT, "fma": vs.VS_ARITH_FMA, "f32": vs.VS_ARITH_F32}[args.arith]
    stream = torch.cuda.current_stream(dev)
    eng = vs.Engine(local_rank, arith=arith, stream=stream.cuda_stream)
    dev_name, cus = eng.device_info()
    # which device every rank really drives: a scaling curve is only one if N DIFFERENT devices took part
    ident = {"rank": rank, "local_rank": local_rank, "device_index": torch.cuda.current_device(), "pci_bus_id": eng.device_pci(),
             "host": socket.gethostname(), "backend": dist.get_backend() if use_dist else None,
             "comm_world_size": dist.get_world_size() if use_dist else 1}
    ranks_seen = [ident]
    if use_dist:
        ranks_seen = [None] * world
        dist.all_gather_object(ranks_seen, ident)
        ranks_seen.sort(key=lambda r_: r_["rank"])
    distinct_devices = len({(r_["host"], r_["pci_bus_id"]) for r_ in ranks_seen})
    # Pre-flight of a node run (the first real 8-GPU run is the driver's, unattended): N ranks must drive N DIFFERENT
    # devices and every rank's communicator must be RCCL over all N of them -- a mis-bound launch (two ranks on one card, a
    # rank that came up on gloo) would still print a curve.  Outside the one-GPU rehearsals of the tests it is an error:
    # rank 0 prints the line with it, every rank leaves with a non-zero code, nothing is timed.
    if world > 1 and not rehearsal:
        wrong = preflight_errors(ranks_seen, world)
        if wrong:
            if rank == 0:
                print_line({"metric": "synthesised Msamples/s (whole node) at 1/2/4/8 MI355X; RMS vs C ref", "value": None,
                            "unit": "Msamples/s", "n_gpus": world, "error": "pre-flight: " + "; ".join(wrong),
                            "ranks_seen": ranks_seen, "distinct_devices": distinct_devices})
            eng.close()
            dist.destroy_process_group()
            sys.exit(5)
    plan = eng.plan(lanes, n_samples)                     # lane records + cos rows -> HBM
    plan_host_ms, plan_upload_ms = plan.timing()
    kernel_name = plan.kernel_name(vs.VS_KIND_SYNTH)
    # the three-role layouts are built on "wavefront w of a workgroup runs on SIMD w % 4": asked of the hardware
    # (HW_ID probe), and the plan says whether it had to fall back to two roles
    c12, c8 = eng.simd_dealing()
    wave_to_simd = dict(plan.roles(), cyclic_12_wavefronts=c12, cyclic_8_wavefronts=c8)
    out = torch.empty((per_gpu, max(pitch, vs.row_pitch(n_samples))), dtype=torch.int16, device=dev)
    if out.shape[1] != pitch:
        out = out.view(-1)[: per_gpu * pitch].view(per_gpu, pitch)   # --dense-rows: the same allocation, rows at the dense pitch

    def launch():
        plan.launch(vs.VS_KIND_SYNTH, out.data_ptr(), out_pitch=pitch)

    def sync_all():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        launch()
    sync_all()

    # ---- timed region: exactly K steps, HIP events around every launch on the launch stream ----
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record(stream)
        launch()
        b.record(stream)
    sync_all()
    elapsed = time.perf_counter() - t0
    health = plan.status()   # raises if a device-side bounded wait ran out in any of the launches

    kern_ms = [a.elapsed_time(b) for a, b in ev]
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    k = torch.tensor([sum(kern_ms) / len(kern_ms)], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(k, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    kern_ms_avg = float(k.item())

    samples_per_step = per_gpu * n_samples * world
    value = samples_per_step * args.steps / elapsed / 1e6

    # the rows of what was just timed, for the parity check inside the cpu_baseline leg: the whole
    # batch when that leg runs (its CPU sample starts at lane 0 and is usually larger than the batch)
    n_check = per_gpu if (world == 1 and not args.no_cpu_baseline) else min(64, per_gpu)
    first_rows = out[:n_check, :n_samples].cpu().numpy() if rank == 0 else None

    # ---- sustained: the same plan back to back for >= SUSTAINED_S seconds, outside the timed region.  The kernel is
    # compute-bound and the chip settles on a lower clock under a load that lasts (DVFS); 20 steps are a 50 ms burst.
    # HIP events on the launch stream in front of the first and behind the last launch.
    n_sus = int(max(50, min(40000, SUSTAINED_S / max(kern_ms_avg * 1e-3, 1e-5) * 1.05)))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(dev)
    e0.record(stream)
    n_block, n_done = n_sus, 0
    while True:
        # (the estimate above comes from the burst, whose first launches run on a cold clock: top up until the
        # launches really span SUSTAINED_S -- one host wait per block, nothing else between the launches)
        for _ in range(n_block):
            launch()
        n_done += n_block
        e1.record(stream)
        torch.cuda.synchronize(dev)
        sus_ms = e0.elapsed_time(e1)
        if sus_ms >= SUSTAINED_S * 1e3 or n_done >= 40000:
            break
        n_block = int(max(10, (SUSTAINED_S * 1e3 - sus_ms) / (sus_ms / n_done) * 1.05 + 1))
    n_sus = n_done
    plan.status()
    st = torch.tensor([sus_ms / n_sus], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(st, op=dist.ReduceOp.MAX)
    sus_kern_ms = float(st.item())
    sustained = {"launches": n_sus, "seconds": round(sus_ms * 1e-3, 3), "kernel_ms_avg": round(sus_kern_ms, 4),
                 "Msamples/s": round(per_gpu * n_samples * world / (sus_kern_ms * 1e-3) / 1e6, 1),
                 "roofline_frac": round(ALGO_BYTES_PER_SAMPLE * per_gpu * n_samples / (sus_kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                 "ratio_to_timed_region": round((per_gpu * n_samples * world / (sus_kern_ms * 1e-3) / 1e6) / value, 4),
                 "how": "back-to-back launches of the timed plan, HIP events in front of the first and behind the last (max over ranks)"}

    # ---- the other arithmetic mode, outside the timed region: 3 warm-ups, then HIP events around
    # each of 10 launches, as in the timed region (mean, median and min reported) ----
    other = vs.VS_ARITH_FMA if arith == vs.VS_ARITH_EXACT else vs.VS_ARITH_EXACT
    eng.set_arith(other)
    other_kernel = plan.kernel_name(vs.VS_KIND_SYNTH)
    for _ in range(3):
        launch()
    torch.cuda.synchronize(dev)
    oev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for a, b in oev:
        a.record(stream)
        launch()
        b.record(stream)
    torch.cuda.synchronize(dev)
    plan.status()
    other_kern = sorted(a.elapsed_time(b) for a, b in oev)
    other_ms = sum(other_kern) / len(other_kern)
    n_tol_rows = min(4096, per_gpu)   # rows of the tolerance modes kept for their RMS against the CPU sample
    other_rows = out[:n_tol_rows, :n_samples].cpu().numpy() if (rank == 0 and other != vs.VS_ARITH_EXACT) else None

    # ---- the third arithmetic, VS_ARITH_F32 (packed single precision, opt-in: include/voice_synth.h), same treatment ----
    eng.set_arith(vs.VS_ARITH_F32)
    f32_kernel = plan.kernel_name(vs.VS_KIND_SYNTH)
    for _ in range(3):
        launch()
    torch.cuda.synchronize(dev)
    fev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for a, b in fev:
        a.record(stream)
        launch()
        b.record(stream)
    torch.cuda.synchronize(dev)
    plan.status()
    f32_kern = sorted(a.elapsed_time(b) for a, b in fev)
    f32_ms = sum(f32_kern) / len(f32_kern)
    f32_rows = out[:n_tol_rows, :n_samples].cpu().numpy() if rank == 0 else None

    # ---- what the row pitch is worth, outside the timed region: the same plan into the same allocation with rows at the
    # pitch vs_row_pitch() names and with dense rows (16-byte multiples), launches of the two INTERLEAVED (the chip's clock
    # drifts over a run: only launches that alternate can be compared), both arithmetic modes, 3 warm-ups + 10 launches
    # with HIP events each (profiles/r05_row_pitch.txt, tools/pitch_probe.py).
    pitch_ab = {"how": "launches alternate between the two pitches; HIP events around each; outside the timed region"}
    ab_pitches = (("pitched", vs.row_pitch(n_samples)), ("dense", dense_pitch))
    for a_mode, a_name in ((vs.VS_ARITH_EXACT, "exact"), (vs.VS_ARITH_FMA, "fma")):
        eng.set_arith(a_mode)
        for _ in range(3):
            for _, p_ in ab_pitches:
                plan.launch(vs.VS_KIND_SYNTH, out.data_ptr(), out_pitch=p_)
        torch.cuda.synchronize(dev)
        aev = {k_: [] for k_, _ in ab_pitches}
        for _ in range(10):
            for k_, p_ in ab_pitches:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(stream)
                plan.launch(vs.VS_KIND_SYNTH, out.data_ptr(), out_pitch=p_)
                b.record(stream)
                aev[k_].append((a, b))
        torch.cuda.synchronize(dev)
        plan.status()
        for k_, p_ in ab_pitches:
            a_ms = sorted(a.elapsed_time(b) for a, b in aev[k_])
            pitch_ab.setdefault(k_, {"row_pitch_samples": p_})[a_name] = {
                "kernel_ms_avg": round(sum(a_ms) / len(a_ms), 4), "kernel_ms_median": round(a_ms[len(a_ms) // 2], 4),
                "kernel_ms_min": round(a_ms[0], 4),
                "Msamples/s_per_gpu": round(per_gpu * n_samples / (sum(a_ms) / len(a_ms) * 1e-3) / 1e6, 1)}
    eng.set_arith(arith)

    # ---- fresh batches, outside the timed region (N = 1 only): what a caller pays who synthesises NEW utterances -- a plan
    # per batch (new seeds: vs_plan_create on the host threads + upload) and its launch, batch k + 1's plan made while
    # batch k's kernel runs (a launch only enqueues).  The timed region above launches ONE plan over and over: that is the
    # kernel; this is the product path around it.
    fresh = None
    if world == 1 and rank == 0 and not args.no_other_configs:
        try:
            fresh = fresh_batches(eng, dev, stream, lanes, n_samples, out, pitch, per_gpu)
        except Exception as exc:  # pragma: no cover - reported in the line
            fresh = {"error": "%s: %s" % (type(exc).__name__, exc)}

    # ---- the same from plain C, no Python in the loop (N = 1 only): cli/vs_bench.c --fresh -- 50 batches, the plan of batch
    # k + 1 made on a second host thread while kernel k runs, plans of two batches ago destroyed meanwhile, device time from
    # in front of the first launch to behind the last; and its steady-state loop (one plan, 50 launches) in the same process
    # for the ratio.  A child process with its own HIP context, after this process's launches have drained.
    fresh_c = None
    if world == 1 and rank == 0 and not args.no_other_configs and args.config == 3 and not args.lanes:
        try:
            torch.cuda.synchronize(dev)
            exe = os.path.join(ROOT, "voice_synth_amd", "bin", "vs_bench")
            env = dict(os.environ, VS_DEVICE=str(dev.index or 0))
            recs = []
            for extra in ([], ["--fresh"]):
                r = subprocess.run([exe, "--arith", args.arith, "--steps", "50", "--warmup", "10"] + extra, capture_output=True, timeout=300, env=env)
                if r.returncode != 0:
                    raise RuntimeError("vs_bench %s: %s" % (" ".join(extra), r.stderr.decode()[-300:]))
                recs.append(json.loads(r.stdout.decode().strip().splitlines()[-1]))
            fresh_c = {"ms_per_batch": recs[1]["ms_per_batch"], "Msamples/s": recs[1]["value"], "batches": recs[1]["batches"],
                       "same_plan_ms_per_launch": recs[0]["ms_per_step"],
                       "ratio_to_same_plan": round(recs[1]["ms_per_batch"] / recs[0]["ms_per_step"], 4),
                       "plan_host_ms_avg": recs[1]["plan_host_ms_avg"], "plan_upload_ms_avg": recs[1]["plan_upload_ms_avg"],
                       "plan_destroy_ms_avg": recs[1]["plan_destroy_ms_avg"],
                       "last_batch_equals_a_plain_launch": recs[1]["last_batch_equals_a_plain_launch"], "how": recs[1]["how"]}
        except Exception as exc:  # pragma: no cover - reported in the line
            fresh_c = {"error": "%s: %s" % (type(exc).__name__, exc)}

    # ---- the other BASELINE configurations, outside the timed region (N = 1 only; at N > 1 the config-4 block below
    # is the second workload): configs 2, 4 (one GPU's shard of the 8-GPU cut) and 5, each launched a few times with
    # HIP events around every launch, in both arithmetic contracts -- so that the driver's line carries every
    # configuration's kernel time, not only the headline's ----
    other_configs = None
    if world == 1 and rank == 0 and not args.no_other_configs and args.config == 3 and not args.lanes:
        other_configs = []
        for cfg_i, n_l, onoise in ((2, 1024, None), (4, 262144 // 8, None), (5, 65536, None), (3, 65536, 20.0)):
            try:
                other_configs += measure_config(eng, dev, stream, cfg_i, n_l, arith, out_noise_db=onoise)
            except Exception as exc:  # pragma: no cover - reported in the line
                other_configs.append({"baseline_config_index": cfg_i - 1, "error": "%s: %s" % (type(exc).__name__, exc)})
        eng.set_arith(arith)

    result = None
    if rank == 0:
        achieved = ALGO_BYTES_PER_SAMPLE * per_gpu * n_samples / (kern_ms_avg * 1e-3) / 1e9
        key = "config%d%s%s_%s_%d" % (args.config, "_shard" if args.config == 4 else "", "_onoise" if args.vowel_n is not None else "",
                                        args.arith, per_gpu)
        rec = _profile_record("pmc_traffic.json", key)
        sq = _profile_record("pmc_valu.json", key)
        prov = _profile_provenance(rec, sq)
        # counters of another tree's kernel are not this kernel's: no figure rather than a stale one
        if rec and not prov["traffic"]["matches_tree"]:
            rec = None
        if sq and not prov["valu"]["matches_tree"]:
            sq = None
        traffic = rec["hbm_bytes_per_launch"] if rec else None
        tflops = FLOP_PER_SAMPLE * per_gpu * n_samples / (kern_ms_avg * 1e-3) / 1e12
        valu = None
        if sq and kernel_name not in str(sq.get("kernel", "")):
            sq = None   # the committed counters are of another kernel: no figure rather than a mixed one
        if sq:
            # wave-instructions from the SQ counters of the committed profile of THIS kernel; the
            # duration is this run's.  Issue ceiling of one wavefront per SIMD measured by
            # tools/ubench: one fp64 instruction per 5.3 shader cycles.
            wi = sq["SQ_INSTS_VALU_per_launch"]
            units = per_gpu * n_samples / 64.0                       # 64-utterance groups x samples
            all_wi = sum(sq.get(k_, 0.0) for k_ in ("SQ_INSTS_VALU_per_launch", "SQ_INSTS_SALU_per_launch",
                                                    "SQ_INSTS_LDS_per_launch", "SQ_INSTS_VMEM_WR_per_launch"))
            rate = wi / (kern_ms_avg * 1e-3) / (4 * cus)            # VALU wave-instructions per second per SIMD
            waves_per_simd = sq.get("SQ_WAVES_per_launch", 4.0 * cus) / (4.0 * cus)
            # the binding roof, stated as a time: every vector instruction of a wavefront occupies its SIMD's vector pipe
            # for 4 cycles (16 lanes wide, 64 lanes per wavefront; fp64 and 32-bit alike on this chip) -- the launch cannot
            # be shorter than instructions x 4 cycles / SIMDs at the clock the profiled launches ran at
            clk = sq.get("clock_GHz_under_profiler")
            pipe_ms = (wi * 4.0 / (4.0 * cus) / (clk * 1e9) * 1e3) if clk else None
            valu = {"valu_wave_instructions_per_sample": round(wi / units, 2),
                    "pipe_bound_ms": round(pipe_ms, 4) if pipe_ms else None,
                    "frac_of_pipe_bound": round(pipe_ms / kern_ms_avg, 4) if pipe_ms else None,
                    "pipe_bound_how": "vector wave-instructions per launch (SQ_INSTS_VALU) x 4 cycles / (4 x %d SIMDs) / %.3f GHz (the clock of the "
                                      "profiled launches); frac = that / this run's kernel_ms_avg" % (cus, clk or 0.0),
                    "all_wave_instructions_per_sample": round(all_wi / units, 2),
                    "waves_per_simd": round(waves_per_simd, 2),
                    "valu_issue_rate_per_simd_MHz": round(rate / 1e6, 1),
                    # SQ_ACTIVE_INST_VALU and SQ_WAVE_CYCLES count quad-cycles (MI355X_MICROARCH.md): the
                    # share of the SIMDs' time in which a vector instruction is executing
                    "valu_pipe_busy_frac": (round(sq["SQ_ACTIVE_INST_VALU_per_launch"] / (sq["SQ_WAVE_CYCLES_per_launch"] / waves_per_simd), 3)
                                            if "SQ_ACTIVE_INST_VALU_per_launch" in sq and "SQ_WAVE_CYCLES_per_launch" in sq else None),
                    "kernel_in_profile": sq.get("kernel"),
                    "source": "profiles/pmc_valu.json[%s]" % key}
        result = {
            "metric": "synthesised Msamples/s (whole node) at 1/2/4/8 MI355X; RMS vs C ref",
            "value": round(value, 1),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if args.arith == "f32" else "f64",
            "data": "synthetic",
            "config": {
                "workload": label,
                "baseline_config_index": args.config - 1,
                "utterances_per_gpu": per_gpu,
                "samples_per_utterance": n_samples,
                "sample_rate_hz": fs,
                "row_pitch_samples": pitch,
                "rows": "dense (16-byte multiple)" if args.dense_rows else "vs_row_pitch(): whole 128-byte lines, their number 3 mod 4",
                "arith": args.arith,
                "parallelism": "utterances sharded over %d GPU(s), no data-path collective" % world,
            },
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4),
                "traffic": traffic,
                "kernel": kernel_name,
                "kernel_ms_avg": round(kern_ms_avg, 4),
                "kernel_ms_min": round(min(kern_ms), 4),
                "algorithmic_bytes_per_launch": ALGO_BYTES_PER_SAMPLE * per_gpu * n_samples,
                "fp64_valu": {"achieved_TFLOPs": round(tflops, 2), "peak_TFLOPs": FP64_VALU_PEAK_TFLOPS,
                              "frac": round(tflops / FP64_VALU_PEAK_TFLOPS, 4),
                              "flop_per_sample": FLOP_PER_SAMPLE},
                "valu": valu,
                "profile": prov,
            },
            "sustained": sustained,
            "other_arith": {"arith": "fma" if arith == vs.VS_ARITH_EXACT else "exact",
                            "kernel": other_kernel,
                            "launches": len(other_kern),
                            "kernel_ms_avg": round(other_ms, 4),
                            "kernel_ms_median": round(other_kern[len(other_kern) // 2], 4),
                            "kernel_ms_min": round(other_kern[0], 4),
                            "Msamples/s_per_gpu": round(per_gpu * n_samples / (other_ms * 1e-3) / 1e6, 1)},
            "f32_arith": {"arith": "f32", "kernel": f32_kernel, "launches": len(f32_kern),
                          "kernel_ms_avg": round(f32_ms, 4), "kernel_ms_median": round(f32_kern[len(f32_kern) // 2], 4),
                          "kernel_ms_min": round(f32_kern[0], 4),
                          "Msamples/s_per_gpu": round(per_gpu * n_samples / (f32_ms * 1e-3) / 1e6, 1),
                          "roofline_frac": round(ALGO_BYTES_PER_SAMPLE * per_gpu * n_samples / (f32_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                          "contract": "opt-in; measured distance from the reference per vowel table in tests/golden/f32_bounds.json "
                                      "(RMS 4.6e-6 .. 1.9e-5 of full scale at the default gain); rms_vs_c_ref below is this workload's"},
            "row_pitch_ab": pitch_ab,
            "other_configs": other_configs,
            "fresh_batches": fresh,
            "fresh_batches_c": fresh_c,
            "plan": {"host_ms": round(plan_host_ms, 2), "upload_ms": round(plan_upload_ms, 2),
                     "note": "vs_plan_create of the per-GPU batch: validation + parameter expansion on host threads, "
                             "sort, cos rows; allocation + upload + wait.  Outside every timed region."},
            "launch_health_word": health,
            "wave_to_simd": wave_to_simd,
            "device": dev_name.strip(),
            "ranks_seen": ranks_seen,
            "distinct_devices": distinct_devices,
        }
        if rehearsal:
            result["rehearsal"] = "VS_BENCH_REHEARSAL: every rank shares device 0 over gloo (tests only; the numbers mean nothing)"
        if world == 1 and not args.no_cpu_baseline:
            tol_rows = {"f32": f32_rows}
            if other_rows is not None:
                tol_rows["fma"] = other_rows
            cb = cpu_baseline(lambda n: configs.config_specs(args.config, n, lane0=0, out_noise_db=args.vowel_n)[0], n_samples,
                              args.cpu_seconds, first_rows, tol_rows)
            result["cpu_baseline"] = cb
            # the tolerance modes against the same CPU sample (first rows of their last launch)
            for name, rec in (cb.pop("tolerance_modes", None) or {}).items():
                result["f32_arith" if name == "f32" else "other_arith"]["rms_vs_c_ref"] = rec
            # the metric's second figure, next to the throughput
            result["rms_vs_c_ref"] = {"lsb": cb.get("gpu_rms_error_lsb"),
                                      "normalised": cb.get("gpu_rms_error_normalised"),
                                      "rows_checked": cb.get("gpu_rows_checked"),
                                      "tolerance_normalised": 1e-5}

    # ---- N > 1: the configuration BASELINE.json names for the node -- config 4, 262144 utterances x
    # 44100 samples cut over the N ranks -- timed like the steps above, and then the same synthesis
    # WITH delivery of the PCM to rank 0 over RCCL, end to end.  Runs LAST and under a watchdog that
    # knows the block's phases (Phases): a phase without progress for its allowance is a stalled
    # exchange -- rank 0 prints the line with what it has and the phase's name, and every rank leaves
    # with a NON-ZERO exit code.  A phase that FAILS on one rank (an exception, no room) is agreed on by
    # all ranks at the phase's end and abandoned together: the line carries the error, the exit code is 0.
    out_lock = threading.Lock()
    printed = [False]
    extra = {}

    def emit():
        with out_lock:
            if rank == 0 and not printed[0]:
                printed[0] = True
                result.update(extra)
                print_line(result)

    leg_done = threading.Event()
    phases = Phases()

    store = _job_store() if use_dist else None

    def watchdog():
        abort_seen = None          # (phase name, when, message): a peer reported a failure while we were in this phase
        while not leg_done.wait(0.5):
            stalled = phases.stalled()
            why = ("phase '%s': no progress for %d s" % stalled) if stalled else None
            cur = phases.current()
            if why is None and store is not None and cur is not None:
                try:
                    if store.check([ABORT_KEY]):
                        if abort_seen is None or abort_seen[0] != cur:
                            abort_seen = (cur, time.monotonic(), store.get(ABORT_KEY).decode(errors="replace"))
                        # the grace runs from the last sign of life of THIS rank's phase, not from the moment the
                        # key appeared: a rank that is still working towards the phase's agreement (building its
                        # chunk plans, say) gets there and leaves with everybody else, exit code 0; only one that
                        # sits in a collective the failed rank will never join shows no progress
                        elif time.monotonic() - max(abort_seen[1], phases.last_tick()) > ABORT_GRACE_S:
                            why = "phase '%s' abandoned: %s" % (cur, abort_seen[2])
                    else:
                        abort_seen = None      # the phase was agreed on and the key cleared (run_phase)
                except Exception:  # pragma: no cover - the store went away with its rank
                    pass
            if why:
                with out_lock:
                    if rank == 0 and not printed[0]:
                        printed[0] = True
                        result["config4"] = dict(phases.partial, error=why)
                        print_line(result)
                os._exit(3)   # a stalled exchange is a finding, not a success

    if world > 1 and not args.no_config4:
        threading.Thread(target=watchdog, daemon=True).start()
        del out
        torch.cuda.empty_cache()
        try:
            extra["config4"] = config4_block(args, eng, dev, stream, rank, world, cus, sync_all, phases)
        except PhaseFailed as exc:
            extra["config4"] = dict(phases.partial, error=str(exc))
        except Exception as exc:  # pragma: no cover - depends on the node
            extra["config4"] = dict(phases.partial, error="%s: %s" % (type(exc).__name__, exc))
        if rank == 0 and isinstance(extra.get("config4"), dict):
            extra["config4"]["ranks_seen"] = ranks_seen
            extra["config4"]["distinct_devices"] = distinct_devices
            if isinstance(extra["config4"].get("gather"), dict):   # whoever reads the gather figures sees who exchanged
                extra["config4"]["gather"]["distinct_devices"] = distinct_devices
                extra["config4"]["gather"]["devices"] = [r_["pci_bus_id"] for r_ in ranks_seen]
    leg_done.set()
    emit()

    step = ["plan.close"]
    if use_dist:
        # The line is out and this rank's work is done, but a teardown step that hangs (a closing barrier a peer
        # never joins, a device that does not answer) must not hold the process for ever -- and must not read as
        # a success either: the deadline leaves with a NON-ZERO code and names the step.  Only when a peer has SAID
        # that it failed (the abort key in the job's store: that peer's own exit code fails the job) is a barrier
        # that never completes the expected outcome.
        def leave():
            time.sleep(TEARDOWN_DEADLINE_S)
            peer_failed = False
            try:
                peer_failed = store is not None and store.check([ABORT_KEY])
            except Exception:  # pragma: no cover - the store went away with its rank
                peer_failed = step[0] in ("barrier", "destroy_process_group")
            sys.stderr.write("bench.py: rank %d: teardown step '%s' still running after %d s%s\n"
                             % (rank, step[0], TEARDOWN_DEADLINE_S, " (a peer reported a failure)" if peer_failed else ""))
            sys.stderr.flush()
            os._exit(0 if (peer_failed and step[0] in ("barrier", "destroy_process_group")) else 4)
        threading.Thread(target=leave, daemon=True).start()

    plan.close()
    step[0] = "eng.close"
    eng.close()
    if use_dist:
        step[0] = "barrier"
        dist.barrier()
        step[0] = "destroy_process_group"
        dist.destroy_process_group()


class PhaseFailed(RuntimeError):
    pass


class Phases:
    """Progress stamps of the config-4 block, read by the watchdog thread: the phase that is running, when
    it last showed progress (entering it, or a chunk delivered: tick()), and how long it may stay silent."""

    def __init__(self):
        self._lock = threading.Lock()
        self._name, self._t, self._allow = None, 0.0, 0.0
        self.partial = {}          # what the block has measured so far (goes into the line if a later phase stalls)

    def enter(self, name, allow_s):
        with self._lock:
            self._name, self._t, self._allow = name, time.monotonic(), float(allow_s)

    def tick(self):
        with self._lock:
            self._t = time.monotonic()

    def leave(self):
        with self._lock:
            self._name = None

    def last_tick(self):
        with self._lock:
            return self._t

    def stalled(self):
        with self._lock:
            if self._name is not None and time.monotonic() - self._t > self._allow:
                return (self._name, int(self._allow))
        return None

    def current(self):
        with self._lock:
            return self._name


ABORT_KEY = "vs_bench_abort"
ABORT_GRACE_S = 10   # a rank that failed says so in the job's store; peers that have shown no progress (Phases.tick)
                     # for this long afterwards sit in a collective they will never leave by themselves


def _job_store():
    """the rendezvous store of the process group (a private accessor of torch.distributed: None if it moved)"""
    try:
        from torch.distributed import distributed_c10d
        return distributed_c10d._get_default_store()
    except Exception:  # pragma: no cover
        return None


def run_phase(phases, name, allow_s, fn, rank, dev):
    """One phase of a multi-rank block: fn() on every rank, then an all-reduce (MIN) of an ok flag, so that a rank
    that failed takes the others out of the block with it instead of leaving them in the NEXT collective.  Raises
    PhaseFailed on every rank if any failed.  A rank that fails while its peers are still inside fn's own
    collectives cannot be agreed with; it says so in the job's store, where the peers' watchdogs find it."""
    import torch
    import torch.distributed as dist

    phases.enter(name, allow_s)
    err, res = None, None
    try:
        res = fn()
    except Exception as exc:  # noqa: BLE001 - reported in the line
        err = exc
        st = _job_store()
        if st is not None:
            try:
                st.set(ABORT_KEY, "rank %d failed in '%s': %s: %s" % (rank, name, type(exc).__name__, exc))
            except Exception:  # pragma: no cover
                pass
    ok = torch.tensor([0 if err else 1], dtype=torch.int32, device=dev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    phases.leave()
    if err is not None:
        # every rank has taken part in the agreement: the key has done its job (a later phase starts clean)
        st = _job_store()
        if st is not None:
            try:
                st.delete_key(ABORT_KEY)
            except Exception:  # pragma: no cover
                pass
    if int(ok.item()) == 0:
        raise PhaseFailed("phase '%s' failed on %s" % (name, ("this rank (%d): %s: %s" % (rank, type(err).__name__, err)) if err
                                                        else "another rank (see its stderr)"))
    return res


def config4_block(args, eng, dev, stream, rank, world, cus, sync_all, phases):
    """BASELINE.json configs[3]: 262144 utterances, mixed vowels, 22.05 kHz, 2 s, cut over the ranks in
    contiguous lane blocks (voice_synth_amd.dist.shard_range, the same cut vs_node_* makes in C), lane
    keys from the global lane index.  `value`: K launches, PCM left sharded, as the headline figure;
    `value_with_gather`: the same synthesis in chunks with every finished chunk travelling to rank 0
    while the next one is being synthesised (PipelinedGather), timed end to end.
    Runs as a sequence of PHASES (see Phases / the watchdog in main): each ends with an all-reduce of an
    ok flag, so that a rank that failed takes the others out of the block with it instead of leaving them
    in the next collective."""
    import torch
    import torch.distributed as dist

    import voice_synth_amd as vs
    from voice_synth_amd import configs
    from voice_synth_amd.dist import PipelinedGather, gather_pcm, shard_range

    def phase(name, allow_s, fn):
        return run_phase(phases, name, allow_s, fn, rank, dev)

    total = args.config4_lanes or 262144
    lo, hi = shard_range(total, rank, world)
    per = hi - lo
    specs, fs, dur, label = configs.config_specs(4, per, lane0=lo)
    lanes, d = vs.lanes_from_specs(specs)
    ns = vs.num_samples(fs, d)
    pitch = vs.row_pitch(ns)
    steps = max(2, min(args.steps, 5))
    nbytes = (total - (shard_range(total, 0, world)[1] - shard_range(total, 0, world)[0])) * ns * 2   # into rank 0
    allow = PHASE_DEADLINE_S + nbytes / 1e9       # 60 s + 1 s per GB that has to reach rank 0

    def timed_steps():
        if os.environ.get("VS_BENCH_FAULT") == "stall_rank1" and rank == 1:
            time.sleep(10 ** 6)      # tests only: a rank that never reaches the phase's collectives
        plan = eng.plan(lanes, ns)
        kernel = plan.kernel_name(vs.VS_KIND_SYNTH)
        out = torch.empty((per, pitch), dtype=torch.int16, device=dev)

        def launch():
            plan.launch(vs.VS_KIND_SYNTH, out.data_ptr(), out_pitch=pitch)

        for _ in range(2):
            launch()
        sync_all()
        phases.tick()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        t0 = time.perf_counter()
        for a, b in ev:
            a.record(stream)
            launch()
            b.record(stream)
        sync_all()
        elapsed = time.perf_counter() - t0
        plan.status()
        kern = sum(a.elapsed_time(b) for a, b in ev) / steps
        t = torch.tensor([elapsed, kern], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kern = float(t[0].item()), float(t[1].item())
        achieved = ALGO_BYTES_PER_SAMPLE * per * ns / (kern * 1e-3) / 1e9
        del out
        plan.close()
        torch.cuda.empty_cache()
        return {"workload": label, "utterances": total, "utterances_per_gpu": per, "samples_per_utterance": ns,
                "steps": steps, "ms_per_step": round(elapsed / steps * 1e3, 4),
                "value": round(total * ns * steps / elapsed / 1e6, 1), "unit": "Msamples/s",
                "roofline_per_gpu": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                     "frac": round(achieved / HBM_PEAK_GBPS, 4), "kernel": kernel,
                                     "kernel_ms_avg": round(kern, 4)}}

    block = phase("config-4 steps", PHASE_DEADLINE_S, timed_steps)
    phases.partial = dict(block)
    if args.no_gather:
        return block

    # ---- with delivery to rank 0 ----
    state = {}

    def gather_setup():
        # rank 0 needs room for the gathered PCM twice (overlapped + comparison); a rank without room RAISES and
        # the agreement at the end of the phase takes every rank out together
        need = 2 * total * ns * 2 + (2 << 30)
        if rank == 0 and torch.cuda.mem_get_info(dev)[0] <= need:
            raise MemoryError("rank 0 has no room for %d bytes of gathered PCM" % need)
        state["pg"] = PipelinedGather(total, ns, GATHER_CHUNK, dev)
        phases.tick()
        state["plans"] = []
        for a_, b_ in state["pg"].edges:
            state["plans"].append(eng.plan((vs.Lane * (b_ - a_)).from_buffer(lanes, a_ * vs.C.sizeof(vs.Lane)), ns))
            phases.tick()

    try:
        phase("gather set-up (buffers, chunk plans)", PHASE_DEADLINE_S, gather_setup)
    except PhaseFailed as exc:
        block["gather"] = {"error": str(exc)}
        return block
    pg, plans = state["pg"], state["plans"]

    def launch_chunk(kk, tensor):
        plans[kk].launch(vs.VS_KIND_SYNTH, tensor.data_ptr(), out_pitch=ns)

    def warm():
        pg.run(launch_chunk, progress=phases.tick)          # warm-up pass (RCCL connections, code objects)
        sync_all()

    def timed_gather():
        g0 = time.perf_counter()
        full_pcm = pg.run(launch_chunk, progress=phases.tick)
        sync_all()
        g = time.perf_counter() - g0
        gt = torch.tensor([g], dtype=torch.float64, device=dev)
        dist.all_reduce(gt, op=dist.ReduceOp.MAX)
        for p_ in plans:
            p_.status()
        return float(gt.item()), full_pcm

    def compare(full_pcm):
        # the un-overlapped comparison: the same chunks, then one gather behind them
        sync_all()
        g0 = time.perf_counter()
        for kk, tns in enumerate(pg.chunks):
            launch_chunk(kk, tns)
        torch.cuda.synchronize(dev)
        phases.tick()
        again = gather_pcm(pg.base, total, dst=0, chunk_rows=GATHER_CHUNK)   # messages of at most 1.4 GB, as the overlapped gather's
        sync_all()
        g2 = time.perf_counter() - g0
        gt = torch.tensor([g2], dtype=torch.float64, device=dev)
        dist.all_reduce(gt, op=dist.ReduceOp.MAX)
        same = bool(torch.equal(again, full_pcm)) if rank == 0 else None
        return float(gt.item()), same

    try:
        phase("gather warm-up (first RCCL exchange)", allow, warm)
        g, full_pcm = phase("gather, overlapped (timed)", allow, timed_gather)
        block["value_with_gather"] = round(total * ns / g / 1e6, 1)
        gather = {"overlapped": True, "chunk_utterances": GATHER_CHUNK, "chunks_per_gpu": len(pg.edges),
                  "ms_compute_and_gather": round(g * 1e3, 3), "bytes_into_rank0": nbytes,
                  "ingress_GB/s": round(nbytes / g / 1e9, 1),
                  "backend": dist.get_backend(),
                  "transport": "torch.distributed send/recv (backend above; nccl = RCCL over xGMI), one grouped receive per chunk on the root"}
        block["gather"] = gather
        phases.partial = dict(block)
        g2, same = phase("gather, compute then gather (comparison)", allow, lambda: compare(full_pcm))
        gather["ms_compute_then_gather"] = round(g2 * 1e3, 3)
        if rank == 0:
            gather["equals_unoverlapped_gather"] = same
        del full_pcm
    except PhaseFailed as exc:
        block.setdefault("gather", {})["error"] = str(exc)
    finally:
        for p_ in plans:
            p_.close()
    return block


if __name__ == "__main__":
    main()
