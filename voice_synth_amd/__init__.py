"""voice_synth_amd -- Python test/bench driver over the C ABI of libvoicesynth.so.

The product is the C library (include/voice_synth.h, voice_synth_amd/csrc) and the two drop-in
command-line programs (voice_synth_amd/cli).  This package only wraps the C ABI with numpy
conveniences for tests/ and bench.py.  Nothing here computes samples: every synthesis call
goes to the gfx950 kernels and raises if the library or the device is missing.
"""
import ctypes as C

import numpy as np

from . import _ffi
from ._ffi import (  # noqa: F401
    CycleRec,
    FlowgenCmd,
    Lane,
    Tuning,
    VowelCmd,
    VsError,
    VS_ARITH_EXACT,
    VS_ARITH_FMA,
    VS_ARITH_F32,
    VS_FLAG_JITTER,
    VS_FLAG_NOISE,
    VS_FLAG_SHIMMER,
    VS_KIND_FILTER,
    VS_KIND_SOURCE,
    VS_KIND_SYNTH,
    VS_KERNEL_AUTO,
    VS_KERNEL_SINGLE,
    VS_KERNEL_WS,
    VS_FAULT_WITHHOLD_PROGRESS,
    VS_FAULT_SHORT_COS_ROWS,
    VS_FAULT_SHARD_PREPARE,
    VS_FAULT_SHARD_HANDOVER,
    VS_FAULT_SIMD_DEALING,
    VS_FAULT_REROUND,
    check,
    load,
)

__all__ = [
    "Lane",
    "Engine",
    "Node",
    "Plan",
    "default_lane",
    "parse_flowgen",
    "parse_vowel",
    "lane_from_cli",
    "lanes_from_specs",
    "num_samples",
    "vowel_coefficients",
]


def default_lane():
    lane = Lane()
    check(load().vs_lane_defaults(C.byref(lane)), "vs_lane_defaults")
    return lane


def _argv(args):
    arr = (C.c_char_p * (len(args) + 1))()
    for i, a in enumerate(args):
        arr[i] = a.encode() if isinstance(a, str) else a
    arr[len(args)] = None
    return arr


def parse_flowgen(args):
    """args: flowgen_shimmer's argv[1:] (list of str).  Returns (rc, FlowgenCmd)."""
    cmd = FlowgenCmd()
    argv = _argv(["flowgen_shimmer"] + list(args))
    rc = load().vs_flowgen_parse(len(args) + 1, argv, C.byref(cmd))
    return rc, cmd


def parse_vowel(args):
    cmd = VowelCmd()
    argv = _argv(["vowel"] + list(args))
    rc = load().vs_vowel_parse(len(args) + 1, argv, C.byref(cmd))
    return rc, cmd


def lane_from_cli(flowgen_args, vowel_args, seed=0):
    """One lane from the two reference command lines (without -o / -i file arguments)."""
    rc, fc = parse_flowgen(["-o", "x.wav"] + list(flowgen_args))
    check(rc, "vs_flowgen_parse %r" % (flowgen_args,))
    rc, vc = parse_vowel(["-i", "x.wav", "-o", "y.wav"] + list(vowel_args))
    check(rc, "vs_vowel_parse %r" % (vowel_args,))
    lane = Lane()
    C.memmove(C.byref(lane), C.byref(fc.lane), C.sizeof(Lane))
    lane.gain = vc.gain
    lane.pre_emphasis = vc.pre_emphasis
    lane.vowel = vc.vowel
    lane.out_snr = vc.snr if vc.noise_arg != -1 else 0.0
    lane.seed = seed
    lane.out_seed = seed  # the two reference processes read the same VS_SEED in the shimmed build
    return lane, fc.dur


def lanes_from_specs(specs):
    """specs: iterable of (flowgen_args, vowel_args, seed).  Returns (Lane array, dur).

    Distinct command lines are parsed once by the C parser; lanes that share them only differ
    in their seed."""
    specs = list(specs)
    arr = (Lane * len(specs))()
    cache = {}
    dur = None
    for i, (fa, va, seed) in enumerate(specs):
        key = (tuple(fa), tuple(va))
        if key not in cache:
            cache[key] = lane_from_cli(fa, va, 0)
        proto, d = cache[key]
        if dur is None:
            dur = d
        elif d != dur:
            raise ValueError("all lanes of a batch share one duration")
        C.memmove(C.byref(arr[i]), C.byref(proto), C.sizeof(Lane))
        arr[i].seed = seed
        arr[i].out_seed = seed
    return arr, dur


def num_samples(fs, dur):
    n = C.c_uint64()
    check(load().vs_num_samples(int(fs), float(dur), C.byref(n)), "vs_num_samples")
    return int(n.value)


def row_pitch(n_samples):
    """vs_row_pitch: the row pitch (samples) the kernels' stores like for rows of n_samples"""
    return int(load().vs_row_pitch(int(n_samples)))


def vowel_coefficients(vowel):
    a = (C.c_double * _ffi.VS_NCOEF)()
    v = ord(vowel) if isinstance(vowel, str) else int(vowel)
    check(load().vs_vowel_coefficients(v, a), "vs_vowel_coefficients")
    return np.array(a[:], dtype=np.float64)


def _as_lane_array(lanes):
    if isinstance(lanes, C.Array):
        return lanes
    lanes = list(lanes)
    arr = (Lane * len(lanes))()
    for i, l in enumerate(lanes):
        C.memmove(C.byref(arr[i]), C.byref(l), C.sizeof(Lane))
    return arr


class Engine:
    """A vs_ctx.  Raises VsError(VS_ERR_NODEVICE) when no gfx950 device is usable."""

    def __init__(self, device=0, arith=VS_ARITH_EXACT, stream=None):
        self._lib = load()
        self._ctx = C.c_void_p()
        check(self._lib.vs_ctx_create(int(device), C.byref(self._ctx)), "vs_ctx_create")
        self.set_arith(arith)
        if stream is not None:
            self.set_stream(stream)

    def close(self):
        if self._ctx:
            self._lib.vs_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_arith(self, arith):
        check(self._lib.vs_ctx_set_arith(self._ctx, int(arith)), "vs_ctx_set_arith")
        self.arith = int(arith)

    def set_tuning(self, **kw):
        """vs_ctx_set_tuning(): keyword arguments are vs_tuning fields (kernel=, ring_slots=,
        ready_min=, ws_pairs=, gen_low=, gen_min=, spin_limit=, fault=); no arguments resets.
        Applies to plans made afterwards."""
        if not kw:
            check(self._lib.vs_ctx_set_tuning(self._ctx, None), "vs_ctx_set_tuning")
            return
        t = Tuning()
        for k, v in kw.items():
            setattr(t, k, int(v))
        check(self._lib.vs_ctx_set_tuning(self._ctx, C.byref(t)), "vs_ctx_set_tuning")

    def set_stream(self, hip_stream):
        check(self._lib.vs_ctx_set_stream(self._ctx, C.c_void_p(int(hip_stream))), "vs_ctx_set_stream")

    def synchronize(self):
        check(self._lib.vs_ctx_synchronize(self._ctx), "vs_ctx_synchronize")

    def timer_mark(self, which):
        """vs_ctx_timer_mark(): an event behind what has been enqueued on the context's stream so far (0 = start, 1 = end)"""
        check(self._lib.vs_ctx_timer_mark(self._ctx, int(which)), "vs_ctx_timer_mark")

    def timer_elapsed(self):
        """vs_ctx_timer_elapsed(): waits for mark 1, milliseconds of device time from mark 0 to it"""
        ms = C.c_double()
        check(self._lib.vs_ctx_timer_elapsed(self._ctx, C.byref(ms)), "vs_ctx_timer_elapsed")
        return ms.value

    def selftest(self):
        """(rc, [division shortcut, philox, isqrt, round2int, noise sample, philox2, wave-to-SIMD dealing, output-noise sample] failure counts)"""
        f = (C.c_uint64 * 8)()
        rc = self._lib.vs_ctx_selftest(self._ctx, f)
        return rc, [int(v) for v in f]

    def simd_dealing(self):
        """vs_ctx_simd_dealing(): (wavefront w of a 12-wavefront workgroup ran on SIMD w % 4, ... of an 8-wavefront one)"""
        a, b = C.c_int(), C.c_int()
        check(self._lib.vs_ctx_simd_dealing(self._ctx, C.byref(a), C.byref(b)), "vs_ctx_simd_dealing")
        return bool(a.value), bool(b.value)

    def device_info(self):
        name = C.create_string_buffer(128)
        cu = C.c_int()
        check(self._lib.vs_ctx_device_info(self._ctx, name, 128, C.byref(cu)), "vs_ctx_device_info")
        return name.value.decode(), cu.value

    def device_pci(self):
        """PCI bus id of the device in use ("0000:05:00.0")"""
        buf = C.create_string_buffer(32)
        check(self._lib.vs_ctx_device_pci(self._ctx, buf, 32), "vs_ctx_device_pci")
        return buf.value.decode()

    # ---- host-buffer conveniences ----
    def synth(self, lanes, n_samples):
        arr = _as_lane_array(lanes)
        out = np.empty((len(arr), n_samples), dtype=np.int16)
        check(self._lib.vs_synth(self._ctx, arr, len(arr), n_samples, out.ctypes.data), "vs_synth")
        return out

    def synth_pinned(self, lanes, n_samples):
        """vs_synth() into PINNED host memory from vs_host_alloc(): the finished rows are DMAed
        straight into the destination.  Returns a numpy view; free it with host_free(view)."""
        arr = _as_lane_array(lanes)
        nbytes = len(arr) * n_samples * 2
        p = C.c_void_p()
        check(self._lib.vs_host_alloc(self._ctx, nbytes, C.byref(p)), "vs_host_alloc")
        buf = (C.c_int16 * (len(arr) * n_samples)).from_address(p.value)
        out = np.frombuffer(buf, dtype=np.int16).reshape(len(arr), n_samples)
        try:
            check(self._lib.vs_synth(self._ctx, arr, len(arr), n_samples, p), "vs_synth")
        except Exception:
            self._lib.vs_host_free(self._ctx, p)
            raise
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[out.ctypes.data] = p.value
        return out

    def host_free(self, view):
        p = self._pinned.pop(view.ctypes.data)
        check(self._lib.vs_host_free(self._ctx, C.c_void_p(p)), "vs_host_free")

    def synth_rows(self, lanes, n_samples, fn):
        """vs_synth_rows(): fn(row0, rows_array) is called for every delivered block (from the
        library's delivery threads, possibly concurrently); rows_array is only valid inside fn."""
        arr = _as_lane_array(lanes)

        def tramp(user, row0, rows, ptr):
            try:
                buf = (C.c_int16 * (rows * n_samples)).from_address(ptr)
                return int(fn(int(row0), np.frombuffer(buf, dtype=np.int16).reshape(rows, n_samples)) or 0)
            except Exception:  # pragma: no cover - surfaces as VS_ERR_IO
                return 1

        cb = _ffi.ROWS_CB(tramp)
        check(self._lib.vs_synth_rows(self._ctx, arr, len(arr), n_samples, cb, None), "vs_synth_rows")

    def trim(self):
        check(self._lib.vs_ctx_trim(self._ctx), "vs_ctx_trim")

    def source(self, lanes, n_samples, log_cycles=0):
        arr = _as_lane_array(lanes)
        out = np.empty((len(arr), n_samples), dtype=np.int16)
        if log_cycles:
            recs = (CycleRec * (len(arr) * log_cycles))()
            ncyc = np.zeros(len(arr), dtype=np.int32)
            check(
                self._lib.vs_source(self._ctx, arr, len(arr), n_samples, out.ctypes.data,
                                    C.addressof(recs), log_cycles, ncyc.ctypes.data),
                "vs_source",
            )
            rec_np = np.frombuffer(recs, dtype=[("S", "<f4"), ("x_pow", "<f4"), ("w_pow", "<f4"), ("T", "<i4")])
            return out, rec_np.reshape(len(arr), log_cycles).copy(), ncyc
        check(self._lib.vs_source(self._ctx, arr, len(arr), n_samples, out.ctypes.data, None, 0, None),
              "vs_source")
        return out

    def filter(self, lanes, flow):
        arr = _as_lane_array(lanes)
        flow = np.ascontiguousarray(flow, dtype=np.int16)
        assert flow.ndim == 2 and flow.shape[0] == len(arr)
        out = np.empty_like(flow)
        check(self._lib.vs_filter(self._ctx, arr, len(arr), flow.shape[1], flow.ctypes.data,
                                  out.ctypes.data), "vs_filter")
        return out

    # ---- device-pointer path ----
    def plan(self, lanes, n_samples):
        return Plan(self, lanes, n_samples)

    def dev_alloc(self, nbytes):
        p = C.c_void_p()
        check(self._lib.vs_dev_alloc(self._ctx, nbytes, C.byref(p)), "vs_dev_alloc")
        return p.value

    def dev_free(self, ptr):
        check(self._lib.vs_dev_free(self._ctx, C.c_void_p(ptr)), "vs_dev_free")

    def dev_download(self, ptr, shape, dtype=np.int16):
        out = np.empty(shape, dtype=dtype)
        check(self._lib.vs_dev_download(self._ctx, out.ctypes.data, C.c_void_p(ptr), out.nbytes),
              "vs_dev_download")
        return out

    def dev_upload(self, ptr, array):
        array = np.ascontiguousarray(array)
        check(self._lib.vs_dev_upload(self._ctx, C.c_void_p(ptr), array.ctypes.data, array.nbytes),
              "vs_dev_upload")


class Node:
    """A vs_node: one batch over several devices (or logical shards of one device)."""

    OVERLAP = 1
    STAGE_ALL = 2

    def __init__(self, devices, arith=VS_ARITH_EXACT):
        self._lib = load()
        self._node = C.c_void_p()
        arr = (C.c_int * len(devices))(*devices)
        check(self._lib.vs_node_create(arr, len(devices), C.byref(self._node)), "vs_node_create")
        check(self._lib.vs_node_set_arith(self._node, int(arith)), "vs_node_set_arith")
        self.shards = len(devices)

    def close(self):
        if self._node:
            self._lib.vs_node_destroy(self._node)
            self._node = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    TRANSPORT_PEER, TRANSPORT_RCCL = 0, 1
    LINKS = {0: "self", 1: "peer", 2: "staged", 3: "rccl"}

    def set_transport(self, transport):
        """vs_node_set_transport(): peer DMA (default) or one RCCL communicator owned by the node"""
        check(self._lib.vs_node_set_transport(self._node, int(transport)), "vs_node_set_transport")

    def link(self, shard):
        """how the shard's PCM reaches the root: 'self', 'peer', 'staged' (through host memory), 'rccl'"""
        v = self._lib.vs_node_link(self._node, int(shard))
        if v < 0:
            raise VsError(v, "vs_node_link")
        return self.LINKS[v]

    def rccl_ranks(self, shard):
        """vs_node_rccl_ranks(): ncclCommCount of the shard's communicator (0 on the peer transport)"""
        v = self._lib.vs_node_rccl_ranks(self._node, int(shard))
        if v < 0:
            raise VsError(v, "vs_node_rccl_ranks")
        return v

    def last_rccl_error(self):
        """vs_node_last_rccl_error(): the ncclResult_t of the last failing RCCL call (0: none)"""
        return int(self._lib.vs_node_last_rccl_error(self._node))

    def set_shard_tuning(self, shard, **kw):
        """vs_ctx_set_tuning() on the context that serves one shard (vs_node_ctx); no keywords resets"""
        ctx = C.c_void_p()
        check(self._lib.vs_node_ctx(self._node, int(shard), C.byref(ctx)), "vs_node_ctx")
        if not kw:
            check(self._lib.vs_ctx_set_tuning(ctx, None), "vs_ctx_set_tuning")
            return
        t = Tuning()
        for k, v in kw.items():
            setattr(t, k, int(v))
        check(self._lib.vs_ctx_set_tuning(ctx, C.byref(t)), "vs_ctx_set_tuning")

    def shard_range(self, n_lanes, shard):
        lo, hi = C.c_size_t(), C.c_size_t()
        check(self._lib.vs_node_shard_range(self._node, n_lanes, shard, C.byref(lo), C.byref(hi)), "vs_node_shard_range")
        return lo.value, hi.value

    def synth_gather(self, lanes, n_samples, root_ptr, root_pitch, flags=1):
        """PCM of all shards gathered into device memory of the root; returns (total_ms, max_compute_ms)"""
        arr = _as_lane_array(lanes)
        tot, comp = C.c_double(), C.c_double()
        check(self._lib.vs_node_synth_gather(self._node, arr, len(arr), n_samples, C.c_void_p(root_ptr), root_pitch,
                                             int(flags), C.byref(tot), C.byref(comp)), "vs_node_synth_gather")
        return tot.value, comp.value

    def synth_rows(self, lanes, n_samples, fn):
        arr = _as_lane_array(lanes)

        def tramp(user, row0, rows, ptr):
            try:
                buf = (C.c_int16 * (rows * n_samples)).from_address(ptr)
                return int(fn(int(row0), np.frombuffer(buf, dtype=np.int16).reshape(rows, n_samples)) or 0)
            except Exception:  # pragma: no cover
                return 1

        cb = _ffi.ROWS_CB(tramp)
        check(self._lib.vs_node_synth_rows(self._node, arr, len(arr), n_samples, cb, None), "vs_node_synth_rows")


class Plan:
    """A vs_plan: lane records + cos tables resident on the device; launches are asynchronous."""

    def __init__(self, engine, lanes, n_samples):
        self.engine = engine
        self._lib = engine._lib
        arr = _as_lane_array(lanes)
        self.n_lanes = len(arr)
        self.n_samples = int(n_samples)
        self._plan = C.c_void_p()
        check(self._lib.vs_plan_create(engine._ctx, arr, len(arr), n_samples, C.byref(self._plan)),
              "vs_plan_create")

    def timing(self):
        """(host_ms, upload_ms) of vs_plan_create for this plan"""
        h, u = C.c_double(), C.c_double()
        check(self._lib.vs_plan_timing(self._plan, C.byref(h), C.byref(u)), "vs_plan_timing")
        return h.value, u.value

    def kernel_name(self, kind=VS_KIND_SYNTH):
        buf = C.create_string_buffer(128)
        check(self._lib.vs_plan_kernel_name(self._plan, int(kind), buf, 128), "vs_plan_kernel_name")
        return buf.value.decode()

    def roles(self):
        """vs_plan_roles(): wavefronts per 64 utterances of the fused kind (1 = the one-wave kernel), the layout, and whether
        the plan fell back from three roles to two because the wavefronts are not dealt w % 4"""
        r, l, f = C.c_int(), C.c_int(), C.c_int()
        check(self._lib.vs_plan_roles(self._plan, C.byref(r), C.byref(l), C.byref(f)), "vs_plan_roles")
        return {"roles": r.value, "layout": "spread" if l.value else "role-major", "simd_fallback": bool(f.value)}

    def reseed(self, seeds, out_seeds=None):
        """vs_plan_reseed(): the same utterances with new draws -- seeds[i] (uint64) belongs to lane i of the plan's lane array"""
        seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
        assert seeds.shape == (self.n_lanes,)
        if out_seeds is not None:
            out_seeds = np.ascontiguousarray(out_seeds, dtype=np.uint64)
            assert out_seeds.shape == (self.n_lanes,)
        check(self._lib.vs_plan_reseed(self._plan, seeds.ctypes.data, out_seeds.ctypes.data if out_seeds is not None else None),
              "vs_plan_reseed")

    def info(self):
        lds, wgs, slots = C.c_size_t(), C.c_size_t(), C.c_size_t()
        check(self._lib.vs_plan_info(self._plan, C.byref(lds), C.byref(wgs), C.byref(slots)), "vs_plan_info")
        return {"lds_bytes": lds.value, "workgroups": wgs.value, "ring_slots": slots.value}

    def launch(self, kind, out_ptr, out_pitch=None, in_ptr=None, in_pitch=None, log_ptr=None,
               log_pitch=0, ncyc_ptr=None):
        out_pitch = self.n_samples if out_pitch is None else out_pitch
        in_pitch = self.n_samples if in_pitch is None else in_pitch
        check(
            self._lib.vs_plan_launch(self._plan, int(kind), C.c_void_p(in_ptr), in_pitch,
                                     C.c_void_p(out_ptr), out_pitch, C.c_void_p(log_ptr), log_pitch,
                                     C.c_void_p(ncyc_ptr)),
            "vs_plan_launch",
        )

    def status(self):
        """Waits for the stream; raises VsError(VS_ERR_INTERNAL) if a device-side wait ran out."""
        flags = C.c_int()
        check(self._lib.vs_plan_status(self._plan, C.byref(flags)), "vs_plan_status")
        return flags.value

    def close(self):
        if self._plan:
            self._lib.vs_plan_destroy(self._plan)
            self._plan = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
