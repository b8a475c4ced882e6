"""ctypes access to the parity checker.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
It offers
  * the CPU restatement (oracle/liboracle.so, built from vs_oracle.c), and
  * run_reference(): the REFERENCE ITSELF (oracle/_ref/flowgen_shimmer, oracle/_ref/vowel,
    compiled from /root/reference with the Philox random() shim), driven through its own
    command line and .wav files.
"""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

from voice_synth_amd._ffi import CycleRec, Lane

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")
REF_DIR = os.path.join(_HERE, "_ref")

_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("oracle/liboracle.so is not built: run `make -C oracle`")
    lib = C.CDLL(LIB_PATH)
    P = C.POINTER
    vp = C.c_void_p
    lib.vs_oracle_source.restype = C.c_int
    lib.vs_oracle_source.argtypes = [P(Lane), C.c_size_t, vp, vp, C.c_size_t, P(C.c_int32), P(C.c_uint64)]
    lib.vs_oracle_filter.restype = C.c_int
    lib.vs_oracle_filter.argtypes = [P(Lane), C.c_size_t, vp, vp]
    lib.vs_oracle_coefficients.restype = C.c_int
    lib.vs_oracle_coefficients.argtypes = [P(Lane), P(C.c_double)]
    lib.vs_oracle_source_batch.restype = C.c_int
    lib.vs_oracle_source_batch.argtypes = [P(Lane), C.c_size_t, C.c_size_t, vp, C.c_int]
    lib.vs_oracle_filter_batch.restype = C.c_int
    lib.vs_oracle_filter_batch.argtypes = [P(Lane), C.c_size_t, C.c_size_t, vp, vp, C.c_int]
    lib.vs_oracle_synth_batch.restype = C.c_int
    lib.vs_oracle_synth_batch.argtypes = [P(Lane), C.c_size_t, C.c_size_t, vp, C.c_int]
    lib.vs_oracle_philox.restype = None
    lib.vs_oracle_philox.argtypes = [P(C.c_uint32), P(C.c_uint32), P(C.c_uint32)]
    lib.vs_oracle_draw.restype = C.c_long
    lib.vs_oracle_draw.argtypes = [C.c_uint64, C.c_uint64]
    lib.vs_oracle_round2int.restype = C.c_int16
    lib.vs_oracle_round2int.argtypes = [C.c_double]
    lib.vs_oracle_truncate.restype = C.c_int16
    lib.vs_oracle_truncate.argtypes = [C.c_float]
    lib.vs_oracle_max_threads.restype = C.c_int
    lib.vs_oracle_max_threads.argtypes = []
    _lib = lib
    return lib


def _arr(lanes):
    if isinstance(lanes, C.Array):
        return lanes
    lanes = list(lanes)
    arr = (Lane * len(lanes))()
    for i, l in enumerate(lanes):
        C.memmove(C.byref(arr[i]), C.byref(l), C.sizeof(Lane))
    return arr


def _ok(rc, where):
    if rc != 0:
        raise RuntimeError("%s failed with %d" % (where, rc))


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    load().vs_oracle_philox(c, k, o)
    return [int(v) for v in o]


def draw(seed, n):
    return int(load().vs_oracle_draw(seed, n))


def source_one(lane, n_samples, max_recs=0):
    """Returns (flow, recs, ncyc, ndraws) for one lane."""
    flow = np.empty(n_samples, dtype=np.int16)
    ncyc = C.c_int32()
    nd = C.c_uint64()
    recs = (CycleRec * max(1, max_recs))()
    _ok(load().vs_oracle_source(C.byref(lane), n_samples, flow.ctypes.data,
                                C.addressof(recs) if max_recs else None, max_recs,
                                C.byref(ncyc), C.byref(nd)), "vs_oracle_source")
    rec_np = np.frombuffer(recs, dtype=[("S", "<f4"), ("x_pow", "<f4"), ("w_pow", "<f4"), ("T", "<i4")]).copy()
    return flow, rec_np[: min(ncyc.value, max_recs)], ncyc.value, nd.value


def source(lanes, n_samples, threads=0):
    arr = _arr(lanes)
    out = np.empty((len(arr), n_samples), dtype=np.int16)
    _ok(load().vs_oracle_source_batch(arr, len(arr), n_samples, out.ctypes.data,
                                      threads or max_threads()), "vs_oracle_source_batch")
    return out


def filter(lanes, flow, threads=0):  # noqa: A001
    arr = _arr(lanes)
    flow = np.ascontiguousarray(flow, dtype=np.int16)
    out = np.empty_like(flow)
    _ok(load().vs_oracle_filter_batch(arr, len(arr), flow.shape[1], flow.ctypes.data,
                                      out.ctypes.data, threads or max_threads()), "vs_oracle_filter_batch")
    return out


def synth(lanes, n_samples, threads=0):
    arr = _arr(lanes)
    out = np.empty((len(arr), n_samples), dtype=np.int16)
    _ok(load().vs_oracle_synth_batch(arr, len(arr), n_samples, out.ctypes.data,
                                     threads or max_threads()), "vs_oracle_synth_batch")
    return out


def max_threads():
    return int(load().vs_oracle_max_threads())


def round2int(x):
    return int(load().vs_oracle_round2int(float(x)))


def truncate(x):
    return int(load().vs_oracle_truncate(float(x)))


# ---------------------------------------------------------------------------------------------
# the reference itself
# ---------------------------------------------------------------------------------------------
def have_reference():
    return all(os.path.exists(os.path.join(REF_DIR, n)) for n in ("flowgen_shimmer", "vowel"))


def _payload(path):
    """PCM payload of a .wav written by the LP64 reference build (72-byte header)."""
    raw = open(path, "rb").read()
    return np.frombuffer(raw[72:], dtype="<i2").copy(), raw[:72]


def run_reference(flowgen_args, vowel_args, seed=0, opt=""):
    """Runs oracle/_ref/flowgen_shimmer then oracle/_ref/vowel in a scratch directory with
    short file names.  Returns dict(flow, pcm, flow_stdout, vowel_stdout, ndraws,
    flow_header).  vowel_args None skips the filter stage."""
    if not have_reference():
        raise RuntimeError("oracle/_ref is not built")
    env = dict(os.environ)
    env["VS_SEED"] = str(int(seed))
    suffix = "_O2" if opt == "O2" else ""
    with tempfile.TemporaryDirectory(prefix="vsref") as d:
        env["VS_DRAWLOG"] = os.path.join(d, "draws.txt")
        fg = subprocess.run([os.path.join(REF_DIR, "flowgen_shimmer" + suffix), "-o", "g.wav"] + list(flowgen_args),
                            cwd=d, env=env, capture_output=True, check=True)
        flow, hdr = _payload(os.path.join(d, "g.wav"))
        ndraws = int(open(env["VS_DRAWLOG"]).read().strip())
        res = {"flow": flow, "flow_stdout": fg.stdout, "ndraws": ndraws, "flow_header": hdr}
        if vowel_args is not None:
            vw = subprocess.run([os.path.join(REF_DIR, "vowel" + suffix), "-i", "g.wav", "-o", "o.wav"] + list(vowel_args),
                                cwd=d, env=env, capture_output=True, check=True)
            pcm, _ = _payload(os.path.join(d, "o.wav"))
            res["pcm"] = pcm
            res["vowel_stdout"] = vw.stdout
        return res
