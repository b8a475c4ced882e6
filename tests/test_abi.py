"""The C-ABI library loads and exports every function include/voice_synth.h declares (no
compute calls: this part of the suite runs without a GPU)."""
import ctypes as C
import os
import re

import voice_synth_amd as vs
from voice_synth_amd import _ffi

HEADER = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "voice_synth.h")


def declared_functions():
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    text = re.sub(r"typedef struct \w+ \{.*?\}\s*\w+;", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vs_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound():
    names = declared_functions()
    assert len(names) >= 32
    lib = C.CDLL(_ffi.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), "libvoicesynth.so does not export %s" % n
        assert n in _ffi.SYMBOLS, "%s is not bound in voice_synth_amd/_ffi.py" % n
    assert set(_ffi.SYMBOLS) == set(names)


def test_struct_sizes_match_the_header():
    assert C.sizeof(_ffi.Lane) == 9 * 4 + 3 * 4 + 8 + 4 * 4 + 41 * 8 + 2 * 4 + 8   # 416 bytes
    assert C.sizeof(_ffi.CycleRec) == 16
    assert C.sizeof(_ffi.DevLane) == 128
    assert C.sizeof(_ffi.Tuning) == 44


def test_version_and_strerror():
    lib = vs.load()
    assert b"gfx950" in lib.vs_version()
    assert lib.vs_strerror(0) == b"ok"
    assert b"no CPU path" in lib.vs_strerror(_ffi.VS_ERR_NODEVICE)


def test_no_silent_fallback_without_device():
    ctx = C.c_void_p()
    rc = vs.load().vs_ctx_create(0, C.byref(ctx))
    assert rc in (_ffi.VS_OK, _ffi.VS_ERR_NODEVICE)
    if rc == _ffi.VS_OK:
        vs.load().vs_ctx_destroy(ctx)
    else:
        assert not ctx.value


def test_product_does_not_reference_the_oracle():
    """No file of the product (package, include/, CLIs, Makefile) may name the oracle, except
    comments that point a reader to it."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bad = []
    for base in ("voice_synth_amd", "include"):
        for dp, _, files in os.walk(os.path.join(root, base)):
            for f in files:
                if not f.endswith((".py", ".c", ".h", ".hip", ".cpp")):
                    continue
                for i, line in enumerate(open(os.path.join(dp, f), errors="replace")):
                    if re.search(r"(import|from|include|dlopen|CDLL).*oracle", line):
                        bad.append("%s:%d" % (os.path.join(dp, f), i + 1))
    assert not bad, bad
    r = os.popen("ldd %s" % _ffi.LIB_PATH).read()
    assert "oracle" not in r


def test_no_trap_instruction_in_the_device_code():
    """Every "cannot happen" of the kernels ends in the launch's error word (VS_ERR_INTERNAL), never in a device
    trap, which would take the caller's HIP context down: the gfx950 listing of the shipped flags (`make isa`)
    holds no s_trap."""
    import shutil
    import subprocess

    import pytest

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("no hipcc on this box")
    subprocess.run(["make", "-s", "isa"], cwd=root, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    text = open(os.path.join(root, "build", "vs_kernels.s")).read()
    assert "vs_synth_ws_kernel" in text
    assert not re.search(r"^\s*s_trap\b", text, flags=re.M)
