"""Raw HW_ID of every wavefront of the wave-to-SIMD probe (vs_ctx_simd_dealing): which SIMD / CU / SE each wavefront of a
12- and an 8-wavefront workgroup ran on.  Diagnostic for tests/test_gpu_parity.py::test_wavefronts_are_dealt..."""
import ctypes as C, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import voice_synth_amd as vs
lib = vs.load()
eng = vs.Engine(0)
print(eng.device_info(), "dealing:", eng.simd_dealing(), "selftest:", eng.selftest())
lib.vs_launch_simd_probe.restype = C.c_int
lib.vs_launch_simd_probe.argtypes = [C.c_int, C.c_uint, C.c_size_t, C.c_void_p, C.c_void_p]
grid = 256
for waves in (12, 8):
    d = eng.dev_alloc(grid * 16 * 4)
    eng.dev_upload(d, np.zeros(grid * 16, np.uint32))
    rc = lib.vs_launch_simd_probe(waves, grid, 160 * 1024 - 8192, C.c_void_p(d), None)
    eng.synchronize()
    a = eng.dev_download(d, (grid, 16), np.uint32)
    simd = (a >> 4) & 3
    pats = collections.Counter(tuple(int(x) for x in simd[g, :waves]) for g in range(grid))
    print("waves", waves, "rc", rc, "distinct SIMD patterns:", len(pats))
    for pat, cnt in pats.most_common(8):
        print("   ", cnt, pat)
    print("   raw HW_ID of workgroup 0:", [hex(int(x)) for x in a[0, :waves]])
    eng.dev_free(d)
