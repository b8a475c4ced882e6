"""Register budget of the wave-specialised kernels, from the compiler's own remarks with the SHIPPED flags
(tools/kernel_resources.py = `make resources`; hipcc cross-compiles gfx950 without a GPU, ~20 s).

The three-role kernel runs three wavefronts per SIMD: 512 VGPRs / 3 = 168 (allocation granule 8).  One register more
and the compiler spills INSIDE the filter wavefront's super-step loop -- scratch loads on the wavefront the launch
waits for -- without any test noticing (round 4 shipped 12 bytes per lane of it in two instantiations).  The two-role
kernels run at occupancy 2 (<= 256).  This test fails on any scratch in vs_synth_ws_kernel<*,*,*> and on a VGPR count
that loses a resident wavefront."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.fixture(scope="module")
def recs():
    if not (os.path.exists(HIPCC) or shutil.which("hipcc")):
        pytest.skip("no hipcc here")
    import kernel_resources
    return {r["name"]: r for r in kernel_resources.resources()}


def test_every_wave_specialised_instantiation_is_built(recs):
    for arith in (0, 1):
        for pre1 in ("true", "false"):
            for roles in (2, 3):
                assert "vs_synth_ws_kernel<%d, %s, %d>" % (arith, pre1, roles) in recs


def test_three_role_kernels_fit_three_wavefronts_per_simd_without_scratch(recs):
    for name, r in recs.items():
        if name.startswith("vs_synth_ws_kernel<") and name.endswith(", 3>"):
            assert r["scratch"] == 0 and r.get("vgpr_spill", 0) == 0, (name, r)
            assert r["vgprs"] <= 168 and r["occupancy"] >= 3, (name, r)
            assert r.get("agprs", 0) == 0, (name, r)


def test_two_role_kernels_keep_two_wavefronts_per_simd_without_scratch(recs):
    for name, r in recs.items():
        if name.startswith("vs_synth_ws_kernel<") and name.endswith(", 2>"):
            assert r["scratch"] == 0 and r.get("vgpr_spill", 0) == 0, (name, r)
            assert r["occupancy"] >= 2, (name, r)


def test_no_kernel_of_the_library_uses_scratch(recs):
    """the one-wave kernels park registers in AGPRs (no scratch traffic); nothing else may spill to memory either --
    but for the exact three-role kernels that take vowel -n's frame powers along (next test)"""
    for name, r in recs.items():
        if not name.startswith("vs_synth_ws_pow_kernel<0"):
            assert r["scratch"] == 0, (name, r)


def test_frame_power_kernels_keep_their_wavefronts(recs):
    """vs_synth_ws_pow_kernel<*,*,*>: the same kernels with one more running sum per lane.  The exact three-role ones are
    at 168 registers already and park a few launch constants in scratch (one 8-byte reload per super-step of 24 samples,
    the rest outside the loop: measured, not free -- DESIGN.md); no instantiation may lose a resident wavefront, and the
    scratch must stay what it is"""
    for arith in (0, 1):
        for pre1 in ("true", "false"):
            for roles in (2, 3):
                name = "vs_synth_ws_pow_kernel<%d, %s, %d>" % (arith, pre1, roles)
                r = recs[name]
                assert r["occupancy"] >= (3 if roles == 3 else 2), (name, r)
                assert r.get("agprs", 0) == 0, (name, r)
                assert r["scratch"] <= (64 if (arith == 0 and roles == 3) else 0), (name, r)


def test_superstep_loops_are_in_step_with_the_8_byte_grid():
    """A wavefront alone on its SIMD pays about one cycle for every 8-byte instruction that straddles an 8-byte boundary
    (profiles/r05_loop_alignment.txt: config 4's shard 4.58 against 4.82 ms, config 2 1.58 against 1.80, for one 4-byte
    instruction more in front of the loop).  The super-step keeps its chunks in step by itself (.p2align 3 behind each
    chunk's s_waitcnt; the tolerance mode keeps the sign of its taps in the registers, so that its 21 multiply-adds per
    sample are 4-byte V_FMAC_F64 with nothing to straddle); this looks at the BUILT code object: in every wave-specialised
    instantiation at most 160 of the super-step's ~1160 8-byte instructions (exact; ~190 in the tolerance mode) sit off
    the 8-byte grid -- the parity that cost 5 % had 750 -- and the alignment costs at most 8 s_nop per 24 samples."""
    obj = os.path.join(ROOT, "voice_synth_amd", "csrc", "vs_kernels.o")
    if not os.path.exists(obj):
        pytest.skip("the library is not built")
    import isa_align
    for arith in (0, 1):
        for pre1 in (0, 1):
            for roles in (2, 3):
                sym = "_Z18vs_synth_ws_kernelILi%dELb%dELi%dEEv12VsKernelArgs" % (arith, pre1, roles)
                n, on, off, nops = isa_align.superstep_alignment(obj, sym)
                assert off <= 160, (sym, n, on, off)
                assert nops <= (8 if arith == 0 else 20), (sym, n, nops)   # (the compiler's own hazard s_nops in the fma loop: 14)
