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
    """the one-wave kernels park registers in AGPRs (no scratch traffic); nothing else may spill to memory either"""
    for name, r in recs.items():
        assert r["scratch"] == 0, (name, r)
