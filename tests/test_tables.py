"""The ten A(z) tables: product copy == oracle copy == numbers in the reference source text
(the last check only where /root/reference exists, i.e. in the build container)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import voice_synth_amd as vs
from oracle import pyoracle as po

IDS = "aiu1234567"
REF = "/root/reference/vowel_new.c"


def _oracle_table(v):
    lane = vs.default_lane()
    lane.vowel = ord(v)
    a = (C.c_double * 23)()
    assert po.load().vs_oracle_coefficients(C.byref(lane), a) == 0
    return np.array(a[:])


def test_product_equals_oracle():
    for v in IDS:
        assert np.array_equal(vs.vowel_coefficients(v), _oracle_table(v)), v


def test_all_tables_monic_and_stable():
    for v in IDS:
        a = vs.vowel_coefficients(v)
        assert a[0] == 1.0 and len(a) == 23
        radius = np.abs(np.roots(a)).max()
        assert radius < 1.0, (v, radius)  # 0.9747 .. 0.9921


def test_unknown_vowel_rejected():
    a = (C.c_double * 23)()
    for v in "eoxAIU089":
        assert vs.load().vs_vowel_coefficients(ord(v), a) == vs._ffi.VS_ERR_RANGE


@pytest.mark.skipif(not os.path.exists(REF), reason="reference sources only exist in the build container")
def test_tables_match_reference_text():
    src = re.sub(r"/\*.*?\*/", "", open(REF).read(), flags=re.S)
    names = {"a": "A_a", "i": "A_i", "u": "A_u"}
    names.update({str(k): "A_zz%d" % k for k in range(1, 8)})
    for v, n in names.items():
        m = re.search(r"double\s+%s\[22\+1\]\s*=\s*\{(.*?)\};" % n, src, flags=re.S)
        vals = np.array([float(t) for t in m.group(1).replace("\n", " ").split(",")])
        assert np.array_equal(vals, vs.vowel_coefficients(v)), v
