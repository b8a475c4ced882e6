"""Philox4x32-10 known answers (Random123 kat_vectors) and the draw
contract: draw n = word (n & 3) of block (n >> 2), shifted right by one."""
import numpy as np

from oracle import pyoracle as po
from voice_synth_amd import configs

KAT = [
    ((0, 0, 0, 0), (0, 0), (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
    ((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2, (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
    ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0),
     (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)),
]


def test_oracle_philox_kat():
    for ctr, key, want in KAT:
        assert tuple(po.philox(ctr, key)) == want


def test_numpy_philox_kat():
    for ctr, key, want in KAT:
        got = configs.philox4x32_10([np.array([c]) for c in ctr], [np.array([k]) for k in key])
        assert tuple(int(g[0]) for g in got) == want


def test_draw_contract():
    seed = 0x0123456789ABCDEF
    key = (seed & 0xFFFFFFFF, seed >> 32)
    for n in (0, 1, 2, 3, 4, 5, 1000, 123457, 2**32 + 5):
        blk = po.philox((((n >> 2) & 0xFFFFFFFF), (n >> 2) >> 32, 0, 0), key)
        assert po.draw(seed, n) == blk[n & 3] >> 1
        assert 0 <= po.draw(seed, n) <= 2**31 - 1
