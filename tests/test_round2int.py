"""round2int() of vowel_new.c:413-427 against the form the filter super-step uses, ceil(x - 0.5): they
differ exactly on the set the super-step tests for (vs_superstep in csrc/vs_dev_filter.h: high word of a
tiny negative value, or a low word of all ones), where it falls back to the literal form.  numpy float64
arithmetic is IEEE, as the device's; the device repeats the comparison in vs_ctx_selftest() [3]."""
import numpy as np


def literal(x):
    x = x.copy()
    dec = x - np.floor(x)
    x = np.where(dec > 0.5, x + 1.0, x)
    x = np.where(x > 32767, 32767.0, np.where(x < -32767, -32767.0, x))
    return np.floor(x).astype(np.int64)


def half_down(x):
    return np.clip(np.ceil(x - 0.5), -32767, 32767).astype(np.int64)


def flagged(x):
    b = x.view(np.uint64)
    hi = (b >> np.uint64(32)).astype(np.uint32).view(np.int32)
    lo = (b & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    return (hi <= np.uint32(0xBC900000).view(np.int32)) | (lo == np.uint32(0xFFFFFFFF))


def cases():
    rng = np.random.default_rng(1)
    out = []
    for lo_e, hi_e in ((-80, 20), (-1022, -40), (-3, 3)):
        n = 2_000_000
        e = rng.integers(lo_e + 1023, hi_e + 1023, n, dtype=np.uint64)
        m = rng.integers(0, 1 << 52, n, dtype=np.uint64)
        s = rng.integers(0, 2, n, dtype=np.uint64)
        out.append(((s << np.uint64(63)) | (e << np.uint64(52)) | m).view(np.float64))
    k = np.arange(-80000, 80001, dtype=np.float64) * 0.5  # every integer and half integer, 16 doubles each way
    for j in range(-16, 17):
        out.append((k.view(np.int64) + j).view(np.float64))
    for sgn in (1.0, -1.0):  # the doubles around +-2^-e, down to the denormals
        base = (sgn * np.ldexp(1.0, -np.arange(0, 1075))).astype(np.float64)
        for j in range(-16, 17):
            out.append((base.view(np.int64) + j).view(np.float64))
    out.append(np.array([0.0, -0.0, 5e-324, -5e-324, 32766.5, 32767.5, -32767.5, 1e9, -1e9], dtype=np.float64))
    x = np.concatenate(out)
    return x[np.isfinite(x)]


def test_half_down_equals_round2int_outside_the_flagged_set():
    x = cases()
    lit, hd, fl = literal(x), half_down(x), flagged(x)
    assert not np.any((lit != hd) & ~fl)
    # ... and the flagged set is where they do differ: the quirk is real, by one
    d = lit != hd
    assert d.any() and np.all(lit[d] == hd[d] + 1)


def test_the_quirk_values_themselves():
    q = np.array([-2.0 ** -54, -2.0 ** -60, -5e-324, 1 - 2.0 ** -53, 2 - 2.0 ** -52, 16384 - 2.0 ** -39])
    assert flagged(q).all()
    assert list(literal(q)) == [1, 1, 1, 2, 3, 16385]
    assert list(half_down(q)) == [0, 0, 0, 1, 2, 16384]
    # the first double beyond -2^-54 is not in the set
    nxt = np.array([-2.0 ** -54]).view(np.int64) + 1
    assert literal(nxt.view(np.float64))[0] == 0 == half_down(nxt.view(np.float64))[0]
