"""The five BASELINE.json configurations as reference command lines.

Every lane is described by the two argv lists the REFERENCE programs would be given plus its
Philox seed, so the same description drives (a) oracle/_ref (the compiled reference), (b) the
CPU restatement and (c) the gfx950 engine.  Mappings fixed by the survey:
  * "shimmer 0.5 dB"  ->  -s 5.76      (the reference's -s is percent amplitude, F10)
  * "mixed /a,e,i,o,u/" -> vowel tables 1,2,4,6,7 (there is no e/o letter, F11)
  * 22.05 kHz         ->  omit -r       (an explicit -r 22050 is rejected, F7)
  * F0 sweeps         ->  per-lane -g with Fg = F0*125/120 + 1 (F8)
"""
import numpy as np

MIXED_VOWELS = "12467"

_M0, _M1, _W0, _W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85


def philox4x32_10(ctr, key):
    """Vectorised Philox4x32-10 over numpy uint64 arrays holding 32-bit words (used only to
    derive per-lane PARAMETERS of config 5; the sample path has its own device statement)."""
    c = [np.asarray(x, dtype=np.uint64) & 0xFFFFFFFF for x in ctr]
    k = [np.asarray(x, dtype=np.uint64) & 0xFFFFFFFF for x in key]
    for _ in range(10):
        p0 = np.uint64(_M0) * c[0]
        p1 = np.uint64(_M1) * c[2]
        n0 = ((p1 >> np.uint64(32)) ^ c[1] ^ k[0]) & 0xFFFFFFFF
        n1 = p1 & 0xFFFFFFFF
        n2 = ((p0 >> np.uint64(32)) ^ c[3] ^ k[1]) & 0xFFFFFFFF
        n3 = p0 & 0xFFFFFFFF
        c = [n0, n1, n2, n3]
        k = [(k[0] + np.uint64(_W0)) & 0xFFFFFFFF, (k[1] + np.uint64(_W1)) & 0xFFFFFFFF]
    return c


JPHS_TABLES = "aiu"       # 11 complex pole pairs each
MNV_TABLES = "1234567"    # 10 complex pole pairs + 2 real poles each


def blend_pole_sets(A1, A2, w):
    """Random-formant-set construction of SURVEY.md section 8d (first reading): a convex blend of
    the POLE SETS of two of the reference's tables of the same family, pole by pole in order of
    angle.  Every blended pole lies on the segment between two poles inside the unit circle, so
    the blend is stable by construction.  Returns the 23 denominator coefficients (A[0] = 1)."""
    def split(A):
        r = np.roots(np.asarray(A, dtype=np.float64))
        up = sorted([z for z in r if z.imag > 1e-9], key=lambda z: np.angle(z))
        re = sorted([z.real for z in r if abs(z.imag) <= 1e-9])
        return up, re
    u1, r1 = split(A1)
    u2, r2 = split(A2)
    if len(u1) != len(u2) or len(r1) != len(r2):
        raise ValueError("pole sets of different structure (blend within one table family)")
    poles = []
    for a, b in zip(u1, u2):
        z = w * a + (1.0 - w) * b
        poles += [z, np.conj(z)]
    poles += [w * a + (1.0 - w) * b for a, b in zip(r1, r2)]
    A = np.real(np.poly(poles))
    A[0] = 1.0
    return A


def random_pole_set(order, rng, rmax=0.97):
    """A stable all-pole denominator of the given order (1..40): conjugate pole pairs at random
    angles with radii in [0.6, rmax], one real pole when the order is odd.  Returns order + 1
    coefficients, A[0] = 1."""
    poles = []
    for _ in range(order // 2):
        z = rng.uniform(0.6, rmax) * np.exp(1j * rng.uniform(0.05, np.pi - 0.05))
        poles += [z, np.conj(z)]
    if order % 2:
        poles.append(rng.uniform(-0.9, 0.9))
    A = np.real(np.poly(poles)) if poles else np.array([1.0])
    A[0] = 1.0
    return A


def wide_order_lanes(orders, lane0=0, seed0=1):
    """Config-3-style utterances (16 kHz, 1 s, jitter + shimmer + noise) whose filters are explicit
    coefficient sets of the given orders (vs_lane.order, 1..40): the MAX_ORDER row of SURVEY.md 8(f4)."""
    from . import Lane, lane_from_cli  # noqa: F401  (late: this module is imported by the package)
    rng = np.random.default_rng(40 + lane0)
    fa = ["-r", "16000", "-d", "1", "-j", "1", "-s", "5.76", "-n", "20"]
    arr = (Lane * len(orders))()
    for i, order in enumerate(orders):
        lane, dur = lane_from_cli(fa, ["-v", "a", "-g", "%.2f" % rng.uniform(1, 10), "-p", ["1", "0", "0.37", "0.9"][i % 4]],
                                  seed0 + lane0 + i)
        A = random_pole_set(int(order), rng)
        lane.vowel = 0
        lane.order = int(order)
        for j in range(len(lane.A)):
            lane.A[j] = float(A[j]) if j <= order else 0.0
        arr[i] = lane
    return arr, 16000, 1.0


def config5_blended_lanes(n_lanes, lane0=0, seed0=1):
    """BASELINE config 5 with the FIRST reading of "randomised formant sets": per-utterance F0
    sweep as in config_specs(5, ...) and, per utterance, 23 coefficients of its own -- a convex
    blend of the pole sets of two tables (VS_VOWEL_CUSTOM) -- plus a per-utterance gain.
    Returns (Lane array, fs, dur, label).  Parameters from Philox key (0xC0FFEE, lane)."""
    import ctypes as C

    import voice_synth_amd as vs

    n = int(n_lanes)
    lanes_idx = np.arange(lane0, lane0 + n, dtype=np.int64)
    specs, fs, dur, _ = config_specs(5, n, lane0, seed0)
    arr, d = vs.lanes_from_specs(specs)
    w = philox4x32_10([np.ones(n), np.zeros(n), np.zeros(n), np.zeros(n)],
                      [np.full(n, 0xC0FFEE), lanes_idx.astype(np.uint64)])
    u = [x.astype(np.float64) / 4294967296.0 for x in w]
    tabs = {t: vs.vowel_coefficients(t) for t in JPHS_TABLES + MNV_TABLES}
    cache = {}
    for i in range(n):
        fam = JPHS_TABLES if u[0][i] < 0.3 else MNV_TABLES
        t1 = fam[int(u[1][i] * len(fam)) % len(fam)]
        t2 = fam[int(u[2][i] * len(fam)) % len(fam)]
        wq = round(float(u[3][i]), 3)           # 1000 blend weights: the root finding is cached
        key = (t1, t2, wq)
        if key not in cache:
            cache[key] = blend_pole_sets(tabs[t1], tabs[t2], wq)
        arr[i].vowel = 0                         # VS_VOWEL_CUSTOM
        for j in range(23):
            arr[i].A[j] = float(cache[key][j])
    label = "config5b: batch %d F0 sweep 80-300 Hz, per-utterance blended pole sets + gain, 16 kHz 1 s" % n
    return arr, fs, d, label


def config_specs(index, n_lanes=None, lane0=0, seed0=1, out_noise_db=None):
    """Returns (specs, fs, dur, label).  specs[i] = (flowgen_args, vowel_args, seed) of lane
    lane0+i.  n_lanes defaults to the configuration's full batch.  out_noise_db: every utterance also asks the vowel
    stage for its own noise, "vowel -n <dB>" (vowel_new.c:302-324) -- not part of any BASELINE
    configuration, the label says so."""
    full = {1: 1, 2: 1024, 3: 65536, 4: 262144, 5: 65536}[index]
    n = full if n_lanes is None else int(n_lanes)
    lanes = np.arange(lane0, lane0 + n, dtype=np.int64)
    if index == 1:
        fs, dur = 16000, 1.0
        specs = [(["-r", "16000", "-d", "1"], ["-v", "a"], seed0 + int(l)) for l in lanes]
        label = "config1: 1 utt /a/ 16 kHz 1 s clean"
    elif index == 2:
        fs, dur = 16000, 1.0
        fa = ["-r", "16000", "-d", "1", "-j", "1", "-s", "5.76"]
        specs = [(fa, ["-v", "a"], seed0 + int(l)) for l in lanes]
        label = "config2: batch 1024 /a/ 16 kHz 1 s jitter 1% shimmer 0.5 dB"
    elif index == 3:
        fs, dur = 16000, 1.0
        fa = ["-r", "16000", "-d", "1", "-j", "1", "-s", "5.76", "-n", "20"]
        specs = [(fa, ["-v", MIXED_VOWELS[int(l) % 5]], seed0 + int(l)) for l in lanes]
        label = "config3: batch 65536 mixed vowels 16 kHz 1 s jitter+shimmer+noise"
    elif index == 4:
        fs, dur = 22050, 2.0
        fa = ["-d", "2", "-j", "1", "-s", "5.76", "-n", "20"]
        specs = [(fa, ["-v", MIXED_VOWELS[int(l) % 5]], seed0 + int(l)) for l in lanes]
        label = "config4: batch 262144 mixed vowels 22.05 kHz 2 s (8 GPU)"
    elif index == 5:
        fs, dur = 16000, 1.0
        # per-lane parameters from Philox key (0xC0FFEE, lane): u0 -> F0, u1 -> table, u2 -> gain
        w = philox4x32_10([np.zeros(n), np.zeros(n), np.zeros(n), np.zeros(n)],
                          [np.full(n, 0xC0FFEE), lanes.astype(np.uint64)])
        u0 = w[0].astype(np.float64) / 4294967296.0
        u1 = w[1].astype(np.float64) / 4294967296.0
        u2 = w[2].astype(np.float64) / 4294967296.0
        specs = []
        ids = "aiu1234567"
        for i, l in enumerate(lanes):
            f0 = round(80.0 + 220.0 * float(u0[i]), 2)
            fg = round(f0 * 125.0 / 120.0 + 1.0, 2)
            gain = round(1.0 + 9.0 * float(u2[i]), 2)
            fa = ["-r", "16000", "-d", "1", "-f", "%.2f" % f0, "-g", "%.2f" % fg,
                  "-j", "1", "-s", "5.76", "-n", "20"]
            va = ["-v", ids[int(u1[i] * 10) % 10], "-g", "%.2f" % gain]
            specs.append((fa, va, seed0 + int(l)))
        label = "config5: batch 65536 F0 sweep 80-300 Hz, random vowel table + gain, 16 kHz 1 s"
    else:
        raise ValueError("config index 1..5")
    if out_noise_db is not None:
        specs = [(fa, list(va) + ["-n", "%g" % out_noise_db], seed) for fa, va, seed in specs]
        label += " + vowel -n %g" % out_noise_db
    return specs, fs, dur, label


def shard_range(n_lanes, rank, world):
    """Contiguous lane block [lo, hi) of `rank` of `world`; blocks differ by at most one lane.  The same cut
    as vs_node_shard_range() in csrc/vs_node.c (tests/test_gpu_node.py holds the two together)."""
    base, rem = divmod(int(n_lanes), int(world))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi
