"""What a block-form fp64 filter would cost in accuracy.

Emulates on the CPU, in numpy float64, the block recurrence a v_mfma_f64_16x16x4_f64 kernel would run:
16 samples per block,  Y[16] = [T | F] . [g*x[16] ; y_prev[22]]  with T the lower-triangular Toeplitz matrix of the
table's impulse response and F its 16 x 22 free-response matrix (both built in long double, rounded once), accumulated
the way the matrix instruction accumulates -- ten chunks of K = 4, one fused multiply-add per term, chunk after chunk
into the same accumulator -- and compares the int16 output with the exact oracle (products and differences rounded one
by one, vowel_new.c:279-281) on the glottal flow of BASELINE config 3 lanes, for all ten tables at gain 10 and gain 1.
ACCURACY is not what rules the block form out (this table); its cost on gfx950 is (profiles/r04_ubench6_fp64_mfma.txt).
Output: profiles/r04_block_form_rms.txt."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import voice_synth_amd as vs  # noqa: E402
from voice_synth_amd import configs  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

L, ORDER = 16, 22
ld = np.longdouble


def round2int(x):
    """vowel_new.c:413-427, vectorised"""
    x = x.astype(np.float64)
    dec = x - np.floor(x)
    x = np.where(dec > 0.5, x + 1, x)
    return np.floor(np.clip(x, -32767, 32767)).astype(np.int16)


def block_matrices(A):
    """T[i][k] = h[i-k] (k <= i), F[i][j-1] = response at block sample i to y[n0 - j] = 1, j = 1..22"""
    a = A.astype(ld)
    h = np.zeros(L, dtype=ld)
    h[0] = 1
    for i in range(1, L):
        h[i] = -sum(a[m] * h[i - m] for m in range(1, min(i, ORDER) + 1))
    T = np.zeros((L, L), dtype=ld)
    for i in range(L):
        for k in range(i + 1):
            T[i, k] = h[i - k]
    F = np.zeros((L, ORDER), dtype=ld)
    for j in range(1, ORDER + 1):
        # state before the block: y[-j] = 1, all other past outputs 0, no input
        past = np.zeros(ORDER + L, dtype=ld)      # past[ORDER + i] = y[i], past[ORDER - m] = y[-m]
        past[ORDER - j] = 1
        for i in range(L):
            past[ORDER + i] = -sum(a[m] * past[ORDER + i - m] for m in range(1, ORDER + 1))
        F[:, j - 1] = past[ORDER:]
    return T.astype(np.float64), F.astype(np.float64)


def fma(a, b, c):
    # float64 fma through long double (64-bit mantissa on x86: the product of two doubles is not exact there, but the
    # error of this emulation, 2^-64 relative per term, is three orders below the 2^-53 roundings it is measuring)
    return (a.astype(ld) * b.astype(ld) + c.astype(ld)).astype(np.float64)


def run_block(flow, A, gain, pre):
    """flow [lanes][n] int16 -> int16, block form"""
    nl, n = flow.shape
    T, F = block_matrices(A)
    M = np.concatenate([T, F, np.zeros((L, 2))], axis=1)          # 16 x 40: ten chunks of K = 4
    out = np.zeros((nl, n), dtype=np.int16)
    state = np.zeros((nl, ORDER))                                   # y[n0-1] .. y[n0-22]
    y1 = np.zeros(nl)
    for n0 in range(0, n, L):
        x = np.zeros((nl, L))
        m = min(L, n - n0)
        x[:, :m] = flow[:, n0:n0 + m].astype(np.float64) * gain     # exact in double (int16 x float gain)
        B = np.concatenate([x, state, np.zeros((nl, 2))], axis=1)  # [lanes][40]
        Y = np.zeros((nl, L))
        for c in range(10):
            for k in range(4):
                col = 4 * c + k
                Y = fma(np.broadcast_to(M[:, col], (nl, L)), np.broadcast_to(B[:, col:col + 1], (nl, L)), Y)
        prev = np.concatenate([y1[:, None], Y[:, :-1]], axis=1)
        o = Y - pre * prev
        out[:, n0:n0 + m] = round2int(o[:, :m])
        y1 = Y[:, L - 1].copy()
        allp = np.concatenate([Y[:, ::-1], state], axis=1)          # newest first
        state = allp[:, :ORDER].copy()
    return out


def main():
    specs, fs, dur, _ = configs.config_specs(3, 40)
    lanes, d = vs.lanes_from_specs(specs)
    ns = vs.num_samples(fs, d)
    flow = po.source(lanes, ns)
    lines = ["block-form fp64 filter (16 samples per block, K = 40 in ten chunks of 4, fma accumulation) against the exact oracle",
             "glottal flow: %d lanes of BASELINE config 3 (jitter 1 %%, shimmer 5.76 %%, glottal noise 20 dB), %d samples each" % (len(lanes), ns),
             "%-6s %-5s %12s %10s %12s %14s %10s" % ("table", "gain", "mismatches", "max |LSB|", "RMS [LSB]", "RMS / 32768", "clipped")]
    worst = 0.0
    for v in "aiu1234567":
        A = vs.vowel_coefficients(v)
        for gain in (10.0, 1.0):
            ls = []
            for l in lanes:
                l2 = vs.Lane.from_buffer_copy(l)
                l2.vowel = ord(v)
                l2.gain = gain
                ls.append(l2)
            want = po.filter(ls, flow)
            got = run_block(flow, A, gain, 1.0)
            dd = got.astype(np.int32) - want.astype(np.int32)
            rms = float(np.sqrt((dd.astype(np.float64) ** 2).mean()))
            worst = max(worst, rms / 32768.0)
            lines.append("%-6s %-5g %12d %10d %12.3e %14.3e %10d" % (v, gain, int((dd != 0).sum()), int(np.abs(dd).max()), rms, rms / 32768.0,
                                                                    int((np.abs(want) == 32767).sum())))
    lines.append("worst RMS / 32768 over the ten tables and both gains: %.3e (north star: <= 1e-5; VS_ARITH_FMA: 0 of 1.05e9 samples differ on config 3)" % worst)
    text = "\n".join(lines)
    print(text)
    with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r04_block_form_rms.txt"), "w") as f:
        f.write(text + "\n")


main()
