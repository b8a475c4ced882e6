"""What a single-precision filter mode would cost in accuracy (F19).

Emulates on the CPU, with numpy float32 arithmetic, the order-22 recurrence of vowel_new.c:266-289
with fp32 state and fp32 coefficients -- (a) reference order, product and difference rounded
separately; (b) fused multiply-adds in two partial sums, the shape a v_fma_f32 / v_pk_fma_f32
kernel would run -- on the glottal flow of BASELINE config 3 and config 5 lanes, and compares the
int16 output with the exact double-precision oracle.  Output: profiles/r02_fp32_rms_table.txt."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import voice_synth_amd as vs
from voice_synth_amd import configs
from oracle import pyoracle as po

f32, f64 = np.float32, np.float64


def round2int(x):
    x = x.astype(f64)
    dec = x - np.floor(x)
    x = np.where(dec > 0.5, x + 1, x)
    return np.floor(np.clip(x, -32767, 32767)).astype(np.int16)


def fma32(a, b, c):
    # float32 fma through float64: the product of two float32 is exact in float64; the sum rounds
    # to 53 bits and then to 24 (a double rounding that matters for < 1e-8 of the operations)
    return (a.astype(f64) * b.astype(f64) + c.astype(f64)).astype(f32)


def run(lanes, ns, fused):
    L = len(lanes)
    flow = po.source(lanes, ns).astype(f32)
    A = np.array([vs.vowel_coefficients(chr(l.vowel)) if l.vowel else np.array(l.A[:]) for l in lanes]).astype(f32)
    gain = np.array([l.gain for l in lanes], dtype=f32)
    pre = np.array([l.pre_emphasis for l in lanes], dtype=f32)
    y = np.zeros((23, L), dtype=f32)
    out = np.zeros((L, ns), dtype=np.int16)
    for n in range(ns):
        acc = flow[:, n] * gain
        if fused:
            p0, p1 = acc, -(A[:, 2] * y[2])
            for j in range(3, 23):
                if j & 1:
                    p0 = fma32(-A[:, j], y[j], p0)
                else:
                    p1 = fma32(-A[:, j], y[j], p1)
            acc = fma32(-A[:, 1], y[1], p0 + p1)
            o = fma32(-pre, y[1], acc)
        else:
            for j in range(1, 23):
                acc = acc - A[:, j] * y[j]
            o = acc - pre * y[1]
        out[:, n] = round2int(o)
        y[1:] = y[:-1].copy()
        y[1] = acc
    return out


def main():
    rows = []
    for index, n in ((3, 60), (5, 60)):
        specs, fs, dur, label = configs.config_specs(index, n)
        lanes, d = vs.lanes_from_specs(specs)
        ns = 8000
        want = po.synth(lanes, ns)
        for fused, name in ((False, "fp32, reference order (mul, sub)"), (True, "fp32, fused, two partial sums")):
            got = run(lanes, ns, fused)
            d_ = got.astype(f64) - want.astype(f64)
            rows.append((label, name, float(np.mean(d_ != 0)), float(np.abs(d_).max()),
                         float(np.sqrt(np.mean(d_ * d_))), float(np.sqrt(np.mean((d_ / 32768.0) ** 2)))))
        for g in (1.0,):
            for l in range(n):
                lanes[l].gain = g
            want = po.synth(lanes, ns)
            got = run(lanes, ns, True)
            d_ = got.astype(f64) - want.astype(f64)
            rows.append((label + " at gain 1 (no clipping)", "fp32, fused, two partial sums", float(np.mean(d_ != 0)), float(np.abs(d_).max()),
                         float(np.sqrt(np.mean(d_ * d_))), float(np.sqrt(np.mean((d_ / 32768.0) ** 2)))))
    lines = ["fp32 filter emulated on the CPU vs the exact fp64 oracle (int16 output, %d samples per utterance, 60 utterances)" % 8000,
             "%-78s %-34s %10s %8s %10s %12s" % ("workload", "arithmetic", "differing", "max LSB", "RMS LSB", "RMS /32768")]
    for r in rows:
        lines.append("%-78s %-34s %9.2f%% %8.0f %10.3f %12.3e" % (r[0], r[1], 100 * r[2], r[3], r[4], r[5]))
    text = "\n".join(lines)
    print(text)
    open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r02_fp32_rms_table.txt"), "w").write(text + "\n")


main()
