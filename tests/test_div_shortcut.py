"""The device replaces the reference's "(1.0*random())/RAND_MAX" (IEEE double division by
2^31-1, flowgen_shimmer.c:325,387,398) by one multiply and two fused multiply-adds.  That is
only admissible if it is the SAME double for every possible draw, so it is checked here
exhaustively over all 2^31 draws (CPU, hardware fma; about two seconds), and again on the
device by vs_ctx_selftest() in the GPU suite."""
import os
import subprocess
import sys

SRC = r"""
#include <stdio.h>
#include <math.h>
int main(void) {
  const double d = 2147483647.0, inv = 0x1.00000002p-31;
  long bad = 0;
  if (inv != 1.0 / d) return 2;
  #pragma omp parallel for reduction(+:bad)
  for (long r = 0; r < (1L << 31); r++) {
    double x = (double)r, q0 = x * inv, e = fma(-q0, d, x), q = fma(e, inv, q0);
    if (q != x / d) bad++;
  }
  printf("%ld\n", bad);
  return 0;
}
"""


def test_exhaustive_equality_with_ieee_division(tmp_path):
    c = tmp_path / "divshort.c"
    c.write_text(SRC)
    exe = tmp_path / "divshort"
    flags = ["-O2", "-fopenmp", "-ffp-contract=off"]
    if "fma" in open("/proc/cpuinfo").read():
        flags.append("-mfma")  # hardware fma; without it glibc's exact software fma is used
    subprocess.run(["gcc"] + flags + [str(c), "-o", str(exe), "-lm"], check=True)
    out = subprocess.run([str(exe)], capture_output=True, check=True, timeout=900)
    assert out.stdout.strip() == b"0"
