"""The generator's short noise sequence replaces the reference's

    w[i] = (signed short)ceil(((1.0*random())/RAND_MAX)*NoiseDistWidth - NoiseDistWidth/2.)

(flowgen_shimmer.c:387, 398: an IEEE division, a product, a difference, ceil, a cast) and the
addition to (short)DC behind it by ONE fused multiply-add and a truncating conversion whose low 16
bits are the sample (vs_noise_sample() in voice_synth_amd/csrc/vs_dev_generator.h; the proof is in its
comment).  That is only admissible if it is the SAME integer
for every possible draw and every width the short sequence accepts (N <= 65534), so it is
checked here on the CPU -- exhaustively over all 2^31 draws for a set of widths that includes
the typical ones, the parities, the powers of two and the upper limit, and for every width at
the edge draws -- and again on the device by vs_ctx_selftest() in the GPU suite (16 widths)."""
import subprocess

SRC = r"""
#include <stdio.h>
#include <stdint.h>
#include <math.h>
static inline int w_literal(uint32_t r, int N) {
  return (int)(int16_t)(int)ceil(((1.0 * (double)r) / 2147483647.0) * (double)N - (double)N / 2.);
}
/* the device's form: low 16 bits of trunc(fma(r, N*inv, (short)DC + 65536 - N/2 + 1 - 1e-10)) */
static inline int x_short(uint32_t r, int N, int dc) {
  const double c = (double)N * 0x1.00000002p-31;
  const double k2 = ((double)(dc + 65536) - (double)N / 2.0 + 1.0) - 1e-10;
  return (int)(int16_t)(int)fma((double)r, c, k2);
}
static inline int w_short(uint32_t r, int N) { return x_short(r, N, 0); }
int main(void) {
  const int widths[] = {1, 2, 3, 2801, 4096, 65533, 65534, 45533, 32768};
  const int dcs[]    = {0, 1, -1,  0,    3,    0,     0,     9830, -16000};
  long bad = 0;
  for (unsigned wi = 0; wi < sizeof widths / sizeof widths[0]; wi++) {
    const int N = widths[wi], dc = dcs[wi];
    #pragma omp parallel for reduction(+:bad)
    for (long r = 0; r < (1L << 31); r++)
      if (x_short((uint32_t)r, N, dc) != (int)(int16_t)(dc + w_literal((uint32_t)r, N))) bad++;
  }
  const uint32_t edge[] = {0u, 1u, 2u, 3u, 0x3FFFFFFFu, 0x40000000u, 0x40000001u, 0x7FFFFFFCu,
                           0x7FFFFFFDu, 0x7FFFFFFEu, 0x7FFFFFFFu, 0x12345678u, 0x2AAAAAAAu, 0x55555555u};
  #pragma omp parallel for reduction(+:bad)
  for (int N = 0; N <= 65534; N++)
    for (unsigned e = 0; e < sizeof edge / sizeof edge[0]; e++)
      if (w_short(edge[e], N) != w_literal(edge[e], N)) bad++;
  /* a pseudo-random sweep over (r, N) pairs */
  uint64_t s = 88172645463325252ull;
  for (long k = 0; k < 200000000L; k++) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    const uint32_t r = (uint32_t)(s >> 33);
    const int N = (int)((s & 0xFFFF) % 65535u);
    const int dc = (int)((s >> 16) % 19661u) - 9830;   /* DC flow up to 0.3 * 32767 either way */
    if ((N >> 1) + 2 + (dc < 0 ? -dc : dc) > 32767) continue; /* outside the short sequence's domain */
    if (x_short(r, N, dc) != dc + w_literal(r, N)) bad++;
  }
  printf("%ld\n", bad);
  return 0;
}
"""


def test_one_fma_noise_sample_equals_the_reference_expression(tmp_path):
    c = tmp_path / "noiseshort.c"
    c.write_text(SRC)
    exe = tmp_path / "noiseshort"
    flags = ["-O2", "-fopenmp", "-ffp-contract=off"]
    if "fma" in open("/proc/cpuinfo").read():
        flags.append("-mfma")  # hardware fma; without it glibc's exact software fma is used
    subprocess.run(["gcc"] + flags + [str(c), "-o", str(exe), "-lm"], check=True)
    out = subprocess.run([str(exe)], capture_output=True, check=True, timeout=1500)
    assert out.stdout.strip() == b"0"
