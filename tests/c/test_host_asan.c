/* Host-side C of the product under AddressSanitizer + UBSan (CPU only; the GPU pool has no
 * sanitizers).  Built and run by tests/test_host_sanitizers.py:
 *   gcc -fsanitize=address,undefined -fno-sanitize-recover=all -ffp-contract=off \
 *       tests/c/test_host_asan.c voice_synth_amd/csrc/vs_host.c -Iinclude -lm
 * Exercises both argv parsers with well-formed, malformed and adversarial argument vectors,
 * validation, the sample-count rule and both RIFF header layouts. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "voice_synth.h"

static int fails = 0;
#define CHECK(cond)                                                     \
  do {                                                                  \
    if (!(cond)) {                                                      \
      fprintf(stderr, "FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond);   \
      fails++;                                                          \
    }                                                                   \
  } while (0)

static int fg(int argc, const char **argv, vs_flowgen_cmd *c)
{
  /* the parser may look at argv[argc] (NULL) but never beyond: give it an exact-size heap copy */
  char **v = (char **)malloc((size_t)(argc + 1) * sizeof(char *));
  for (int i = 0; i < argc; i++) v[i] = strdup(argv[i]);
  v[argc] = NULL;
  int rc = vs_flowgen_parse(argc, v, c);
  for (int i = 0; i < argc; i++) free(v[i]);
  free(v);
  return rc;
}
static int vw(int argc, const char **argv, vs_vowel_cmd *c)
{
  char **v = (char **)malloc((size_t)(argc + 1) * sizeof(char *));
  for (int i = 0; i < argc; i++) v[i] = strdup(argv[i]);
  v[argc] = NULL;
  int rc = vs_vowel_parse(argc, v, c);
  for (int i = 0; i < argc; i++) free(v[i]);
  free(v);
  return rc;
}

int main(void)
{
  vs_flowgen_cmd f;
  vs_vowel_cmd v;
  {
    const char *a[] = {"flowgen_shimmer"};
    CHECK(fg((int)(sizeof(a) / sizeof(a[0])), a, &f) == VS_USAGE);
  }
  {
    const char *a[] = {"flowgen_shimmer", "-o"};
    CHECK(fg((int)(sizeof(a) / sizeof(a[0])), a, &f) == VS_USAGE);
  }
  {
    const char *a[] = {"flowgen_shimmer", "-"};
    CHECK(fg((int)(sizeof(a) / sizeof(a[0])), a, &f) == VS_USAGE);
  }
  {
    const char *a[] = {"flowgen_shimmer", "-", "x"};
    CHECK(fg((int)(sizeof(a) / sizeof(a[0])), a, &f) == VS_USAGE); /* "-" alone: argv[i][1] is the terminator */
  }
  {
    const char *a[] = {"flowgen_shimmer", "-o", "x.wav", "-r", "16000", "-d", "1", "-j", "1", "-s", "5.76",
                       "-n", "20", "-z", "0.3", "-l", "0.1", "-k", "0.7", "-c", "0.6", "-f", "200", "-g", "210",
                       "-a", "9000"};
    CHECK(fg((int)(sizeof(a) / sizeof(a[0])), a, &f) == VS_OK);
    CHECK(f.lane.fs == 16000 && f.lane.amp == 9000 && f.wav_arg == 2);
    CHECK(f.lane.flags == (VS_FLAG_JITTER | VS_FLAG_SHIMMER | VS_FLAG_NOISE));
    CHECK(vs_lane_validate(&f.lane) == VS_OK);
  }
  {
    const char *a[] = {"flowgen_shimmer", "-o", "x.wav", "-d", "nonsense", "-r"};
    CHECK(fg((int)(sizeof(a) / sizeof(a[0])), a, &f) == VS_USAGE);
  }
  {
    const char *a[] = {"flowgen_shimmer", "-o", "x.wav", "-a", "99999999999999999999"};
    (void)fg((int)(sizeof(a) / sizeof(a[0])), a, &f); /* atoi overflow is the reference's behaviour too; must not crash */
  }
  {
    const char *a[] = {"flowgen_shimmer", "-o", "", "-r", ""};
    (void)fg((int)(sizeof(a) / sizeof(a[0])), a, &f);
  }
  {
    const char *a[] = {"vowel", "-i", "a.wav", "-o", "b.wav", "-v", "4", "-g", "2.5", "-p", "0.25", "-n", "20"};
    CHECK(vw((int)(sizeof(a) / sizeof(a[0])), a, &v) == VS_OK && v.vowel == '4' && v.noise_arg == 12);
  }
  {
    const char *a[] = {"vowel", "-v"};
    CHECK(vw((int)(sizeof(a) / sizeof(a[0])), a, &v) == VS_USAGE);
  }
  {
    const char *a[] = {"vowel", "-i", "a.wav", "-v", ""};
    CHECK(vw((int)(sizeof(a) / sizeof(a[0])), a, &v) == VS_USAGE); /* empty vowel string: argv[i][0] == 0 */
  }
  {
    const char *a[] = {"vowel", "-i", "a", "-v", "a", "-q", "1"};
    CHECK(vw((int)(sizeof(a) / sizeof(a[0])), a, &v) == VS_USAGE);
  }
  /* validation corner cases */
  {
    vs_lane l;
    vs_lane_defaults(&l);
    CHECK(vs_lane_validate(&l) == VS_OK);
    l.F0 = 0.0f;
    CHECK(vs_lane_validate(&l) != VS_OK); /* fs/F0 = inf must not reach the int conversion */
    vs_lane_defaults(&l);
    l.fs = 2000000000;
    l.F0 = 50.0f;
    l.Fg = 60.0f;
    (void)vs_lane_validate(&l);
    vs_lane_defaults(&l);
    l.vowel = 12345;
    CHECK(vs_lane_validate(&l) == VS_ERR_RANGE);
    CHECK(vs_lane_validate(NULL) == VS_ERR_ARG);
  }
  /* sample count and headers */
  {
    uint64_t n = 0;
    CHECK(vs_num_samples(16000, 1.0f, &n) == VS_OK && n == 16000);
    CHECK(vs_num_samples(0, 1.0f, &n) == VS_ERR_ARG);
    CHECK(vs_num_samples(16000, 1.0f, NULL) == VS_ERR_ARG);
    unsigned char h[72];
    int32_t fs;
    int tag, bits;
    uint64_t db;
    CHECK(vs_wav_header_write(h, 44, 22050, 2.0f) == 44);
    CHECK(vs_wav_header_read(h, 44, &fs, &tag, &bits, &db) == 44 && fs == 22050 && tag == 1 && bits == 16 && db == 88200);
    CHECK(vs_wav_header_write(h, 72, 11025, 0.77f) == 72);
    CHECK(vs_wav_header_read(h, 72, &fs, &tag, &bits, &db) == 72 && fs == 11025);
    CHECK(vs_wav_header_read(h, 10, &fs, &tag, &bits, &db) == VS_ERR_IO); /* truncated */
    CHECK(vs_wav_header_write(h, 50, 16000, 1.0f) == VS_ERR_ARG);
    CHECK(vs_wav_header_write(NULL, 44, 16000, 1.0f) == VS_ERR_ARG);
    double A[VS_NCOEF];
    CHECK(vs_vowel_coefficients('7', A) == VS_OK && A[0] == 1.0);
    CHECK(vs_vowel_coefficients('e', A) == VS_ERR_RANGE);
    CHECK(vs_vowel_coefficients('a', NULL) == VS_ERR_ARG);
    CHECK(vs_vowel_name('3') != NULL && vs_vowel_name('x') == NULL);
    for (int c = -12; c <= 2; c++) CHECK(vs_strerror(c) != NULL);
  }
  if (fails) return 1;
  printf("ok\n");
  return 0;
}
