/*
 * oracle/rng_shim.c -- link-time replacement of random()/srandom() for the compiled reference.
 *
 * TEST INFRASTRUCTURE ONLY.  The reference programs seed glibc random() from the wall clock
 * (/root/reference/flowgen_shimmer.c:241, /root/reference/vowel_new.c:234) and have no seed
 * option, so their output is not reproducible as shipped.  Linking this object into the
 * reference build (oracle/Makefile -> oracle/_ref/) makes every random() call return the
 * next draw of the Philox stream defined in oracle/philox.h, keyed by the environment
 * variable VS_SEED (unsigned 64-bit, default 0).  The reference sources are not modified.
 *
 * VS_DRAWLOG, when set to a file name, receives the total number of draws consumed
 * (written at exit) -- used by tests to pin the sequential draw-count behaviour.
 */
#include <stdio.h>
#include <stdlib.h>
#include "philox.h"

static vs_draw_stream g_stream;
static int g_init = 0;

static void vs_shim_report(void)
{
  const char *path = getenv("VS_DRAWLOG");
  if (path && *path) {
    FILE *f = fopen(path, "w");
    if (f) {
      fprintf(f, "%llu\n", (unsigned long long)g_stream.n);
      fclose(f);
    }
  }
}

static void vs_shim_init(void)
{
  const char *s = getenv("VS_SEED");
  unsigned long long seed = s ? strtoull(s, NULL, 0) : 0ull;
  vs_draw_init(&g_stream, (uint64_t)seed);
  if (!g_init) atexit(vs_shim_report);
  g_init = 1;
}

void srandom(unsigned int seed)
{
  (void)seed; /* the clock value the reference passes is ignored on purpose */
  vs_shim_init();
}

long int random(void)
{
  if (!g_init) vs_shim_init();
  return vs_draw_next(&g_stream);
}
