/*
 * cli_common.h -- shared bits of the two drop-in command-line programs.
 *
 * The programs keep the reference's argv, stdout text, exit codes and .wav layout
 *  and do all sample computation through include/voice_synth.h on the
 * GPU.  Environment (additions; the reference has no such knobs):
 *   VS_SEED        Philox key of the draw stream (default: time(NULL), like srandom(time(NULL)))
 *   VS_WAV_HEADER  44 (default, the ILP32 layout the reference documents) or 72 (what an LP64
 *                  build of the reference writes and reads)
 *   VS_DEVICE      HIP device ordinal (default 0)
 *   VS_ARITH       "exact" (default) or "fma"
 */
#ifndef VS_CLI_COMMON_H
#define VS_CLI_COMMON_H

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "voice_synth.h"

static inline int vs_cli_header_bytes(void)
{
  const char *e = getenv("VS_WAV_HEADER");
  if (e && atoi(e) == 72) return 72;
  return 44;
}

static inline uint64_t vs_cli_seed(void)
{
  const char *e = getenv("VS_SEED");
  if (e && *e) return (uint64_t)strtoull(e, NULL, 0);
  return (uint64_t)time(NULL);
}

static inline int vs_cli_open_ctx(vs_ctx **ctx)
{
  const char *d = getenv("VS_DEVICE");
  int rc = vs_ctx_create(d ? atoi(d) : 0, ctx);
  if (rc != VS_OK) {
    fprintf(stderr, "voice_synth: cannot open GPU device: %s\n", vs_strerror(rc));
    return rc;
  }
  const char *a = getenv("VS_ARITH");
  if (a && strcmp(a, "fma") == 0) vs_ctx_set_arith(*ctx, VS_ARITH_FMA);
  return VS_OK;
}

#endif
