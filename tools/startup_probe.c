/* Where a drop-in program's start-up goes: the calls `bin/flowgen_shimmer -o f.wav -r 16000 -d 1` makes, one
 * utterance, each timed on the monotonic clock (tools/cli_startup.py builds and runs this on the GPU box).
 *   gcc -O2 -Iinclude -o /tmp/startup_probe tools/startup_probe.c -Lvoice_synth_amd/lib -lvoicesynth -lm -Wl,-rpath,$PWD/voice_synth_amd/lib */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "voice_synth.h"

static double now_ms(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return 1e3 * (double)ts.tv_sec + 1e-6 * (double)ts.tv_nsec;
}

int main(int argc, char **argv)
{
  const int filter = argc > 1 && strcmp(argv[1], "filter") == 0;
  double t0 = now_ms(), t;
  vs_lane lane;
  vs_lane_defaults(&lane);
  lane.fs = 16000;
  lane.seed = 1;
  uint64_t ns = 0;
  vs_num_samples(lane.fs, 1.0f, &ns);
  vs_ctx *ctx = NULL;
  int rc = vs_ctx_create(0, &ctx);
  t = now_ms();
  printf("vs_ctx_create            %8.2f ms (rc %d)\n", t - t0, rc);
  if (rc != VS_OK) return 1;
  int16_t *x = (int16_t *)malloc(ns * 2), *y = (int16_t *)malloc(ns * 2);
  vs_cycle_rec *recs = (vs_cycle_rec *)calloc(200, sizeof(vs_cycle_rec));
  int32_t ncyc = 0;
  for (int rep = 0; rep < 3; rep++) {
    t0 = now_ms();
    rc = vs_source(ctx, &lane, 1, (size_t)ns, x, recs, 200, &ncyc);
    t = now_ms();
    printf("vs_source  (call %d)      %8.2f ms (rc %d, %d cycles)\n", rep + 1, t - t0, rc, ncyc);
    if (filter) {
      t0 = now_ms();
      rc = vs_filter(ctx, &lane, 1, (size_t)ns, x, y);
      t = now_ms();
      printf("vs_filter  (call %d)      %8.2f ms (rc %d)\n", rep + 1, t - t0, rc);
    }
  }
  t0 = now_ms();
  vs_ctx_destroy(ctx);
  printf("vs_ctx_destroy           %8.2f ms\n", now_ms() - t0);
  return 0;
}
