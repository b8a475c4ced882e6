/* Where vs_plan_create's host time goes, piece by piece, without a device (links vs_planhost.o + vs_host.o):
 *   gcc -O2 -Iinclude -Ivoice_synth_amd/csrc tools/planhost_time.c voice_synth_amd/csrc/vs_planhost.o voice_synth_amd/csrc/vs_host.o -lm -lpthread -o /tmp/planhost_time
 * config-3-like batch (homogeneous period) and a config-5-like one (F0 sweep: the order pass runs). */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include "voice_synth.h"
#include "vs_device.h"
#include "vs_planhost.h"
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; }
static void *nop(void *a) { return a; }
#define REP(label, stmt)                                            \
  do {                                                              \
    double best = 1e9;                                              \
    for (int r = 0; r < 7; r++) { double t0 = now(); stmt; double t1 = now(); if (t1 - t0 < best) best = t1 - t0; } \
    printf("  %-44s %.3f ms\n", label, best);                       \
  } while (0)
int main(void)
{
  const size_t n = 65536;
  vs_lane *lanes = malloc(n * sizeof(vs_lane));
  VsDevLane *dl = malloc(n * sizeof(VsDevLane));
  printf("cpus online %ld\n", sysconf(_SC_NPROCESSORS_ONLN));
  VsPlanWs *ws = vs_planws_create();
  for (int sweep = 0; sweep < 2; sweep++) {
    for (size_t i = 0; i < n; i++) {
      vs_lane_defaults(&lanes[i]);
      lanes[i].vowel = "12467"[i % 5];
      lanes[i].flags |= VS_FLAG_JITTER | VS_FLAG_SHIMMER | VS_FLAG_NOISE;
      lanes[i].jitter = 0.01f; lanes[i].shimmer = 0.1f; lanes[i].noise = 100.0f; lanes[i].seed = i; lanes[i].fs = 16000;
      if (sweep) { lanes[i].F0 = 80.0f + 220.0f * (float)((i * 2654435761u) & 0xFFFF) / 65536.0f; lanes[i].Fg = lanes[i].F0 * 125.0f / 120.0f + 1.0f; }
    }
    printf(sweep ? "F0 sweep (config 5 like)\n" : "homogeneous (config 3 like)\n");
    VsBatchStats st; int rc = 0; uint32_t *order = NULL;
    { VsDevLane *out; REP("vs_expand_all_ordered_ws (what the plan calls)", rc |= vs_expand_all_ordered_ws(ws, lanes, n, 0, &out, NULL, &st)); }
    REP("vs_expand_all_ordered (own workspace + copy)", rc |= vs_expand_all_ordered(lanes, dl, n, NULL, &st));
    REP("vs_kernel_order (keys + radix sort, serial)", { rc |= vs_kernel_order(lanes, n, &order); free(order); });
    REP("vs_expand_all_stats (threads)", rc |= vs_expand_all_stats(lanes, dl, n, 0, &st));
    REP("vs_expand_lane, serial loop", for (size_t i = 0; i < n; i++) rc |= vs_expand_lane(&lanes[i], (int)i, &dl[i]));
    REP("vs_lane_validate, serial loop", for (size_t i = 0; i < n; i++) rc |= vs_lane_validate(&lanes[i]));
    REP("touch fs + seed of every lane, serial", { volatile long s = 0; for (size_t i = 0; i < n; i++) s += lanes[i].fs + (long)lanes[i].seed; });
    { double *taps; size_t rows; REP("vs_tap_table_build", { rc |= vs_tap_table_build(lanes, dl, n, 0, &taps, &rows); free(taps); }); }
    printf("  rc %d\n", rc);
  }
  { pthread_t th[15]; REP("15 x pthread_create + join of nothing", { for (int t = 0; t < 15; t++) pthread_create(&th[t], NULL, nop, NULL); for (int t = 0; t < 15; t++) pthread_join(th[t], NULL); }); }
  return 0;
}
