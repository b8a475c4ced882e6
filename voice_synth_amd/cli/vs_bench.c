/*
 * vs_bench -- the throughput of the fused source->filter path from plain C, no Python:
 *
 *     vs_bench [--lanes N] [--steps K] [--warmup W] [--arith exact|fma] [--host]
 *
 * Workload: BASELINE.json configs[2] -- N utterances (default 65536), vowel table "12467"[lane % 5],
 * 16 kHz, 1 s, jitter 1 %, shimmer 0.5 dB (-s 5.76), glottal noise 20 dB, lane key = 1 + lane --
 * built from the reference's own command lines through vs_flowgen_parse()/vs_vowel_parse(), i.e.
 * exactly what voice_synth_amd/configs.py describes for bench.py.  Timed: K launches of one plan
 * into a device buffer (host clock around launch ... vs_plan_status, which waits), after W warm-up
 * launches.  --host times vs_synth() into a pinned host buffer instead (PCIe included).
 * One line of JSON on stdout.  bench.py remains the driver's benchmark; this is the same
 * measurement for a maintainer who only has the C side.
 */
#include <time.h>

#include "cli_common.h"

static double now_s(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int main(int argc, char **argv)
{
  size_t n_lanes = 65536;
  int steps = 20, warmup = 5, host = 0;
  const char *arith = "exact";
  for (int i = 1; i < argc; i++) {
    if (!strcmp(argv[i], "--lanes") && i + 1 < argc) n_lanes = (size_t)strtoull(argv[++i], NULL, 0);
    else if (!strcmp(argv[i], "--steps") && i + 1 < argc) steps = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--warmup") && i + 1 < argc) warmup = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--arith") && i + 1 < argc) arith = argv[++i];
    else if (!strcmp(argv[i], "--host")) host = 1;
    else {
      fprintf(stderr, "usage: vs_bench [--lanes N] [--steps K] [--warmup W] [--arith exact|fma] [--host]\n");
      return 1;
    }
  }
  if (n_lanes == 0 || steps < 1 || warmup < 0) return 1;

  /* the lane records, from the reference's command lines */
  char *fg_argv[] = {"flowgen_shimmer", "-o", "x.wav", "-r", "16000", "-d", "1", "-j", "1", "-s", "5.76", "-n", "20", NULL};
  vs_flowgen_cmd fc;
  if (vs_flowgen_parse(13, fg_argv, &fc) != VS_OK) return 1;
  uint64_t n_samples = 0;
  vs_num_samples(fc.lane.fs, fc.dur, &n_samples);
  vs_lane *lanes = (vs_lane *)malloc(n_lanes * sizeof(vs_lane));
  if (!lanes) return 1;
  for (size_t l = 0; l < n_lanes; l++) {
    char v[2] = {"12467"[l % 5], 0};
    char *vw_argv[] = {"vowel", "-i", "x.wav", "-o", "y.wav", "-v", v, NULL};
    vs_vowel_cmd vc;
    if (vs_vowel_parse(7, vw_argv, &vc) != VS_OK) return 1;
    lanes[l] = fc.lane;
    lanes[l].gain = vc.gain;
    lanes[l].pre_emphasis = vc.pre_emphasis;
    lanes[l].vowel = vc.vowel;
    lanes[l].seed = 1 + (uint64_t)l;
    lanes[l].out_seed = lanes[l].seed;
  }

  vs_ctx *ctx = NULL;
  if (vs_cli_open_ctx(&ctx) != VS_OK) return 1;
  if (!strcmp(arith, "fma")) vs_ctx_set_arith(ctx, VS_ARITH_FMA);
  int rc = VS_OK;
  double t = 0.0;
  char kernel[160] = "";
  if (host) {
    void *pcm = NULL;
    rc = vs_host_alloc(ctx, n_lanes * n_samples * sizeof(int16_t), &pcm);
    for (int k = 0; rc == VS_OK && k < warmup; k++) rc = vs_synth(ctx, lanes, n_lanes, n_samples, (int16_t *)pcm);
    const double t0 = now_s();
    for (int k = 0; rc == VS_OK && k < steps; k++) rc = vs_synth(ctx, lanes, n_lanes, n_samples, (int16_t *)pcm);
    t = now_s() - t0;
    snprintf(kernel, sizeof(kernel), "vs_synth (chunked, pinned destination)");
    if (pcm) vs_host_free(ctx, pcm);
  } else {
    const size_t pitch = (n_samples + 7) & ~(size_t)7;
    vs_plan *plan = NULL;
    void *out = NULL;
    rc = vs_plan_create(ctx, lanes, n_lanes, n_samples, &plan);
    if (rc == VS_OK) rc = vs_dev_alloc(ctx, n_lanes * pitch * sizeof(int16_t), &out);
    for (int k = 0; rc == VS_OK && k < warmup; k++)
      rc = vs_plan_launch(plan, VS_KIND_SYNTH, NULL, 0, (int16_t *)out, pitch, NULL, 0, NULL);
    if (rc == VS_OK) rc = vs_plan_status(plan, NULL);
    const double t0 = now_s();
    for (int k = 0; rc == VS_OK && k < steps; k++)
      rc = vs_plan_launch(plan, VS_KIND_SYNTH, NULL, 0, (int16_t *)out, pitch, NULL, 0, NULL);
    if (rc == VS_OK) rc = vs_plan_status(plan, NULL); /* waits for the stream; non-zero if a device-side wait ran out */
    t = now_s() - t0;
    if (plan) vs_plan_kernel_name(plan, VS_KIND_SYNTH, kernel, sizeof(kernel));
    if (out) vs_dev_free(ctx, out);
    if (plan) vs_plan_destroy(plan);
  }
  if (rc != VS_OK) {
    fprintf(stderr, "vs_bench: %s\n", vs_strerror(rc));
    vs_ctx_destroy(ctx);
    free(lanes);
    return 1;
  }
  const double samples = (double)n_lanes * (double)n_samples;
  printf("{\"metric\": \"synthesised Msamples/s\", \"value\": %.1f, \"unit\": \"Msamples/s\", \"ms_per_step\": %.4f, "
         "\"steps\": %d, \"warmup\": %d, \"utterances\": %zu, \"samples_per_utterance\": %llu, \"arith\": \"%s\", "
         "\"path\": \"%s\", \"GB_per_s_of_pcm\": %.1f}\n",
         samples * steps / t / 1e6, t / steps * 1e3, steps, warmup, n_lanes, (unsigned long long)n_samples, arith,
         kernel, 2.0 * samples * steps / t / 1e9);
  vs_ctx_destroy(ctx);
  free(lanes);
  return 0;
}
