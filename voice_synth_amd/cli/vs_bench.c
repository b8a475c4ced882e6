/*
 * vs_bench -- the throughput of the fused source->filter path from plain C, no Python:
 *
 *     vs_bench [--lanes N] [--steps K] [--warmup W] [--arith exact|fma|f32] [--host] [--fresh] [--gpus G [--rccl] [--no-verify]]
 *
 * Workload: BASELINE.json configs[2] -- N utterances (default 65536), vowel table "12467"[lane % 5],
 * 16 kHz, 1 s, jitter 1 %, shimmer 0.5 dB (-s 5.76), glottal noise 20 dB, lane key = 1 + lane --
 * built from the reference's own command lines through vs_flowgen_parse()/vs_vowel_parse(), i.e.
 * exactly what voice_synth_amd/configs.py describes for bench.py.  Timed: K launches of one plan
 * into a device buffer (host clock around launch ... vs_plan_status, which waits), after W warm-up
 * launches.  --host times vs_synth() into a pinned host buffer instead (PCIe included).
 * --gpus G times vs_node_synth_gather(): N utterances PER DEVICE (weak scaling, like bench.py),
 * contiguous blocks over devices 0..G-1 (or the ordinals in VS_DEVICES, e.g. "0,0" = two logical
 * shards of one device), every finished chunk copied into device 0's memory behind the synthesis
 * of the next one; the line then also carries the slowest shard's compute time and how every
 * shard reaches the root ("links").  --rccl: the chunks travel by ncclSend / ncclRecv on a communicator
 * the node object owns (vs_node_set_transport) instead of peer DMA -- host code in C, RCCL gather.
 * After the timed steps the gathered PCM is compared, row by row, with what device 0 gives when it synthesises the same
 * lanes alone ("gathered_equals_one_device"; --no-verify skips it), and the line lists the PCI bus id of the device
 * behind every shard ("devices", "distinct_devices").
 * --fresh: what a caller pays who never launches the same plan twice -- K batches (default 50) of N utterances nobody has
 * synthesised yet (new seeds; the descriptions exist beforehand), a plan per batch: a second host thread makes the plan of
 * batch k + 1 (and takes the plan of batch k - 2 down) while kernel k runs, this thread only launches.  Timed on the
 * device, from in front of the first launch to behind the last (vs_ctx_timer_*); the reference re-seeds every run
 * (flowgen_shimmer.c:241), so fresh draws per batch IS its behaviour.  The line also says what a plan cost the planning
 * thread and how long the launching thread waited for one.
 * One line of JSON on stdout.  bench.py remains the driver's benchmark; this is the same
 * measurement for a maintainer who only has the C side.
 */
#include <pthread.h>
#include <semaphore.h>
#include <time.h>

#include "cli_common.h"

static double now_s(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* --fresh: the planning thread */
#define FRESH_AHEAD 2 /* plans that may exist beyond the one being launched */
typedef struct Fresh {
  vs_ctx *ctx;
  vs_lane **descr;    /* the caller's descriptions, one array per batch, made before the clock starts */
  size_t n_lanes, n_samples;
  int batches;
  vs_plan **plans;
  sem_t ready, room, launched;
  int rc;
  double host_ms_sum, upload_ms_sum, create_wall_ms_sum, destroy_ms_sum, create_wall_ms_max;
} Fresh;

static double now_s(void);
static void *fresh_planner(void *arg)
{
  Fresh *f = (Fresh *)arg;
  for (int k = 0; k < f->batches; k++) {
    sem_wait(&f->room);
    /* the plan of two batches ago has been launched AND the batch behind it too: its kernel is over or about to be --
     * taking it down here (hipFree waits for the device) costs this thread time, not the launching one */
    if (k >= FRESH_AHEAD + 1) {
      sem_wait(&f->launched);
      const double t0 = now_s();
      vs_plan_destroy(f->plans[k - FRESH_AHEAD - 1]);
      f->plans[k - FRESH_AHEAD - 1] = NULL;
      f->destroy_ms_sum += (now_s() - t0) * 1e3;
    }
    const vs_lane *lanes = f->descr[k];
    const double t0 = now_s();
    vs_plan *p = NULL;
    const int rc = vs_plan_create(f->ctx, lanes, f->n_lanes, f->n_samples, &p);
    const double wall = (now_s() - t0) * 1e3;
    if (rc != VS_OK) {
      f->rc = rc;
      f->plans[k] = NULL;
      sem_post(&f->ready);
      return NULL;
    }
    double h = 0.0, u = 0.0;
    vs_plan_timing(p, &h, &u);
    f->host_ms_sum += h;
    f->upload_ms_sum += u;
    f->create_wall_ms_sum += wall;
    if (wall > f->create_wall_ms_max) f->create_wall_ms_max = wall;
    f->plans[k] = p;
    sem_post(&f->ready);
  }
  return NULL;
}

int main(int argc, char **argv)
{
  size_t n_lanes = 65536;
  int steps = 20, warmup = 5, host = 0, gpus = 0, use_rccl = 0, verify = 1, fresh = 0, steps_given = 0;
  const char *arith = "exact";
  for (int i = 1; i < argc; i++) {
    if (!strcmp(argv[i], "--lanes") && i + 1 < argc) n_lanes = (size_t)strtoull(argv[++i], NULL, 0);
    else if (!strcmp(argv[i], "--steps") && i + 1 < argc) steps = atoi(argv[++i]), steps_given = 1;
    else if (!strcmp(argv[i], "--fresh")) fresh = 1;
    else if (!strcmp(argv[i], "--warmup") && i + 1 < argc) warmup = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--arith") && i + 1 < argc) arith = argv[++i];
    else if (!strcmp(argv[i], "--host")) host = 1;
    else if (!strcmp(argv[i], "--gpus") && i + 1 < argc) gpus = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--rccl")) use_rccl = 1;
    else if (!strcmp(argv[i], "--no-verify")) verify = 0;
    else {
      fprintf(stderr, "usage: vs_bench [--lanes N] [--steps K] [--warmup W] [--arith exact|fma|f32] [--host] [--fresh] [--gpus G [--rccl] [--no-verify]]\n");
      return 1;
    }
  }
  if (fresh && !steps_given) steps = 50;
  const int arith_id = !strcmp(arith, "fma") ? VS_ARITH_FMA : (!strcmp(arith, "f32") ? VS_ARITH_F32 : VS_ARITH_EXACT);
  if (n_lanes == 0 || steps < 1 || warmup < 0 || gpus < 0 || gpus > 64) return 1;
  const size_t per_gpu = n_lanes;
  if (gpus > 0) n_lanes *= (size_t)gpus;

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

  if (gpus > 0) {
    int devs[64];
    for (int d = 0; d < gpus; d++) devs[d] = d;
    const char *list = getenv("VS_DEVICES");
    for (int d = 0; list && *list && d < gpus; d++) {
      devs[d] = atoi(list);
      const char *c = strchr(list, ',');
      list = c ? c + 1 : "";
    }
    vs_node *node = NULL;
    int rc = vs_node_create(devs, gpus, &node);
    vs_ctx *root = NULL;
    void *out = NULL;
    double total_ms = 0.0, shard_ms = 0.0, sum_ms = 0.0, worst_shard = 0.0;
    if (rc == VS_OK && arith_id != VS_ARITH_EXACT) rc = vs_node_set_arith(node, arith_id);
    if (rc == VS_OK && use_rccl) {
      rc = vs_node_set_transport(node, VS_NODE_TRANSPORT_RCCL);
      if (rc != VS_OK) fprintf(stderr, "vs_bench: RCCL transport: %s (ncclResult %d)\n", vs_strerror(rc), vs_node_last_rccl_error(node));
    }
    char links[64 * 8 + 4] = "";
    for (int d = 0; rc == VS_OK && d < gpus; d++) {
      static const char *const names[] = {"self", "peer", "staged", "rccl"};
      const int l = vs_node_link(node, d);
      snprintf(links + strlen(links), sizeof(links) - strlen(links), "%s\"%s\"", d ? ", " : "", (l >= 0 && l < 4) ? names[l] : "?");
    }
    /* what RCCL itself says every shard's communicator spans (ncclCommCount): `gpus` on a healthy node */
    char rccl_ranks[64 * 6 + 4] = "";
    for (int d = 0; rc == VS_OK && d < gpus; d++)
      snprintf(rccl_ranks + strlen(rccl_ranks), sizeof(rccl_ranks) - strlen(rccl_ranks), "%s%d", d ? ", " : "", vs_node_rccl_ranks(node, d));
    /* which device serves each shard: a node run must show that `gpus` DIFFERENT devices took part */
    char pcis[64 * 24 + 4] = "";
    char seen[64][32];
    int distinct = 0;
    for (int d = 0; rc == VS_OK && d < gpus; d++) {
      vs_ctx *c = NULL;
      char id[32] = "?";
      rc = vs_node_ctx(node, d, &c);
      if (rc == VS_OK) (void)vs_ctx_device_pci(c, id, sizeof(id));
      snprintf(pcis + strlen(pcis), sizeof(pcis) - strlen(pcis), "%s\"%s\"", d ? ", " : "", id);
      int known = 0;
      for (int e = 0; e < distinct; e++) known = known || !strcmp(seen[e], id);
      if (!known) snprintf(seen[distinct++], sizeof(seen[0]), "%s", id);
    }
    if (rc == VS_OK) rc = vs_node_ctx(node, 0, &root);
    if (rc == VS_OK) rc = vs_dev_alloc(root, n_lanes * n_samples * sizeof(int16_t), &out);
    for (int k = 0; rc == VS_OK && k < warmup + steps; k++) {
      rc = vs_node_synth_gather(node, lanes, n_lanes, n_samples, (int16_t *)out, n_samples, VS_NODE_OVERLAP, &total_ms, &shard_ms);
      if (k >= warmup) {
        sum_ms += total_ms;
        if (shard_ms > worst_shard) worst_shard = shard_ms;
      }
    }
    /* the gathered PCM against ONE device synthesising the same lanes alone, chunk by chunk (a lane's draws are
     * keyed by the seed in its record, so the cut over the devices must not show: flowgen_shimmer.c:121-122,
     * vowel_new.c:90 -- nothing but per-utterance state exists) */
    size_t rows_verified = 0, rows_differ = 0;
    if (rc == VS_OK && verify) {
      const size_t chunk = 16384 < n_lanes ? 16384 : n_lanes;
      int16_t *a = (int16_t *)malloc(chunk * n_samples * sizeof(int16_t));
      int16_t *b = (int16_t *)malloc(chunk * n_samples * sizeof(int16_t));
      void *scratch = NULL;
      if (!a || !b) rc = VS_ERR_NOMEM;
      if (rc == VS_OK) rc = vs_dev_alloc(root, chunk * n_samples * sizeof(int16_t), &scratch);
      for (size_t r0 = 0; rc == VS_OK && r0 < n_lanes; r0 += chunk) {
        const size_t rows = (n_lanes - r0 < chunk) ? (n_lanes - r0) : chunk;
        vs_plan *plan = NULL;
        rc = vs_plan_create(root, lanes + r0, rows, n_samples, &plan);
        if (rc == VS_OK) rc = vs_plan_launch(plan, VS_KIND_SYNTH, NULL, 0, (int16_t *)scratch, n_samples, NULL, 0, NULL);
        if (rc == VS_OK) rc = vs_plan_status(plan, NULL);
        if (rc == VS_OK) rc = vs_dev_download(root, a, scratch, rows * n_samples * sizeof(int16_t));
        if (rc == VS_OK) rc = vs_dev_download(root, b, (char *)out + r0 * n_samples * sizeof(int16_t), rows * n_samples * sizeof(int16_t));
        for (size_t r = 0; rc == VS_OK && r < rows; r++)
          rows_differ += memcmp(a + r * n_samples, b + r * n_samples, n_samples * sizeof(int16_t)) != 0;
        if (rc == VS_OK) rows_verified += rows;
        if (plan) vs_plan_destroy(plan);
      }
      if (scratch) vs_dev_free(root, scratch);
      free(a);
      free(b);
      if (rc == VS_OK && rows_differ) {
        fprintf(stderr, "vs_bench: %zu of %zu gathered rows differ from what one device gives\n", rows_differ, rows_verified);
        rc = VS_ERR_INTERNAL;
      }
    }
    if (rc != VS_OK) fprintf(stderr, "vs_bench: %s\n", vs_strerror(rc));
    else
      printf("{\"metric\": \"synthesised Msamples/s (whole node), PCM gathered into device %d\", \"value\": %.1f, "
             "\"unit\": \"Msamples/s\", \"n_gpus\": %d, \"ms_per_step\": %.4f, \"slowest_shard_compute_ms\": %.4f, "
             "\"steps\": %d, \"warmup\": %d, \"utterances_per_gpu\": %zu, \"samples_per_utterance\": %llu, "
             "\"arith\": \"%s\", \"links\": [%s], \"rccl_comm_ranks\": [%s], \"devices\": [%s], \"distinct_devices\": %d, \"rows_verified_against_one_device\": %zu, "
             "\"gathered_equals_one_device\": %s, "
             "\"path\": \"vs_node_synth_gather, copies behind the synthesis (plans of every chunk included)\"}\n",
             devs[0], (double)n_lanes * (double)n_samples * steps / (sum_ms * 1e-3) / 1e6, gpus, sum_ms / steps, worst_shard,
             steps, warmup, per_gpu, (unsigned long long)n_samples, arith, links, rccl_ranks, pcis, distinct, rows_verified,
             verify ? "true" : "null");
    if (out) vs_dev_free(root, out);
    vs_node_destroy(node);
    free(lanes);
    return rc == VS_OK ? 0 : 1;
  }

  vs_ctx *ctx = NULL;
  if (vs_cli_open_ctx(&ctx) != VS_OK) return 1;
  vs_ctx_set_arith(ctx, arith_id);
  int rc = VS_OK;
  double t = 0.0;
  char kernel[160] = "";
  if (fresh) {
    const size_t pitch = vs_row_pitch((size_t)n_samples);
    void *out = NULL;
    Fresh f;
    memset(&f, 0, sizeof(f));
    f.ctx = ctx;
    f.n_lanes = n_lanes;
    f.n_samples = (size_t)n_samples;
    f.batches = steps;
    f.plans = (vs_plan **)calloc((size_t)steps, sizeof(vs_plan *));
    /* the batches' descriptions are the caller's INPUT: utterances nobody has synthesised yet (new seeds), all of them
     * described before the clock starts (27 MB per batch of 65536) */
    f.descr = (vs_lane **)calloc((size_t)steps, sizeof(vs_lane *));
    if (!f.plans || !f.descr) rc = VS_ERR_NOMEM;
    for (int k = 0; rc == VS_OK && k < steps; k++) {
      f.descr[k] = (vs_lane *)malloc(n_lanes * sizeof(vs_lane));
      if (!f.descr[k]) {
        rc = VS_ERR_NOMEM;
        break;
      }
      memcpy(f.descr[k], lanes, n_lanes * sizeof(vs_lane));
      for (size_t l = 0; l < n_lanes; l++) {
        f.descr[k][l].seed = 1 + (uint64_t)l + (uint64_t)(k + 1) * (uint64_t)n_lanes;
        f.descr[k][l].out_seed = f.descr[k][l].seed;
      }
    }
    if (rc == VS_OK) rc = vs_dev_alloc(ctx, n_lanes * pitch * sizeof(int16_t), &out);
    /* warm-up: batches made and launched one after the other (the first plan of a process, the clock of the chip) */
    for (int k = 0; rc == VS_OK && k < (warmup > 0 ? warmup : 1); k++) {
      vs_plan *p = NULL;
      rc = vs_plan_create(ctx, lanes, n_lanes, n_samples, &p);
      if (rc == VS_OK) rc = vs_plan_launch(p, VS_KIND_SYNTH, NULL, 0, (int16_t *)out, pitch, NULL, 0, NULL);
      if (rc == VS_OK) rc = vs_plan_status(p, NULL);
      if (p && k == 0) vs_plan_kernel_name(p, VS_KIND_SYNTH, kernel, sizeof(kernel));
      if (p) vs_plan_destroy(p);
    }
    double wait_ms = 0.0, wait_max = 0.0, dev_ms = 0.0;
    size_t rows_differ = 0;
    if (rc == VS_OK) {
      sem_init(&f.ready, 0, 0);
      sem_init(&f.room, 0, FRESH_AHEAD + 1);
      sem_init(&f.launched, 0, 0);
      pthread_t th;
      if (pthread_create(&th, NULL, fresh_planner, &f) != 0) rc = VS_ERR_INTERNAL;
      const double t0 = now_s();
      int last = -1;
      for (int k = 0; rc == VS_OK && k < steps; k++) {
        const double w0 = now_s();
        sem_wait(&f.ready); /* plan k exists (normally it has for a while: the planner is ahead) */
        const double w = (now_s() - w0) * 1e3;
        if (k > 0) { /* (the first plan is made with nothing to hide behind) */
          wait_ms += w;
          if (w > wait_max) wait_max = w;
        }
        if (!f.plans[k]) {
          rc = f.rc != VS_OK ? f.rc : VS_ERR_INTERNAL;
          break;
        }
        if (k == 0) rc = vs_ctx_timer_mark(ctx, 0); /* in front of the first launch */
        if (rc == VS_OK) rc = vs_plan_launch(f.plans[k], VS_KIND_SYNTH, NULL, 0, (int16_t *)out, pitch, NULL, 0, NULL);
        last = k;
        if (k >= 1) sem_post(&f.launched); /* plan k - 1 has a launch behind it: two batches on, it may be taken down */
        sem_post(&f.room);
      }
      if (rc == VS_OK) rc = vs_ctx_timer_mark(ctx, 1); /* behind the last */
      if (rc == VS_OK) rc = vs_ctx_timer_elapsed(ctx, &dev_ms);
      t = now_s() - t0;
      if (rc != VS_OK) { /* let the planner run out */
        for (int k = 0; k < steps + FRESH_AHEAD + 2; k++) {
          sem_post(&f.room);
          sem_post(&f.launched);
        }
      }
      pthread_join(th, NULL);
      /* what the last batch left in the buffer against one plain launch of the same utterances into a second buffer: plans
       * made next to running kernels, from blocks other plans have just given back, must synthesise the same samples */
      if (rc == VS_OK && verify) {
        void *out2 = NULL;
        vs_plan *p2 = NULL;
        const size_t bytes = n_lanes * pitch * sizeof(int16_t);
        int16_t *a = (int16_t *)malloc(bytes), *b = (int16_t *)malloc(bytes);
        if (!a || !b) rc = VS_ERR_NOMEM;
        if (rc == VS_OK) rc = vs_ctx_synchronize(ctx);
        if (rc == VS_OK) rc = vs_dev_alloc(ctx, bytes, &out2);
        if (rc == VS_OK) rc = vs_plan_create(ctx, f.descr[steps - 1], n_lanes, n_samples, &p2);
        if (rc == VS_OK) rc = vs_plan_launch(p2, VS_KIND_SYNTH, NULL, 0, (int16_t *)out2, pitch, NULL, 0, NULL);
        if (rc == VS_OK) rc = vs_plan_status(p2, NULL);
        if (rc == VS_OK) rc = vs_dev_download(ctx, a, out, bytes);
        if (rc == VS_OK) rc = vs_dev_download(ctx, b, out2, bytes);
        for (size_t r = 0; rc == VS_OK && r < n_lanes; r++)
          rows_differ += memcmp(a + r * pitch, b + r * pitch, n_samples * sizeof(int16_t)) != 0;
        if (rc == VS_OK && rows_differ) {
          fprintf(stderr, "vs_bench: %zu rows of the last batch differ from a plain launch of the same utterances\n", rows_differ);
          rc = VS_ERR_INTERNAL;
        }
        if (p2) vs_plan_destroy(p2);
        if (out2) vs_dev_free(ctx, out2);
        free(a);
        free(b);
      }
      for (int k = 0; k <= last && rc == VS_OK; k++)
        if (f.plans[k] && k >= steps - FRESH_AHEAD - 2) rc = vs_plan_status(f.plans[k], NULL); /* the health word of the ones still here */
      for (int k = 0; k < steps; k++)
        if (f.plans[k]) vs_plan_destroy(f.plans[k]);
      sem_destroy(&f.ready);
      sem_destroy(&f.room);
      sem_destroy(&f.launched);
    }
    if (out) vs_dev_free(ctx, out);
    for (int k = 0; f.descr && k < steps; k++) free(f.descr[k]);
    free(f.descr);
    free(f.plans);
    if (rc != VS_OK) {
      fprintf(stderr, "vs_bench: %s\n", vs_strerror(rc));
      vs_ctx_destroy(ctx);
      free(lanes);
      return 1;
    }
    const double samples = (double)n_lanes * (double)n_samples;
    printf("{\"metric\": \"synthesised Msamples/s, a plan per batch of new utterances\", \"value\": %.1f, \"unit\": \"Msamples/s\", "
           "\"ms_per_batch\": %.4f, \"ms_per_batch_host_clock\": %.4f, \"batches\": %d, \"utterances_per_batch\": %zu, "
           "\"samples_per_utterance\": %llu, \"arith\": \"%s\", \"kernel\": \"%s\", \"row_pitch_samples\": %zu, "
           "\"plan_host_ms_avg\": %.3f, \"plan_upload_ms_avg\": %.3f, \"plan_create_wall_ms_avg\": %.3f, \"plan_create_wall_ms_max\": %.3f, "
           "\"plan_destroy_ms_avg\": %.3f, \"launcher_waited_ms_avg\": %.4f, \"launcher_waited_ms_max\": %.4f, "
           "\"last_batch_equals_a_plain_launch\": %s, "
           "\"how\": \"vs_plan_create on a second host thread while the kernel of the batch before runs; device time from in front of the "
           "first launch to behind the last (vs_ctx_timer_*)\"}\n",
           samples * steps / (dev_ms * 1e-3) / 1e6, dev_ms / steps, t / steps * 1e3, steps, n_lanes, (unsigned long long)n_samples, arith,
           kernel, pitch, f.host_ms_sum / steps, f.upload_ms_sum / steps, f.create_wall_ms_sum / steps, f.create_wall_ms_max,
           steps > FRESH_AHEAD + 1 ? f.destroy_ms_sum / (steps - FRESH_AHEAD - 1) : 0.0, steps > 1 ? wait_ms / (steps - 1) : 0.0, wait_max,
           verify ? "true" : "null");
    vs_ctx_destroy(ctx);
    free(lanes);
    return 0;
  }
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
    const size_t pitch = vs_row_pitch((size_t)n_samples); /* the buffer is ours: the pitch the kernels' stores like */
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
         "\"path\": \"%s\", \"row_pitch_samples\": %zu, \"GB_per_s_of_pcm\": %.1f}\n",
         samples * steps / t / 1e6, t / steps * 1e3, steps, warmup, n_lanes, (unsigned long long)n_samples, arith,
         kernel, host ? (size_t)n_samples : vs_row_pitch((size_t)n_samples), 2.0 * samples * steps / t / 1e9);
  vs_ctx_destroy(ctx);
  free(lanes);
  return 0;
}
