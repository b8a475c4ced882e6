/*
 * vs_api.c -- host side of the C ABI (include/voice_synth.h): device context, plans
 * (parameter expansion + cos tables + upload) and launches.  C11 + pthreads, the HIP runtime through
 * its C API; the kernels are launched through the extern "C" launchers of vs_kernels.hip.
 *
 * The device-free half of plan creation (lane expansion with the reference's operand types, cos rows, ring
 * policy, the order of the lanes) is csrc/vs_planhost.c.
 * This translation unit is compiled with -ffp-contract=off.
 */
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "vs_internal.h"

double vs_now_ms(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return 1e3 * (double)ts.tv_sec + 1e-6 * (double)ts.tv_nsec;
}

int vs_ctx_create(int device, vs_ctx **out)
{
  if (!out) return VS_ERR_ARG;
  *out = NULL;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return VS_ERR_NODEVICE;
  if (device < 0 || device >= count) return VS_ERR_NODEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return VS_ERR_NODEVICE;
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return VS_ERR_NODEVICE; /* code object is gfx950 only */
  if (hipSetDevice(device) != hipSuccess) return VS_ERR_NODEVICE;
  vs_ctx *ctx = (vs_ctx *)calloc(1, sizeof(vs_ctx));
  if (!ctx) return VS_ERR_NOMEM;
  ctx->device = device;
  ctx->arith = VS_ARITH_EXACT;
  ctx->stream = NULL;
  ctx->upload = NULL;
  ctx->last_hip_error = 0;
  snprintf(ctx->name, sizeof(ctx->name), "%.80s (%.40s)", prop.name, prop.gcnArchName);
  ctx->cu_count = prop.multiProcessorCount;
  memset(&ctx->tuning, 0, sizeof(ctx->tuning));
  memset(&ctx->pool, 0, sizeof(ctx->pool));
  /* Experiments only (tools/gpu_sweep.sh): with VS_DEBUG_TUNING=1 in the environment the knobs
   * are read ONCE, here, and go through the same validation as vs_ctx_set_tuning().  Without it
   * no environment variable can change what the library launches. */
  {
    const char *dbg = getenv("VS_DEBUG_TUNING");
    if (dbg && strcmp(dbg, "1") == 0) {
      vs_tuning t;
      memset(&t, 0, sizeof(t));
      const char *v;
      if ((v = getenv("VS_KERNEL")) != NULL) t.kernel = strcmp(v, "ws") == 0 ? VS_KERNEL_WS : (strcmp(v, "single") == 0 ? VS_KERNEL_SINGLE : VS_KERNEL_AUTO);
      if ((v = getenv("VS_RING_SLOTS")) != NULL) t.ring_slots = atoi(v);
      if ((v = getenv("VS_READY_MIN")) != NULL) t.ready_min = atoi(v);
      if ((v = getenv("VS_WS_PAIRS")) != NULL) t.ws_pairs = atoi(v);
      if ((v = getenv("VS_WS_ROLES")) != NULL) t.ws_roles = atoi(v);
      if ((v = getenv("VS_GEN_LOW")) != NULL) t.gen_low = atoi(v);
      if ((v = getenv("VS_GEN_MIN")) != NULL) t.gen_min = atoi(v);
      if ((v = getenv("VS_MIXED_RINGS")) != NULL) t.mixed_rings = (atoi(v) == 0) ? -1 : (atoi(v) == 1 ? 0 : atoi(v));
      if ((v = getenv("VS_WS_PRIO")) != NULL) t.ws_filter_prio = (atoi(v) == 0) ? -1 : atoi(v);
      if (vs_ctx_set_tuning(ctx, &t) != VS_OK) {
        free(ctx);
        return VS_ERR_ARG;
      }
    }
  }
  *out = ctx;
  return VS_OK;
}

/* The HIP runtime sets up its copy path (staging buffers, the transfer queue of this device) at a process's first
 * host-to-device copy: 27 ms on a bare process whatever the size (tools/hip_startup_probe.c), ~100 ms in one that
 * has loaded PyTorch (profiles/r04_plan_cost.txt).  Paid HERE, once per context, the first time a plan that copies
 * is made and before that plan's clock starts -- so that a first vs_plan_create costs what every later one costs --
 * and NOT by vs_ctx_create: the drop-in programs make zero-copy plans only (VS_PLAN_ZERO_COPY) and never copy. */
int vs_copy_path_warm(vs_ctx *ctx)
{
  if (ctx->copy_warm) return VS_OK;
  const double t0 = vs_now_ms();
  void *scratch = NULL;
  const size_t warm_bytes = 1u << 20;
  void *host = malloc(warm_bytes);
  hipError_t e = host ? hipSetDevice(ctx->device) : hipErrorOutOfMemory;
  /* on the stream the records of later plans go up on (creating a stream costs milliseconds the first time) -- a
   * NON-BLOCKING stream of the context's own, never the legacy stream: that one waits for every blocking stream of the
   * process, the caller's running kernel included, and vs_plan_create promises to run next to a launch */
  if (e == hipSuccess && !ctx->own_upload) e = hipStreamCreateWithFlags(&ctx->own_upload, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipMalloc(&scratch, warm_bytes);
  if (e == hipSuccess) {
    memset(host, 0, warm_bytes);
    e = hipMemcpyAsync(scratch, host, warm_bytes, hipMemcpyHostToDevice, ctx->own_upload);
    if (e == hipSuccess) e = hipMemcpyAsync(host, scratch, sizeof(int), hipMemcpyDeviceToHost, ctx->own_upload);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->own_upload);
    (void)hipFree(scratch);
  }
  free(host);
  if (e != hipSuccess) {
    ctx->last_hip_error = (int)e;
    return (e == hipErrorOutOfMemory) ? VS_ERR_NOMEM : VS_ERR_HIP;
  }
  ctx->copy_warm = 1;
  ctx->copy_warm_ms = vs_now_ms() - t0;
  return VS_OK;
}

int vs_ctx_set_tuning(vs_ctx *ctx, const vs_tuning *t)
{
  if (!ctx) return VS_ERR_ARG;
  if (!t) {
    memset(&ctx->tuning, 0, sizeof(ctx->tuning));
    return VS_OK;
  }
  if (t->kernel != VS_KERNEL_AUTO && t->kernel != VS_KERNEL_SINGLE && t->kernel != VS_KERNEL_WS) return VS_ERR_ARG;
  if (t->ring_slots < 0 || t->ring_slots > 65536) return VS_ERR_ARG;
  if (t->ready_min < 0 || t->ready_min > 64) return VS_ERR_ARG;
  if (t->ws_pairs < 0 || t->ws_pairs == 3 || t->ws_pairs > 4) return VS_ERR_ARG;
  /* a lane short of gen_low samples starts a round at once: below one super-step the filter wave
   * could starve while the generator waits for company */
  if (t->gen_low != 0 && t->gen_low < VS_SS) return VS_ERR_ARG;
  if (t->gen_min < 0 || t->gen_min > 64) return VS_ERR_ARG;
  if (t->spin_limit < 0) return VS_ERR_ARG;
  if (t->fault < 0 || t->fault > VS_FAULT_REROUND) return VS_ERR_ARG;
  if (t->ws_filter_prio < -1 || t->ws_filter_prio > 3) return VS_ERR_ARG;
  if (t->ws_roles != 0 && t->ws_roles != 2 && t->ws_roles != 3) return VS_ERR_ARG;
  if (t->mixed_rings < -1 || t->mixed_rings > 4096) return VS_ERR_ARG;
  ctx->tuning = *t;
  return VS_OK;
}

void vs_ctx_destroy(vs_ctx *ctx)
{
  if (!ctx) return;
  vs_pool_release(ctx);
  if (ctx->own_upload) {
    (void)hipSetDevice(ctx->device);
    (void)hipStreamDestroy(ctx->own_upload);
  }
  vs_plan_cache_release(ctx);
  for (int k = 0; k < 2; k++)
    if (ctx->timer[k]) (void)hipEventDestroy(ctx->timer[k]);
  vs_planws_destroy(ctx->planws);
  if (ctx->plan_pin) (void)hipHostFree(ctx->plan_pin);
  free(ctx);
}

int vs_ctx_set_stream(vs_ctx *ctx, void *hip_stream)
{
  if (!ctx) return VS_ERR_ARG;
  ctx->stream = (hipStream_t)hip_stream;
  return VS_OK;
}

int vs_ctx_set_arith(vs_ctx *ctx, int arith)
{
  if (!ctx || (arith != VS_ARITH_EXACT && arith != VS_ARITH_FMA && arith != VS_ARITH_F32)) return VS_ERR_ARG;
  ctx->arith = arith;
  return VS_OK;
}

int vs_ctx_last_hip_error(const vs_ctx *ctx) { return ctx ? ctx->last_hip_error : 0; }

int vs_ctx_device_info(const vs_ctx *ctx, char *name, size_t name_len, int *cu_count)
{
  if (!ctx) return VS_ERR_ARG;
  if (name && name_len) snprintf(name, name_len, "%s", ctx->name);
  if (cu_count) *cu_count = ctx->cu_count;
  return VS_OK;
}

int vs_ctx_device_pci(const vs_ctx *ctx, char *bus_id, size_t len)
{
  if (!ctx || !bus_id || len < 16) return VS_ERR_ARG;
  bus_id[0] = '\0';
  const hipError_t e = hipDeviceGetPCIBusId(bus_id, (int)len, ctx->device);
  if (e != hipSuccess) return VS_ERR_HIP;
  return VS_OK;
}

int vs_ctx_synchronize(vs_ctx *ctx)
{
  if (!ctx) return VS_ERR_ARG;
  VS_HIP(ctx, hipSetDevice(ctx->device));
  VS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return VS_OK;
}

int vs_ctx_timer_mark(vs_ctx *ctx, int which)
{
  if (!ctx || (which != 0 && which != 1)) return VS_ERR_ARG;
  VS_HIP(ctx, hipSetDevice(ctx->device));
  if (!ctx->timer[which]) VS_HIP(ctx, hipEventCreate(&ctx->timer[which]));
  VS_HIP(ctx, hipEventRecord(ctx->timer[which], ctx->stream));
  return VS_OK;
}

int vs_ctx_timer_elapsed(vs_ctx *ctx, double *ms)
{
  if (!ctx || !ms || !ctx->timer[0] || !ctx->timer[1]) return VS_ERR_ARG;
  float f = 0.0f;
  VS_HIP(ctx, hipSetDevice(ctx->device));
  VS_HIP(ctx, hipEventSynchronize(ctx->timer[1]));
  VS_HIP(ctx, hipEventElapsedTime(&f, ctx->timer[0], ctx->timer[1]));
  *ms = (double)f;
  return VS_OK;
}

/* One probe launch per workgroup size: a workgroup per CU (most of the LDS each, like the fused launches), every
 * wavefront reports its HW_ID; "cyclic" = in every workgroup the first four wavefronts run on four different SIMDs
 * and wavefront w runs where wavefront w % 4 does. */
static int simd_probe(vs_ctx *ctx)
{
  if (ctx->simd_probed) return ctx->simd_probed > 0 ? VS_OK : VS_ERR_HIP;
  const unsigned grid = (unsigned)(ctx->cu_count > 0 ? ctx->cu_count : 256);
  unsigned *d = NULL, *h = (unsigned *)calloc((size_t)grid * 16, sizeof(unsigned));
  ctx->simd_probed = -1;
  ctx->simd_cyclic12 = ctx->simd_cyclic8 = 0;
  ctx->simd_odd_wgs = 0;
  if (!h) return VS_ERR_NOMEM;
  hipError_t e = hipSetDevice(ctx->device);
  /* on the context's own non-blocking stream, not the caller's launch stream: the first plan that wants a three-role
   * layout may be made while a kernel of the caller's runs there (the probe still needs the CUs: it starts when they are
   * free, but it does not make vs_plan_create wait for work that was queued behind that kernel) */
  if (e == hipSuccess && !ctx->own_upload) e = hipStreamCreateWithFlags(&ctx->own_upload, hipStreamNonBlocking);
  hipStream_t ps = ctx->own_upload;
  if (e == hipSuccess) e = hipMalloc((void **)&d, (size_t)grid * 16 * sizeof(unsigned));
  int ok[2] = {1, 1};
  unsigned odd = 0;
  for (int pass = 0; pass < 2 && e == hipSuccess; pass++) {
    const int waves = pass == 0 ? 12 : 8;
    memset(h, 0, (size_t)grid * 16 * sizeof(unsigned));
    e = hipMemcpyAsync(d, h, (size_t)grid * 16 * sizeof(unsigned), hipMemcpyHostToDevice, ps);
    if (e == hipSuccess) e = vs_launch_simd_probe(waves, grid, (size_t)VS_LDS_LIMIT - 8192, d, ps);
    if (e == hipSuccess) e = hipMemcpyAsync(h, d, (size_t)grid * 16 * sizeof(unsigned), hipMemcpyDeviceToHost, ps);
    if (e == hipSuccess) e = hipStreamSynchronize(ps);
    if (e != hipSuccess) break;
    for (unsigned g = 0; g < grid; g++) {
      /* "dealt four at a time": the first four wavefronts land on four DIFFERENT SIMDs (in whatever order -- MI355X
       * shows four rotations of (0, 2, 1, 3)) and wavefront w joins wavefront w % 4, on the same CU */
      int good = 1;
      unsigned seen = 0;
      for (int w = 0; w < waves; w++) {
        const unsigned hw = h[(size_t)g * 16 + (size_t)w], first = h[(size_t)g * 16 + (size_t)(w & 3)];
        if (!(hw & 0x80000000u)) good = 0;
        if (w < 4) seen |= 1u << ((hw >> 4) & 3u);
        else if (((hw >> 4) & 3u) != ((first >> 4) & 3u) || ((hw >> 8) & 0xFFu) != ((first >> 8) & 0xFFu)) good = 0;
      }
      if (seen != 0xFu) good = 0;
      if (!good) {
        ok[pass] = 0;
        odd++;
      }
    }
  }
  if (d) (void)hipFree(d);
  free(h);
  if (e != hipSuccess) {
    ctx->last_hip_error = (int)e;
    return VS_ERR_HIP;
  }
  ctx->simd_cyclic12 = ok[0];
  ctx->simd_cyclic8 = ok[1];
  ctx->simd_odd_wgs = odd;
  ctx->simd_probed = 1;
  return VS_OK;
}

int vs_ctx_simd_dealing(vs_ctx *ctx, int *cyclic12, int *cyclic8)
{
  if (!ctx) return VS_ERR_ARG;
  const int rc = simd_probe(ctx);
  if (cyclic12) *cyclic12 = ctx->simd_cyclic12;
  if (cyclic8) *cyclic8 = ctx->simd_cyclic8;
  return rc;
}

int vs_ctx_selftest(vs_ctx *ctx, uint64_t *failures)
{
  if (!ctx) return VS_ERR_ARG;
  unsigned long long *d = NULL, h[VS_SELFTEST_COUNTERS] = {0};
  VS_HIP(ctx, hipSetDevice(ctx->device));
  VS_HIP(ctx, hipMalloc((void **)&d, sizeof(h)));
  hipError_t e = hipMemsetAsync(d, 0, sizeof(h), ctx->stream);
  if (e == hipSuccess) e = vs_launch_selftest(d, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  (void)hipFree(d);
  if (e != hipSuccess) {
    ctx->last_hip_error = (int)e;
    return VS_ERR_HIP;
  }
  /* [6]: workgroups of the wave-to-SIMD probe that were not dealt four at a time (all of them if the probe failed) */
  h[6] = (simd_probe(ctx) == VS_OK) ? ctx->simd_odd_wgs : 1ull;
  unsigned long long any = 0;
  for (int k = 0; k < VS_SELFTEST_COUNTERS; k++) {
    if (failures) failures[k] = h[k];
    any |= h[k];
  }
  return any ? VS_ERR_INTERNAL : VS_OK;
}

int vs_dev_alloc(vs_ctx *ctx, size_t bytes, void **ptr)
{
  if (!ctx || !ptr) return VS_ERR_ARG;
  VS_HIP(ctx, hipSetDevice(ctx->device));
  VS_HIP(ctx, hipMalloc(ptr, bytes ? bytes : 1));
  return VS_OK;
}
int vs_dev_free(vs_ctx *ctx, void *ptr)
{
  if (!ctx) return VS_ERR_ARG;
  VS_HIP(ctx, hipSetDevice(ctx->device));
  VS_HIP(ctx, hipFree(ptr));
  return VS_OK;
}
int vs_dev_upload(vs_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes)
{
  if (!ctx) return VS_ERR_ARG;
  VS_HIP(ctx, hipSetDevice(ctx->device));
  VS_HIP(ctx, hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
  VS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return VS_OK;
}
int vs_dev_download(vs_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes)
{
  if (!ctx) return VS_ERR_ARG;
  VS_HIP(ctx, hipSetDevice(ctx->device));
  VS_HIP(ctx, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
  VS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return VS_OK;
}

/* Page-locked host memory for what a big plan sends to the device.  From pageable memory the runtime copies with a
 * kernel of its own, and that kernel waits until the chip has room -- i.e. until the launch before it has ENDED, when
 * that launch fills every CU for its whole duration as the fused kernel does: the plan of batch k + 1 then goes up
 * behind kernel k instead of next to it (bench.py fresh_batches: upload 1.8 ms instead of 0.3).  From page-locked memory
 * it is a DMA transfer. */
static void *plan_pinned_alloc(void *user, size_t bytes)
{
  vs_ctx *ctx = (vs_ctx *)user;
  void *p = NULL;
  if (hipSetDevice(ctx->device) != hipSuccess) return NULL;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return NULL;
  return p;
}
static void plan_pinned_free(void *user, void *ptr)
{
  (void)user;
  if (ptr) (void)hipHostFree(ptr);
}

static void retire_unref(VsRetire *r)
{
  if (r && --r->refs == 0) {
    (void)hipEventDestroy(r->ev);
    free(r);
  }
}
/* A device block of at least `bytes` for a plan: a retired one of a fitting size (at most twice what is asked for -- a
 * plan of 64 utterances does not sit on the 8 MB of a batch's records) once the launches that read it are over, or a new
 * one.  *cap = what it really holds. */
static hipError_t plan_block_get(vs_ctx *ctx, size_t bytes, void **ptr, size_t *cap)
{
  if (bytes == 0) bytes = 1;
  /* of the fitting blocks the one that has been retired longest: its launches are most likely over already (the block of
   * the plan destroyed a moment ago would make this call wait for a kernel that has only just started) */
  int best = -1;
  for (int k = 0; k < VS_PLAN_CACHE_SLOTS; k++) {
    const VsBlock *b = &ctx->plan_cache[k];
    if (b->ptr && b->bytes >= bytes && b->bytes <= 2 * bytes + 4096 && (best < 0 || b->stamp < ctx->plan_cache[best].stamp)) best = k;
  }
  if (best >= 0) {
    VsBlock *b = &ctx->plan_cache[best];
    hipError_t e = b->retired ? hipEventSynchronize(b->retired->ev) : hipSuccess;
    retire_unref(b->retired);
    *ptr = b->ptr;
    *cap = b->bytes;
    b->ptr = NULL;
    b->retired = NULL;
    if (e == hipSuccess) return hipSuccess;
    (void)hipFree(*ptr); /* cannot tell whether it is still read: not ours to hand on */
    (void)hipGetLastError();
  }
  *cap = bytes;
  return hipMalloc(ptr, bytes);
}
/* ... and back, when its plan is destroyed: behind the plan's last launch (retire: shared by the plan's blocks, NULL if it
 * was never launched).  A full cache gives up its oldest block (hipFree: that one wait for the device is the price of the
 * 33rd retired block). */
static void plan_block_put(vs_ctx *ctx, void *ptr, size_t cap, VsRetire *retire)
{
  if (!ptr) return;
  int slot = -1, oldest = 0;
  for (int k = 0; k < VS_PLAN_CACHE_SLOTS; k++) {
    if (!ctx->plan_cache[k].ptr) {
      slot = k;
      break;
    }
    if (ctx->plan_cache[k].stamp < ctx->plan_cache[oldest].stamp) oldest = k;
  }
  if (slot < 0) {
    VsBlock *b = &ctx->plan_cache[oldest];
    retire_unref(b->retired);
    (void)hipFree(b->ptr);
    b->ptr = NULL;
    slot = oldest;
  }
  VsBlock *b = &ctx->plan_cache[slot];
  if (cap == 0) {
    (void)hipFree(ptr);
    return;
  }
  b->ptr = ptr;
  b->bytes = cap;
  b->retired = retire;
  if (retire) retire->refs++;
  b->stamp = ++ctx->plan_cache_stamp;
}
void vs_plan_cache_release(vs_ctx *ctx)
{
  (void)hipSetDevice(ctx->device);
  for (int k = 0; k < VS_PLAN_CACHE_SLOTS; k++) {
    VsBlock *b = &ctx->plan_cache[k];
    if (!b->ptr) continue;
    retire_unref(b->retired);
    (void)hipFree(b->ptr);
    b->ptr = NULL;
    b->retired = NULL;
  }
}

int vs_plan_create_impl(vs_ctx *ctx, const vs_lane *lanes, size_t n_lanes, size_t n_samples,
                        int mode, vs_plan **out)
{
  const int filter_only = (mode & VS_PLAN_FILTER_ONLY) ? 1 : 0;
  const int zero_copy = (mode & VS_PLAN_ZERO_COPY) ? 1 : 0;
  if (!ctx || !lanes || !out || n_lanes == 0 || n_samples == 0) return VS_ERR_ARG;
  if (n_lanes > (size_t)0x7FFFFFC0 || n_samples > (size_t)0x7FFFFF00) return VS_ERR_UNSUPPORTED;
  *out = NULL;
  if (!zero_copy) {
    const int wrc = vs_copy_path_warm(ctx);
    if (wrc != VS_OK) return wrc;
  }
  const vs_tuning *tune = &ctx->tuning;
  int rc = VS_OK;
  /* the records are made in the context's plan workspace (csrc/vs_planhost.c: worker threads that sleep between plans,
   * record buffers that are kept -- a fresh 8 MB block per plan is two thousand page faults); everything else of this
   * call is released at `done` */
  if (!ctx->planws) {
    ctx->planws = vs_planws_create();
    vs_planws_set_big_allocator(ctx->planws, plan_pinned_alloc, plan_pinned_free, ctx);
  }
  if (!ctx->planws) return VS_ERR_NOMEM;
  VsDevLane *dl = NULL;
  double *costab = NULL;   /* the cos rows, one per distinct T2 */
  size_t costab_len = 0, costab_cap = 0;
  int *row_of_T2 = NULL;   /* first entry of the row of T2 in costab, -1: not built yet */
  double *awide = NULL;    /* wide plans: the 40 taps of every lane record, in the records' (sorted) order */
  void *small_tmp = NULL;  /* stand-in for the page-locked staging block when there is none */
  VsGroupSlot *gmap = NULL; /* mixed rings: which group, which ring depth, which LDS region per (workgroup, slot) */
  vs_plan *p = NULL;
  const double t_host0 = vs_now_ms();
  /* Wavefronts are formed from lanes with similar periods: a generator round costs as much as its longest lane and
   * the cos rows of a wavefront are staged once per distinct T2, so a batch with an F0 sweep (BASELINE config 5) is
   * put in the order of (P, T2, options) before it is cut into groups of 64 (vs_expand_all_ordered_ws: one parallel
   * pass makes the records, a batch whose keys differ is then sorted and gathered).  Placement is internal: each lane
   * still writes its own output row (VsDevLane.row), and a lane's result does not depend on its neighbours.  Stable, so
   * homogeneous batches keep their order. */
  VsBatchStats st; /* gathered by the threads that make the records: no walk over all of them per question */
  rc = vs_expand_all_ordered_ws(ctx->planws, lanes, n_lanes, filter_only, &dl, NULL, &st);
  if (rc != VS_OK) goto done;

  /* vowel -n: one noise width per frame; the rows of that table hold the lane with the MOST frames (the shortest frame) */
  int tmax = st.tmax;
  int min_lframe = st.min_lframe; /* shortest frame of the batch, once any lane asks for output noise */
  const bool any_onoise = st.any_onoise != 0;
  if (!filter_only) {
    /* one cos row per distinct T2, in the order the records meet them; every record learns where its row starts */
    row_of_T2 = (int *)malloc(((size_t)st.max_T2 + 1) * sizeof(int));
    if (!row_of_T2) {
      rc = VS_ERR_NOMEM;
      goto done;
    }
    for (int t = 0; t <= st.max_T2; t++) row_of_T2[t] = -1;
    for (size_t l = 0; l < n_lanes; l++) {
      const int T2 = dl[l].T2;
      if (row_of_T2[T2] < 0) {
        if (costab_len + (size_t)T2 > costab_cap) {
          size_t cap = costab_cap ? 2 * costab_cap : 1024;
          while (cap < costab_len + (size_t)T2) cap *= 2;
          double *grown = (double *)realloc(costab, cap * sizeof(double));
          if (!grown) {
            rc = VS_ERR_NOMEM;
            goto done;
          }
          costab = grown;
          costab_cap = cap;
        }
        vs_cos_row(T2, &costab[costab_len]);
        row_of_T2[T2] = (int)costab_len;
        costab_len += (size_t)T2;
      }
      dl[l].tab_off = row_of_T2[T2];
    }
  }
  /* the tap table rides behind the cos rows (one allocation, one copy): a spare entry, then rows 0..9 the ten tables,
   * then the lanes' own sets; records with a set of their own learn their row here */
  size_t taps_off = 0;
  {
    double *taps = NULL;
    size_t tap_rows = 0;
    rc = vs_tap_table_build(lanes, dl, n_lanes, st.n_custom, &taps, &tap_rows);
    if (rc != VS_OK) goto done;
    taps_off = costab_len + 1;
    double *grown = (double *)realloc(costab, (taps_off + tap_rows * VS_ORDER) * sizeof(double));
    if (!grown) {
      free(taps);
      rc = VS_ERR_NOMEM;
      goto done;
    }
    costab = grown;
    costab[costab_len] = 0.0;
    memcpy(costab + taps_off, taps, tap_rows * VS_ORDER * sizeof(double));
    costab_len = taps_off + tap_rows * VS_ORDER; /* what goes up; the cos rows end at taps_off - 1 */
    free(taps);
  }
  const bool pre1 = st.pre1 != 0;
  const bool wide = st.wide != 0;
  if (!any_onoise) min_lframe = 0;
  if (any_onoise && min_lframe <= 0) {
    rc = VS_ERR_UNSUPPORTED;
    goto done;
  }
  /* Launch shape.  A full chip is 4 x cu_count SIMDs.  The fused kind runs WAVE-SPECIALISED
   * whenever it can: two or three wavefronts per 64 utterances with one job each, coupled through
   * the LDS ring (vs_synth_ws_kernel; which roles and which layout: below, and DESIGN.md section 4.2).
   * The one-wave kernel remains for the source-only and filter-only kinds, the per-cycle log and rings too
   * long for a group to fit the LDS.  (vowel -n is two streaming passes behind whichever kernel wrote the PCM.) */
  const unsigned cus = (unsigned)(ctx->cu_count > 0 ? ctx->cu_count : 256);
  /* Utterances per wavefront: 64 -- unless the longest period of the batch does not fit a 64-column
   * ring (the reference takes any rate but an explicit 22050, flowgen_shimmer.c:535-540, and sizes
   * its buffer by the period, fg:569): then the narrow build of the one-wave kernel serves 16
   * utterances per wavefront on a ring with four times the slots.  Slow (three quarters of every
   * wavefront idle), but it synthesises what used to be VS_ERR_UNSUPPORTED. */
  int group_lanes = VS_WAVE;
  if (!filter_only) {
    int probe = 0;
    if (vs_ring_policy_for(VS_WAVE, tmax, 0, &probe, NULL, 0, 1.7) == VS_ERR_UNSUPPORTED) group_lanes = VS_NARROW_LANES;
  }
  const size_t G = (size_t)group_lanes;
  const unsigned grid = (unsigned)((n_lanes + G - 1) / G);
  int wave_specialised = 1;
  if (tune->kernel == VS_KERNEL_WS) wave_specialised = 1;
  if (tune->kernel == VS_KERNEL_SINGLE) wave_specialised = 0;
  if (filter_only || wide || group_lanes != VS_WAVE) wave_specialised = 0;
  if (wide) {
    awide = (double *)malloc(n_lanes * (size_t)VS_WIDE_ORDER * sizeof(double));
    if (!awide) {
      rc = VS_ERR_NOMEM;
      goto done;
    }
    double A[VS_MAX_NCOEF];
    for (size_t l = 0; l < n_lanes; l++) {
      rc = vs_lane_taps(&lanes[(size_t)dl[l].row], A);
      if (rc != VS_OK) goto done;
      for (int j = 1; j <= VS_WIDE_ORDER; j++) awide[l * VS_WIDE_ORDER + (size_t)(j - 1)] = A[j];
    }
  }
  unsigned wg_per_cu = (grid + cus - 1) / cus;
  if (wg_per_cu < 1) wg_per_cu = 1;
  if (wg_per_cu > 4) wg_per_cu = 4;
  int ltab_entries = 0;
  if (!filter_only) {
    /* cos rows staged per wavefront: the distinct T2 among its 64 lanes, each row rounded up to
     * a multiple of 8 (vs_stage_cos_rows), worst wavefront */
    for (size_t w0 = 0; w0 < n_lanes; w0 += G) {
      int seen[VS_WAVE], nseen = 0, sum = 0;
      for (size_t l = w0; l < n_lanes && l < w0 + G; l++) {
        bool dup = false;
        for (int k = 0; k < nseen; k++) dup = dup || (seen[k] == dl[l].T2);
        if (!dup) {
          seen[nseen++] = dl[l].T2;
          sum += (dl[l].T2 + 7) & ~7;
        }
      }
      if (sum > ltab_entries) ltab_entries = sum;
    }
  }
  int cap = 0; /* default: four workgroups per CU */
  if (wg_per_cu < 4) {
    /* what the other residents of a group's LDS need: its cos rows (a batch of many periods stages up to 64 of them
     * per wavefront) and the progress words of three roles; at least 4 KiB */
    size_t others = (size_t)ltab_entries * sizeof(double) + VS_SYNC_WORDS_3 * VS_WAVE * sizeof(int) + 64;
    if (others < 4096) others = 4096;
    const long room = (long)(VS_LDS_LIMIT / wg_per_cu) - (long)others;
    cap = (int)(room / (long)(group_lanes * 2)) - VS_TRASH_ROWS;
    if (cap < VS_SS) cap = VS_SS; /* the policy lifts it to what the longest cycle needs, or refuses */
  }
  if (tune->ring_slots > 0) cap = tune->ring_slots;
  int slots = 0, ready_min = 32;
  size_t lds_bytes = 0;
  int ws_pairs = 1, ws_pair_bytes = 0, ws_roles = 2, ws_layout = VS_WS_LAYOUT_ROLE_MAJOR, simd_fallback = 0;
  bool all_deep = true; /* every group's ring holds 1.65 of its longest cycles or more (see the thresholds below) */
  size_t n_wg_mixed = 0, mixed_lds = 0; /* mixed rings (below): gmap is [n_wg_mixed][4] */
  int mixed_c_min = 0;
  if (!filter_only) {
    /* Half-filled chips (at most two groups per CU) have LDS to spare: rings of 2.4 of the longest cycle instead of
     * 1.7.  With a SIMD per wavefront nobody fills the gaps a starved filter wavefront leaves, and a deeper ring is what
     * keeps it fed while rounds wait for (nearly) all lanes (config 4's shard, same box: two roles 5.2 / 4.07 ms -> 5.1 /
     * 3.66 ms with 576 slots instead of 408; three roles 4.87 / 4.08 -> 4.80 / 3.40, profiles/r04_config4_roles.txt). */
    const double depth = (wg_per_cu <= 2 && group_lanes == VS_WAVE) ? 2.4 : 1.7;
    rc = vs_ring_policy_for(group_lanes, tmax, cap, &slots, &ready_min, tune->ring_slots, depth);
    if (rc != VS_OK) goto done;
    /* Super-step threshold, per 64-utterance group, from how many of the group's longest cycles
     * its ring holds (rho): a group whose ring holds barely one cycle cannot wait for all of its
     * lanes.  One-wave kernel: the table of vs_ring_policy (replayed period sequences).
     * Wave-specialised kernel, measured (profiles/README.md, r03_kernel_experiments.txt): over a deep
     * ring (rho >= 1.65: BASELINE configs 3 and 4) the filter wavefront waits for ALL of its lanes --
     * they then share one position, and the super-step loop runs without the copy of the window, the
     * exec masks and the per-lane bounds that a wavefront of stragglers costs.  Over shallower rings:
     * three quarters of the lanes when generator and filter share a SIMD (the long periods of config 5's
     * F0 sweep), 62 % when each has its own. */
    const bool ws_shared_simd = wave_specialised && grid > 2u * cus;
    for (size_t w0 = 0; w0 < n_lanes; w0 += G) {
      int tb = 1;
      for (size_t l = w0; l < n_lanes && l < w0 + G; l++)
        if ((int)dl[l].tbound > tb) tb = (int)dl[l].tbound;
      const double rho = (double)(slots - VS_SS) / (double)tb;
      if (rho < 1.65) all_deep = false;
      int thr = 32;
      if (rho >= 1.65) thr = 64;
      else if (rho >= 1.45) thr = 58;
      else if (rho >= 1.33) thr = 48;
      if (wave_specialised) thr = (rho >= 1.65) ? 64 : (ws_shared_simd ? 48 : 40);
      for (size_t l = w0; l < n_lanes && l < w0 + G; l++) dl[l].ready_min = thr;
    }
    /* Mixed rings.  A full grid whose groups differ in period (BASELINE config 5's F0 sweep: P = 53 .. 200) and a
     * uniform ring sized for the longest of them: the long-period groups get barely one cycle (rho 1.1), their filter
     * cannot wait for all lanes, the plan falls back to two roles with the divergent filter loop -- while the
     * short-period groups sit on five cycles' worth of LDS they do not need.  Instead: every group gets the depth ITS
     * periods need (1.7 cycles), and a workgroup is put together from groups ACROSS the period range (sorted by
     * period, dealt to the workgroups in snake order: the longest with the shortest), so that the four rings of a
     * workgroup share the CU's LDS unevenly and still fit.  Every group is deep then, the three-role kernel with the
     * all-lanes filter loop runs everywhere (VsGroupSlot, vs_device.h). */
    if (wave_specialised && !all_deep && ws_shared_simd && group_lanes == VS_WAVE && tune->ring_slots == 0 &&
        tune->mixed_rings >= 0 && (tune->ws_pairs == 0 || tune->ws_pairs == 4)) {
      /* the shallowest ring: 216 slots if the workgroups can afford it (short periods hand over often -- nine
       * super-steps of room instead of six are worth 4 % on config 5, tools/sweep5.sh), else 192, 168, 144; the table
       * itself is built without a device in sight (vs_mixed_rings_build, csrc/vs_planhost.c) */
      int floor_slots = tune->mixed_rings > 1 ? ((tune->mixed_rings + VS_SS - 1) / VS_SS) * VS_SS : 216;
      for (;;) {
        int c_min = 0, c_max = 0;
        const int mrc = vs_mixed_rings_build(dl, n_lanes, floor_slots, &gmap, &n_wg_mixed, &mixed_lds, &c_min, &c_max);
        if (mrc == VS_ERR_NOMEM) {
          rc = mrc;
          goto done;
        }
        if (mrc == VS_OK) {
          for (size_t l = 0; l < n_lanes; l++) dl[l].ready_min = 64; /* every ring is deep: the filter waits for all of its lanes */
          all_deep = true;
          slots = c_max;
          mixed_c_min = c_min;
          break;
        }
        if (tune->mixed_rings > 1 || floor_slots <= 144) break; /* does not fit: uniform rings, as before */
        floor_slots -= VS_SS;
      }
    }
    ready_min = tune->ready_min > 0 ? tune->ready_min : 0; /* 0: the groups' own thresholds */
    /* ring rows + the trash rows (lanes that must not emit write there) + the cos rows */
    /* (mixed rings: what one group WOULD take at the uniform depth of the shallowest ring -- the shape decisions below
     * see a group that fits four to a workgroup, which is what the table guarantees; the launch takes the table's sum) */
    lds_bytes = (size_t)((gmap ? mixed_c_min : slots) + VS_TRASH_ROWS) * G * sizeof(int16_t) + (size_t)ltab_entries * sizeof(double);
    if (lds_bytes > VS_LDS_LIMIT) {
      rc = VS_ERR_UNSUPPORTED;
      goto done;
    }
    /* wave-specialised launch shape: one group = ring + cos rows + its progress words; as many groups
     * per workgroup as give one workgroup per CU AND fit the CU's LDS; when not even one group fits
     * next to its progress words, the one-wave kernel runs instead */
    ws_pair_bytes = (int)((lds_bytes + 2 * VS_WAVE * sizeof(int) + 15) & ~(size_t)15);
    if (wave_specialised) {
      if ((size_t)ws_pair_bytes > VS_LDS_LIMIT) wave_specialised = 0;
      ws_pairs = (grid <= cus) ? 1 : (grid <= 2u * cus ? 2 : 4);
      if (tune->ws_pairs > 0) ws_pairs = tune->ws_pairs;
      while (ws_pairs > 1 && (size_t)ws_pairs * (size_t)ws_pair_bytes > VS_LDS_LIMIT) ws_pairs >>= 1;
      if (gmap) ws_pairs = 4; /* the table's shape; it was checked against the LDS group by group */
      /* Full grids (four groups per workgroup, the wavefronts of a group share a SIMD): three roles
       * -- open phase | noise | filter -- if the extra progress words and order boxes still fit next
       * to four rings.  Otherwise two: generator | filter. */
      const int bytes3 = (int)((lds_bytes + VS_SYNC_WORDS_3 * VS_WAVE * sizeof(int) + 15) & ~(size_t)15);
      /* only over deep rings: the filter wavefront of the three-role kernel waits for ALL of its lanes, which a
       * ring of barely one cycle cannot feed (BASELINE config 5's F0 sweep over uniform rings: 4.4 ms against 3.66 with
       * two roles, profiles/r03_kernel_experiments.txt).  With or without glottal noise: until round 5 batches without it
       * (BASELINE config 2's shape) took two roles -- the third wavefront only relays progress words there, and round 3
       * measured 3.01 ms against 2.75 on the full grid -- but with the kernels as they are now three roles win on every
       * shape measured (round 6, profiles/r06_roles_without_noise.txt: config 2 at 1024 utterances 1.56 against 1.59 ms
       * exact, 0.99 against 1.13 fma, 0.84 against 0.92 f32; at 65536 2.57 / 2.68 exact; an F0 sweep without noise 2.99 /
       * 3.02): the open-phase wavefront hands the bookkeeping of finished cycles to a wavefront that has nothing else to do */
      if (wave_specialised && ws_pairs == 4 && all_deep && (gmap || (size_t)4 * (size_t)bytes3 <= VS_LDS_LIMIT))
        ws_roles = 3;
      /* Half-filled chips (one or two groups per workgroup: BASELINE config 4's shard, the 16384-utterance chunks
       * of the pipelines): the lone filter wavefront is the bound and the one generator wavefront next door takes
       * three quarters of its time (4.65 against 3.52 ms alone, profiles/r04_config4_roles.txt) -- in VS_ARITH_FMA the
       * generator IS the bound.  Three roles with the filter wavefront ALONE on its SIMD and the open-phase and the
       * noise wavefront together on the next one (VS_WS_LAYOUT_SPREAD_2X3; with one group per workgroup the three
       * wavefronts have a SIMD each anyway): the generator's work takes 1.84 ms that way.  The role-major layout
       * of two groups would put the open-phase wavefront on the FILTER's SIMD (6.4 ms). */
      if (wave_specialised && ws_pairs <= 2 && all_deep && (size_t)ws_pairs * (size_t)bytes3 <= VS_LDS_LIMIT)
        ws_roles = 3;
      if (tune->ws_roles == 2) ws_roles = 2;
      if (tune->ws_roles == 3 && (gmap || (size_t)ws_pairs * (size_t)bytes3 <= VS_LDS_LIMIT)) ws_roles = 3;
      /* Both three-role layouts are built on "wavefront w runs on the SIMD of wavefront w % 4": role-major puts the three wavefronts
       * of ONE group on one SIMD (12 wavefronts), the spread layout keeps the filter wavefront alone (8).  Asked of
       * the hardware once per context (a probe launch, vs_ctx_simd_dealing); where it does not hold, two roles --
       * slower than three done right, much faster than three in the wrong order (6.4 against 2.6 ms). */
      if (ws_roles == 3 && ws_pairs > 1) {
        const bool dealt = tune->fault != VS_FAULT_SIMD_DEALING && simd_probe(ctx) == VS_OK &&
                           (ws_pairs == 4 ? ctx->simd_cyclic12 : ctx->simd_cyclic8);
        if (!dealt) {
          ws_roles = 2;
          simd_fallback = 1;
        }
      }
      if (ws_roles == 3) ws_pair_bytes = bytes3;
      if (ws_roles == 3 && ws_pairs == 2) ws_layout = VS_WS_LAYOUT_SPREAD_2X3;
      /* the mixed-rings table was laid out for four groups per workgroup with the progress words of three roles
       * (two roles use fewer of them); any other shape: not built for it */
      if (gmap && ws_pairs != 4) {
        rc = VS_ERR_INTERNAL; /* cannot happen: the table is only made for full grids, which take four groups per workgroup */
        goto done;
      }
    }
    if (gmap && !wave_specialised) {
      rc = VS_ERR_INTERNAL;
      goto done;
    }
  }

  p = (vs_plan *)calloc(1, sizeof(vs_plan));
  if (!p) {
    rc = VS_ERR_NOMEM;
    goto done;
  }
  p->ctx = ctx;
  p->n_lanes = n_lanes;
  p->n_samples = n_samples;
  p->ring_slots = slots;
  p->ready_min = ready_min;
  p->ltab_entries = ltab_entries;
  p->lds_bytes = gmap ? mixed_lds : lds_bytes;
  /* what ONE group needs when a plan is launched on the one-wave kernel (source-only kind, the per-cycle log): a ring of
   * the plan's deepest depth + the worst wavefront's cos rows -- not the four-group sum a mixed-rings workgroup takes */
  p->lds_one_wave = filter_only ? 0 : (size_t)(slots + VS_TRASH_ROWS) * G * sizeof(int16_t) + (size_t)ltab_entries * sizeof(double);
  if (p->lds_one_wave > VS_LDS_LIMIT) {
    free(p);
    p = NULL;
    rc = VS_ERR_UNSUPPORTED;
    goto done;
  }
  p->ring_slots_min = gmap ? mixed_c_min : slots;
  p->grid = grid;
  p->ondw_pitch = min_lframe ? (long)((n_samples + (size_t)min_lframe - 1) / (size_t)min_lframe) : 0;
  /* one frame length for every lane: the filter wavefronts of the wave-specialised kernels take the frame powers along
   * (vs_synth_ws_pow_kernel), the streaming pass fills in what they leave */
  p->pow_lframe = (min_lframe > 0 && wave_specialised && !st.no_lframe && st.max_lframe == min_lframe) ? min_lframe : 0;
  p->wave_specialised = wave_specialised;
  p->ws_pairs = ws_pairs;
  p->ws_roles = ws_roles;
  p->ws_layout = ws_layout;
  p->simd_fallback = simd_fallback;
  p->ws_shared_simd = (wave_specialised && grid > 2u * cus) ? 1 : 0;
  p->group_lanes = group_lanes;
  p->ws_pair_bytes = ws_pair_bytes;
  p->filter_only = filter_only;
  p->pre1 = pre1 ? 1 : 0;
  p->wide = wide ? 1 : 0;
  p->flow_pitch = (n_samples + 7) & ~(size_t)7;
  p->tuning = *tune;

  const double t_host1 = vs_now_ms();
  hipError_t e = hipSetDevice(ctx->device);
  const size_t zc_lanes = n_lanes * sizeof(VsDevLane), zc_cos = (costab_len + 1) * sizeof(double);
  const size_t zc_off_cos = (zc_lanes + 63) & ~(size_t)63, zc_off_err = (zc_off_cos + zc_cos + 63) & ~(size_t)63;
  const size_t zc_off_wide = zc_off_err + 64, zc_wide = wide ? n_lanes * (size_t)VS_WIDE_ORDER * sizeof(double) : 0;
  /* plans that copy: [cos rows + taps | mixed-rings table | error word], at least VS_SMALL_BLOCK_MIN bytes */
  const size_t small_gmap_bytes = gmap ? n_wg_mixed * 4 * sizeof(VsGroupSlot) : 0;
  const size_t small_off_gmap = (zc_cos + 63) & ~(size_t)63, small_off_err = (small_off_gmap + small_gmap_bytes + 63) & ~(size_t)63;
  const size_t small_alloc = (small_off_err + 64 > VS_SMALL_BLOCK_MIN) ? small_off_err + 64 : VS_SMALL_BLOCK_MIN;
  if (zero_copy) {
    /* one pinned, device-mapped host block: the CPU writes the records, the kernel reads them over PCIe (a few
     * hundred bytes per utterance, once), the error word is read back by the CPU -- no copy engine involved */
    void *dev = NULL;
    if (e == hipSuccess) e = hipHostMalloc(&p->zc_host, zc_off_wide + zc_wide + 64, hipHostMallocMapped);
    if (e == hipSuccess) e = hipHostGetDevicePointer(&dev, p->zc_host, 0);
    if (e == hipSuccess) {
      char *hb = (char *)p->zc_host, *db = (char *)dev;
      memcpy(hb, dl, zc_lanes);
      if (costab_len) memcpy(hb + zc_off_cos, costab, costab_len * sizeof(double));
      p->zc_err = (int *)(hb + zc_off_err);
      *p->zc_err = 0;
      if (wide) memcpy(hb + zc_off_wide, awide, zc_wide);
      p->d_lanes = (VsDevLane *)db;
      p->d_costab = (double *)(db + zc_off_cos);
      p->d_err = (int *)(db + zc_off_err);
      if (wide) p->d_awide = (double *)(db + zc_off_wide);
      p->d_taps = p->d_costab + taps_off;
    }
  } else {
    if (e == hipSuccess) e = plan_block_get(ctx, n_lanes * sizeof(VsDevLane), (void **)&p->d_lanes, &p->cap_lanes);
    /* the small parts -- cos rows + tap table, the mixed-rings table, the error word -- share ONE device block that is
     * never smaller than VS_SMALL_BLOCK_MIN and goes up in ONE copy: the runtime moves copies of up to 16 KiB with a
     * kernel of its own, and that kernel waits until the chip has room, i.e. until a fused launch that is running has
     * ENDED (it fills every CU for its whole duration), while a copy of 64 KiB is a DMA transfer that runs next to it --
     * 0.03 ms instead of 2.2 behind a launch (tools/overlap_probe.py, profiles/r05_plan_cost.txt) */
    if (e == hipSuccess) e = plan_block_get(ctx, small_alloc, (void **)&p->d_small, &p->cap_small);
    if (e == hipSuccess) {
      p->d_costab = (double *)p->d_small;
      p->d_err = (int *)(p->d_small + small_off_err);
      if (gmap) p->d_group_map = (VsGroupSlot *)(p->d_small + small_off_gmap);
    }
    if (e == hipSuccess && wide) e = hipMalloc((void **)&p->d_awide, n_lanes * (size_t)VS_WIDE_ORDER * sizeof(double));
    if (e == hipSuccess) p->d_taps = p->d_costab + taps_off;
  }
  if (e == hipSuccess && wave_specialised) e = plan_block_get(ctx, (n_samples + 32) * sizeof(int16_t), (void **)&p->d_sink, &p->cap_sink);
  if (e == hipSuccess && gmap && zero_copy) e = hipMalloc((void **)&p->d_group_map, n_wg_mixed * 4 * sizeof(VsGroupSlot));
  if (e == hipSuccess && p->ondw_pitch)
    e = plan_block_get(ctx, n_lanes * (size_t)p->ondw_pitch * sizeof(float), (void **)&p->d_ondw, &p->cap_ondw);
  if (e == hipSuccess && p->pow_lframe) e = plan_block_get(ctx, n_lanes * sizeof(int32_t), (void **)&p->d_odone, &p->cap_odone);
  if (e == hipSuccess && wide && !filter_only) {
    const size_t flow_bytes = n_lanes * p->flow_pitch * sizeof(int16_t);
    if (mode & VS_PLAN_POOL_SCRATCH) {
      /* grows only with the first (largest) chunk of a pipeline, i.e. while nothing runs */
      if (vs_pool_device(ctx, &ctx->pool.d_flow, &ctx->pool.d_flow_bytes, flow_bytes) != VS_OK) e = hipErrorOutOfMemory;
      p->d_flow = (int16_t *)ctx->pool.d_flow;
      p->owns_flow = 0;
    } else {
      e = hipMalloc((void **)&p->d_flow, flow_bytes);
      p->owns_flow = 1;
    }
  }
  /* the records go up on a stream of the context's own -- inside a chunk pipeline on the pipeline's upload stream --,
   * never on the launch stream: the wait below is for THESE copies and not for whatever kernel the caller has running
   * (the previous chunk's, or batch k's while batch k + 1 is being planned); the launch that uses the records is
   * enqueued after this function has returned, i.e. after the copies have completed */
  hipStream_t up = ctx->upload;
  if (!up && !zero_copy) {
    if (!ctx->own_upload && e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->own_upload, hipStreamNonBlocking);
    up = ctx->own_upload;
  }
  if (!zero_copy) {
    if (e == hipSuccess && wide)
      e = hipMemcpyAsync(p->d_awide, awide, n_lanes * (size_t)VS_WIDE_ORDER * sizeof(double), hipMemcpyHostToDevice, up);
    /* (the error word is zeroed by the copy, not by hipMemsetAsync: a process's first memset loads the runtime's fill kernel, 20 ms) */
    void *small_src = NULL;
    if (e == hipSuccess) {
      if (ctx->plan_pin_bytes < small_alloc) {
        if (ctx->plan_pin) (void)hipHostFree(ctx->plan_pin);
        ctx->plan_pin = NULL;
        ctx->plan_pin_bytes = 0;
        if (hipHostMalloc(&ctx->plan_pin, 2 * small_alloc, hipHostMallocDefault) == hipSuccess) ctx->plan_pin_bytes = 2 * small_alloc;
        else ctx->plan_pin = NULL;
      }
      small_src = ctx->plan_pin;
      if (!small_src) small_src = small_tmp = calloc(1, small_alloc); /* no page-locked memory: a pageable block of the same size */
      if (!small_src) e = hipErrorOutOfMemory;
    }
    if (e == hipSuccess) {
      char *blk = (char *)small_src;
      if (costab_len) memcpy(blk, costab, costab_len * sizeof(double));
      if (small_gmap_bytes) memcpy(blk + small_off_gmap, gmap, small_gmap_bytes);
      memset(blk + small_off_err, 0, 64);
      e = hipMemcpyAsync(p->d_small, blk, small_alloc, hipMemcpyHostToDevice, up);
    }
    if (e == hipSuccess)
      e = hipMemcpyAsync(p->d_lanes, dl, n_lanes * sizeof(VsDevLane), hipMemcpyHostToDevice, up);
    if (e == hipSuccess) e = hipStreamSynchronize(up);
  }
  if (e != hipSuccess) {
    ctx->last_hip_error = (int)e;
    vs_plan_destroy(p); /* frees whatever was allocated */
    p = NULL;
    rc = VS_ERR_HIP;
    goto done;
  }
  const double t_host2 = vs_now_ms();
  p->host_ms = t_host1 - t_host0;
  p->upload_ms = t_host2 - t_host1;
  *out = p;
  rc = VS_OK;
done:
  free(costab);
  free(row_of_T2);
  free(awide);
  free(gmap);
  free(small_tmp);
  return rc;
}

int vs_plan_create(vs_ctx *ctx, const vs_lane *lanes, size_t n_lanes, size_t n_samples,
                              vs_plan **out)
{
  return vs_plan_create_impl(ctx, lanes, n_lanes, n_samples, 0, out);
}

void vs_plan_destroy(vs_plan *p)
{
  if (!p) return;
  (void)hipSetDevice(p->ctx->device);
  /* the blocks every plan has go back to the context's cache, behind the launches that still read them (no hipFree, which
   * would wait for the device: plan_block_put); the rare ones are freed */
  vs_ctx *ctx = p->ctx;
  VsRetire *retire = NULL;
  if (p->last_launch) {
    retire = (VsRetire *)malloc(sizeof(VsRetire));
    if (retire) {
      retire->ev = p->last_launch;
      retire->refs = 1; /* ours, until the blocks have theirs */
    } else { /* no memory for 16 bytes: wait here instead */
      (void)hipEventSynchronize(p->last_launch);
      (void)hipEventDestroy(p->last_launch);
    }
  }
  if (p->zc_host) {
    (void)hipHostFree(p->zc_host); /* records, cos rows, error word and wide taps of a zero-copy plan */
  } else {
    plan_block_put(ctx, p->d_lanes, p->cap_lanes, retire);
    plan_block_put(ctx, p->d_small, p->cap_small, retire); /* cos rows + taps, the mixed-rings table, the error word */
    if (p->d_awide) (void)hipFree(p->d_awide);
  }
  plan_block_put(ctx, p->d_sink, p->cap_sink, retire);
  if (p->d_seeds) (void)hipFree(p->d_seeds);
  if (p->h_seeds) (void)hipHostFree(p->h_seeds);
  if (p->seeds_copied) (void)hipEventDestroy(p->seeds_copied);
  if (p->d_group_map && !p->d_small) (void)hipFree(p->d_group_map); /* (inside d_small when the plan copies) */
  plan_block_put(ctx, p->d_ondw, p->cap_ondw, retire);
  plan_block_put(ctx, p->d_odone, p->cap_odone, retire);
  if (p->d_flow && p->owns_flow) (void)hipFree(p->d_flow);
  retire_unref(retire);
  free(p);
}

/* behind every launch of a plan: the event its device blocks will retire behind (plan_block_put) */
static int plan_mark_launch(vs_plan *p)
{
  vs_ctx *ctx = p->ctx;
  if (!p->last_launch) VS_HIP(ctx, hipEventCreateWithFlags(&p->last_launch, hipEventDisableTiming));
  /* a caller who moved the context to another stream between two launches of this plan: the new stream waits for the old
   * launches first, so that the event recorded on it still stands behind EVERY launch that reads the plan's blocks */
  else if (p->last_stream != ctx->stream) VS_HIP(ctx, hipStreamWaitEvent(ctx->stream, p->last_launch, 0));
  VS_HIP(ctx, hipEventRecord(p->last_launch, ctx->stream));
  p->last_stream = ctx->stream;
  return VS_OK;
}

int vs_plan_status(vs_plan *p, int *flags)
{
  if (!p) return VS_ERR_ARG;
  vs_ctx *ctx = p->ctx;
  int word = 0;
  VS_HIP(ctx, hipSetDevice(ctx->device));
  if (p->zc_err) { /* zero-copy plan: the word lives in host memory the device writes through PCIe */
    VS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    word = *(volatile int *)p->zc_err;
  } else {
    VS_HIP(ctx, hipMemcpyAsync(&word, p->d_err, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    VS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  if (flags) *flags = word;
  return word ? VS_ERR_INTERNAL : VS_OK;
}

int vs_plan_reseed(vs_plan *p, const uint64_t *seeds, const uint64_t *out_seeds)
{
  if (!p || !seeds) return VS_ERR_ARG;
  if (p->filter_only && !out_seeds) out_seeds = seeds;
  if (!out_seeds) out_seeds = seeds;
  vs_ctx *ctx = p->ctx;
  VS_HIP(ctx, hipSetDevice(ctx->device));
  const size_t n = p->n_lanes;
  if (p->zc_host) return VS_ERR_INTERNAL; /* (the small calls' zero-copy plans live for one call inside vs_source / vs_filter) */
  if (!p->seeds_copied) {
    /* all three or none: a first call that fails half-way leaves the plan as it was, and the next one starts over */
    unsigned long long *d = NULL, *h = NULL;
    hipEvent_t ev = NULL;
    hipError_t e = hipMalloc((void **)&d, 2 * n * sizeof(uint64_t));
    if (e == hipSuccess) e = hipHostMalloc((void **)&h, 2 * n * sizeof(uint64_t), hipHostMallocDefault);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (e != hipSuccess) {
      if (d) (void)hipFree(d);
      if (h) (void)hipHostFree(h);
      ctx->last_hip_error = (int)e;
      return e == hipErrorOutOfMemory ? VS_ERR_NOMEM : VS_ERR_HIP;
    }
    p->d_seeds = d;
    p->h_seeds = h;
    p->seeds_copied = ev;
  } else {
    VS_HIP(ctx, hipEventSynchronize(p->seeds_copied)); /* the previous reseed's upload has left the pinned buffer */
  }
  /* the caller's arrays are his again when this returns: they go through the plan's pinned buffer */
  memcpy(p->h_seeds, seeds, n * sizeof(uint64_t));
  memcpy(p->h_seeds + n, out_seeds, n * sizeof(uint64_t));
  /* on the launch stream: behind the launches that still read the old seeds, in front of the next one */
  VS_HIP(ctx, hipMemcpyAsync(p->d_seeds, p->h_seeds, 2 * n * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
  VS_HIP(ctx, hipEventRecord(p->seeds_copied, ctx->stream));
  VS_HIP(ctx, vs_launch_reseed(p->d_lanes, p->d_seeds, p->d_seeds + n, (int)n, ctx->stream));
  return plan_mark_launch(p); /* (the reseed kernel writes the records: the blocks retire behind it) */
}

/* diagnostic builds (tools/diag_bench.py): device buffer of grid*8 uint64 cycle counters */
int vs_plan_set_diag(vs_plan *p, void *diag_dev)
{
  if (!p) return VS_ERR_ARG;
  p->d_diag = (unsigned long long *)diag_dev;
  return VS_OK;
}

int vs_plan_timing(const vs_plan *p, double *host_ms, double *upload_ms)
{
  if (!p) return VS_ERR_ARG;
  if (host_ms) *host_ms = p->host_ms;
  if (upload_ms) *upload_ms = p->upload_ms;
  return VS_OK;
}

int vs_plan_kernel_name(const vs_plan *p, int kind, char *buf, size_t len)
{
  if (!p || !buf || len == 0) return VS_ERR_ARG;
  /* VS_ARITH_F32 is the wave-specialised kernels' alone: everything else runs VS_ARITH_FMA for it */
  const int one_wave_arith = p->ctx->arith == VS_ARITH_F32 ? VS_ARITH_FMA : p->ctx->arith;
  if (p->wide && kind != VS_KIND_SOURCE) {
    snprintf(buf, len, "%svs_filter_wide_kernel<%d>", kind == VS_KIND_SYNTH ? "vs_synth_kernel<0, 1, false, false> + " : "",
             one_wave_arith);
    return VS_OK;
  }
  const int ws = p->wave_specialised && kind == VS_KIND_SYNTH;
  const int pre1 = p->pre1 && p->ctx->arith == VS_ARITH_EXACT && kind != VS_KIND_SOURCE;
  if (ws) /* both arithmetic contracts have a pre-emphasis-1.0 instantiation of the wave-specialised kernels */
    snprintf(buf, len, "vs_synth_ws_%skernel<%d, %s, %d>", (p->d_ondw && p->pow_lframe) ? "pow_" : "", p->ctx->arith,
             p->pre1 ? "true" : "false", p->ws_roles);
  else
    snprintf(buf, len, "vs_synth_kernel<%d, %d, false, %s>%s", kind == VS_KIND_SOURCE ? 0 : one_wave_arith, kind,
             pre1 ? "true" : "false", p->group_lanes != VS_WAVE ? " (narrow build: 16 utterances per wavefront)" : "");
  if (p->d_ondw && kind != VS_KIND_SOURCE) { /* vowel -n: the two passes behind it */
    const size_t used = strlen(buf);
    snprintf(buf + used, len - used, " + vs_out_power_%skernel + vs_out_noise_kernel", (ws && p->pow_lframe) ? "fill_" : "");
  }
  return VS_OK;
}

int vs_plan_roles(const vs_plan *p, int *roles, int *layout, int *simd_fallback)
{
  if (!p) return VS_ERR_ARG;
  if (roles) *roles = p->wave_specialised ? p->ws_roles : 1;
  if (layout) *layout = p->wave_specialised ? p->ws_layout : 0;
  if (simd_fallback) *simd_fallback = p->simd_fallback;
  return VS_OK;
}

int vs_plan_info(const vs_plan *p, size_t *lds_bytes, size_t *n_workgroups,
                            size_t *ring_slots)
{
  if (!p) return VS_ERR_ARG;
  if (lds_bytes) *lds_bytes = p->lds_bytes;
  if (n_workgroups) *n_workgroups = p->grid;
  if (ring_slots) *ring_slots = (size_t)p->ring_slots;
  return VS_OK;
}

int vs_plan_launch(vs_plan *p, int kind, const int16_t *in_dev, size_t in_pitch,
                              int16_t *out_dev, size_t out_pitch, vs_cycle_rec *log_dev,
                              size_t log_pitch, int32_t *ncyc_dev)
{
  if (!p || !out_dev) return VS_ERR_ARG;
  if (kind != VS_KIND_SYNTH && kind != VS_KIND_SOURCE && kind != VS_KIND_FILTER) return VS_ERR_ARG;
  if (out_pitch < p->n_samples) return VS_ERR_ARG;
  if (kind == VS_KIND_FILTER && (!in_dev || in_pitch < p->n_samples)) return VS_ERR_ARG;
  if (p->filter_only && kind != VS_KIND_FILTER) return VS_ERR_ARG;
  if (log_dev && log_pitch == 0) return VS_ERR_ARG;
  vs_ctx *ctx = p->ctx;
  VsKernelArgs a;
  memset(&a, 0, sizeof(a));
  a.lanes = p->d_lanes;
  a.costab = p->d_costab;
  a.taps = p->d_taps;
  a.in = in_dev;
  a.out = out_dev;
  a.log = (kind == VS_KIND_FILTER) ? NULL : (void *)log_dev;
  a.ncyc = ncyc_dev;
  a.in_pitch = (long)in_pitch;
  a.out_pitch = (long)out_pitch;
  a.log_pitch = (long)log_pitch;
  a.n_lanes = (int)p->n_lanes;
  a.n_samples = (int)p->n_samples;
  a.ring_slots = p->ring_slots;
  a.ltab_entries = p->ltab_entries;
  a.ready_min = p->ready_min;
  /* a SIMD per wavefront and fused multiply-adds: the filter is as quick as the generator and does
   * better not to wait for the last lane (see the thresholds in vs_plan_create_impl) */
  if (p->wave_specialised && !p->ws_shared_simd && p->ctx->arith != VS_ARITH_EXACT && a.ready_min == 0) a.ready_min = 40;
  a.diag = p->d_diag;
  a.err = p->d_err;
  a.sink = p->d_sink;
  a.ondw = (kind == VS_KIND_SOURCE) ? NULL : p->d_ondw;
  a.ondw_pitch = p->ondw_pitch;
  a.ws_pairs = p->ws_pairs;
  a.ws_roles = p->ws_roles;
  a.ws_layout = p->ws_layout;
  a.group_map = p->d_group_map;
  a.group_lanes = p->group_lanes;
  a.ws_pair_bytes = p->ws_pair_bytes;
  /* a generator round starts when gen_min/64 of the lanes that still need cycles have room -- or at
   * once when a lane is about to run its filter dry (fewer than gen_low samples buffered).  The
   * open-phase wavefront of the three-role kernel is idle two thirds of the time and would start
   * rounds for half of the lanes all day long (209 rounds at 58 % attendance instead of 141 at 86 %,
   * profiles/r03_kernel_experiments.txt): it waits for everybody unless a lane is nearly dry. */
  a.gen_min = p->tuning.gen_min > 0 ? p->tuning.gen_min : (p->ws_roles == 3 ? 64 : (p->ws_pairs == 4 ? 32 : 16));
  a.gen_low = p->tuning.gen_low > 0 ? p->tuning.gen_low : (p->ws_roles == 3 ? 32 : 2 * VS_SS);
  /* three roles on a half-filled chip (the filter wavefront alone on its SIMD, deep rings): rounds for three quarters
   * of the lanes, and at once for a lane that is down to 144 samples -- its filter's SIMD idles while it waits
   * (sweep: profiles/r04_config4_roles.txt) */
  if (p->ws_roles == 3 && p->ws_pairs <= 2) {
    if (p->tuning.gen_min <= 0) a.gen_min = 48;
    if (p->tuning.gen_low <= 0) a.gen_low = 144;
  }
  /* three roles over mixed rings (an F0 sweep): the short-period groups hand over twice as often per sample as
   * config 3's and their rings are shallow in SAMPLES (216 slots = 9 super-steps): a round starts for three quarters of
   * the lanes, and at once for a lane that is down to 56 samples (tools/sweep5.sh: 3.36 -> 3.22 ms with 192 / 48, another
   * 1 % exact and 3 % fma with 216 / 56, profiles/r05_config5_sweep.txt) */
  if (p->ws_roles == 3 && p->d_group_map) {
    if (p->tuning.gen_min <= 0) a.gen_min = 48;
    if (p->tuning.gen_low <= 0) a.gen_low = 56;
  }
  a.spin_limit = p->tuning.spin_limit > 0 ? p->tuning.spin_limit : (1 << 22);
  a.fault = p->tuning.fault;
  a.ws_filter_prio = p->tuning.ws_filter_prio == 0 ? 3 : (p->tuning.ws_filter_prio < 0 ? 0 : p->tuning.ws_filter_prio);
  /* 16-byte vector stores need every row start 4-byte aligned */
  int vec = ((out_pitch & 1) == 0) && ((((uintptr_t)out_dev) & 3) == 0);
  if (kind == VS_KIND_FILTER) vec = vec && ((in_pitch & 1) == 0) && ((((uintptr_t)in_dev) & 3) == 0);
  a.vec_ok = vec;
  VS_HIP(ctx, hipSetDevice(ctx->device));
  if (p->wide && kind != VS_KIND_SOURCE) {
    /* 23..40 taps: un-fused.  The source kernel leaves the flow in HBM, the wide filter kernel
     * reads it back (its input is the caller's for the filter-only kind). */
    a.awide = p->d_awide;
    if (kind == VS_KIND_SYNTH) {
      VsKernelArgs src = a;
      src.out = p->d_flow;
      src.out_pitch = (long)p->flow_pitch;
      src.ondw = NULL;
      src.vec_ok = 1; /* rows of the plan's own buffer start 16-byte aligned */
      VS_HIP(ctx, vs_launch_kernel(ctx->arith, VS_KIND_SOURCE, src.log != NULL, false, false, &src, p->grid,
                                   p->lds_one_wave, ctx->stream));
      a.in = p->d_flow;
      a.in_pitch = (long)p->flow_pitch;
      a.log = NULL;
      a.ncyc = NULL;
    }
    VS_HIP(ctx, vs_launch_filter_wide(ctx->arith, &a, p->grid, ctx->stream));
  } else {
    /* (the launcher takes the wave-specialised kernels for the fused kind without a log only) */
    const int ws = p->wave_specialised != 0 && kind == VS_KIND_SYNTH && a.log == NULL;
    if (ws && a.ondw && p->pow_lframe) { /* -> vs_synth_ws_pow_kernel */
      a.odone = p->d_odone;
      a.pow_lframe = p->pow_lframe;
    }
    VS_HIP(ctx, vs_launch_kernel(ctx->arith, kind, a.log != NULL, ws != 0, p->pre1 != 0, &a, p->grid,
                                 ws ? p->lds_bytes : p->lds_one_wave, ctx->stream));
  }
  /* vowel -n (vowel_new.c:302-324): two streaming passes over the finished PCM -- every frame's power, then the noise */
  if (a.ondw) VS_HIP(ctx, vs_launch_out_noise(&a, ctx->stream));
  return plan_mark_launch(p);
}

