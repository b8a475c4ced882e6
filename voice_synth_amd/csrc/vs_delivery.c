/*
 * vs_delivery.c -- the host-buffer entry points of the C ABI: vs_synth_rows() (finished rows
 * handed to a callback in pinned chunks while later chunks are still being synthesised),
 * vs_synth() on top of it, vs_source() / vs_filter(), pinned host memory for callers.
 *
 * What these replace in the reference is its output path: flowgen_shimmer.c:413-421 and
 * vowel_new.c:327 fwrite each cycle / frame as it is produced.  Here the PCM of a batch is
 * produced in HBM and has to cross PCIe (63 GB/s), which takes ~10x longer than synthesising it,
 * so the crossing is what is organised:
 *
 *   compute chunks   the batch is cut into chunks of at most VS_COMPUTE_CHUNK utterances; chunk
 *                    k+1 is planned on the host and synthesised on the device while chunk k is
 *                    being delivered (two device buffers, one event each);
 *   delivery         VS_DELIVERY_THREADS workers, each with its own stream and its own 16 MiB
 *                    pinned staging buffer, take row blocks of a finished chunk: DMA into the
 *                    staging buffer, then the callback (vs_synth: one memcpy into the caller's
 *                    pageable buffer; vs_batch: the .wav files).  Several DMAs and several
 *                    callbacks are in flight at once; a caller that passes PINNED memory
 *                    (vs_host_alloc) gets the DMA straight into it, no staging, no memcpy.
 *
 * The PCM buffers and the pinned staging buffers belong to the context and are reused by later
 * calls (vs_ctx_trim releases them): once they have reached their size no call allocates them
 * again, and hipFree of them (which waits for the device) never happens while a pipeline runs.
 * What IS allocated per chunk are the plan's own small records (lane records, cos rows, error
 * word: vs_plan_create_impl), uploaded on a stream of their own beside the running kernel.
 *
 * C11 + pthreads; the HIP runtime through its C API.
 */
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "vs_internal.h"

#define VS_COMPUTE_CHUNK 16384 /* utterances per compute chunk of the host-buffer pipeline */

/* ------------------------------------------------------------------------------------------
 * context pool
 * ---------------------------------------------------------------------------------------- */
int vs_pool_device(vs_ctx *ctx, void **ptr, size_t *have, size_t bytes)
{
  if (*ptr && *have >= bytes) return VS_OK;
  VS_HIP(ctx, hipSetDevice(ctx->device));
  if (*ptr) {
    VS_HIP(ctx, hipFree(*ptr)); /* waits for the device: only between pipelines */
    *ptr = NULL;
    *have = 0;
  }
  const size_t want = (bytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
  VS_HIP(ctx, hipMalloc(ptr, want));
  *have = want;
  return VS_OK;
}

int vs_pool_streams(vs_ctx *ctx, size_t row_bytes)
{
  VsPool *P = &ctx->pool;
  /* a staging buffer holds at least one whole row (callbacks receive whole rows) */
  size_t want = VS_STAGING_BYTES;
  if (row_bytes > want) want = (row_bytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
  if (P->streams_ready && P->staging_bytes >= want) return VS_OK;
  /* Anything below may fail half way (VS_HIP returns): until ALL of it has succeeded the pool says
   * "not ready, no staging", so that the next call starts over -- frees whatever is there and
   * allocates afresh -- instead of trusting a staging pointer that a failed call left NULL. */
  const size_t had = P->staging_bytes;
  P->streams_ready = 0;
  P->staging_bytes = 0;
  VS_HIP(ctx, hipSetDevice(ctx->device));
  for (int t = 0; t < VS_DELIVERY_THREADS; t++) {
    if (!P->copy_stream[t]) VS_HIP(ctx, hipStreamCreateWithFlags(&P->copy_stream[t], hipStreamNonBlocking));
    if (P->staging[t] && had < want) { /* only between pipelines */
      VS_HIP(ctx, hipHostFree(P->staging[t]));
      P->staging[t] = NULL;
    }
    if (!P->staging[t]) VS_HIP(ctx, hipHostMalloc(&P->staging[t], want, hipHostMallocDefault));
  }
  P->staging_bytes = want;
  if (!P->compute_stream) VS_HIP(ctx, hipStreamCreateWithFlags(&P->compute_stream, hipStreamNonBlocking));
  if (!P->upload_stream) VS_HIP(ctx, hipStreamCreateWithFlags(&P->upload_stream, hipStreamNonBlocking));
  for (int k = 0; k < 2; k++)
    if (!P->done[k]) VS_HIP(ctx, hipEventCreateWithFlags(&P->done[k], hipEventDisableTiming));
  P->streams_ready = 1;
  return VS_OK;
}

void vs_pool_release(vs_ctx *ctx)
{
  VsPool *P = &ctx->pool;
  (void)hipSetDevice(ctx->device);
  for (int k = 0; k < 2; k++) {
    if (P->d_out[k]) (void)hipFree(P->d_out[k]);
    if (P->done[k]) (void)hipEventDestroy(P->done[k]);
  }
  if (P->d_in) (void)hipFree(P->d_in);
  if (P->d_aux) (void)hipFree(P->d_aux);
  if (P->d_flow) (void)hipFree(P->d_flow);
  if (P->zc_io) (void)hipHostFree(P->zc_io);
  for (int t = 0; t < VS_DELIVERY_THREADS; t++) {
    if (P->staging[t]) (void)hipHostFree(P->staging[t]);
    if (P->copy_stream[t]) (void)hipStreamDestroy(P->copy_stream[t]);
  }
  if (P->compute_stream) (void)hipStreamDestroy(P->compute_stream);
  if (P->upload_stream) (void)hipStreamDestroy(P->upload_stream);
  memset(P, 0, sizeof(*P));
}

int vs_ctx_trim(vs_ctx *ctx)
{
  if (!ctx) return VS_ERR_ARG;
  vs_pool_release(ctx);
  vs_plan_cache_release(ctx);
  return VS_OK;
}

int vs_host_alloc(vs_ctx *ctx, size_t bytes, void **ptr)
{
  if (!ctx || !ptr) return VS_ERR_ARG;
  VS_HIP(ctx, hipSetDevice(ctx->device));
  VS_HIP(ctx, hipHostMalloc(ptr, bytes ? bytes : 1, hipHostMallocDefault));
  return VS_OK;
}

int vs_host_free(vs_ctx *ctx, void *ptr)
{
  if (!ctx) return VS_ERR_ARG;
  VS_HIP(ctx, hipSetDevice(ctx->device));
  VS_HIP(ctx, hipHostFree(ptr));
  return VS_OK;
}

/* is p pinned host memory the device can DMA into? */
static bool vs_is_pinned(const void *p)
{
  hipPointerAttribute_t at;
  memset(&at, 0, sizeof(at));
  if (hipPointerGetAttributes(&at, p) != hipSuccess) {
    (void)hipGetLastError(); /* plain malloc memory: not an error for the caller */
    return false;
  }
  return at.type == hipMemoryTypeHost;
}

/* ------------------------------------------------------------------------------------------
 * vs_synth_rows: the pipeline
 * ---------------------------------------------------------------------------------------- */
typedef struct Block {   /* rows [row0, row0 + rows) of the batch, sitting in device buffer buf */
  size_t row0, rows;
  int buf;
  size_t buf_row0;       /* first row of the chunk that buffer holds */
} Block;

typedef struct Pipe {
  vs_ctx *ctx;
  size_t n_samples, pitch; /* samples per device row */
  vs_rows_cb cb;
  void *user;
  int16_t *direct;         /* pinned destination [n_lanes][n_samples], or NULL: staging + callback */
  pthread_mutex_t mu;
  pthread_cond_t cv_work, cv_free;
  Block *work;             /* queue of blocks: work[head .. tail) */
  size_t head, tail, cap;
  int pending[2];          /* blocks of the chunk in d_out[k] not delivered yet */
  bool closing;
  int rc;                  /* first failure: VS_ERR_*; callbacks' non-zero becomes VS_ERR_IO.  Under mu. */
  int hip_error;
} Pipe;

typedef struct Worker {
  Pipe *pipe;
  int t;
} Worker;

static void pipe_fail(Pipe *p, int rc, int hip_error)
{
  pthread_mutex_lock(&p->mu);
  if (p->rc == VS_OK) {
    p->rc = rc;
    p->hip_error = hip_error;
  }
  pthread_mutex_unlock(&p->mu);
}

static int pipe_rc(Pipe *p)
{
  pthread_mutex_lock(&p->mu);
  const int rc = p->rc;
  pthread_mutex_unlock(&p->mu);
  return rc;
}

static void *worker(void *arg)
{
  Worker *w = (Worker *)arg;
  Pipe *p = w->pipe;
  const int t = w->t;
  vs_ctx *ctx = p->ctx;
  VsPool *P = &ctx->pool;
  if (hipSetDevice(ctx->device) != hipSuccess) pipe_fail(p, VS_ERR_HIP, 0);
  for (;;) {
    Block b;
    int rc_now;
    pthread_mutex_lock(&p->mu);
    while (p->head == p->tail && !p->closing) pthread_cond_wait(&p->cv_work, &p->mu);
    if (p->head == p->tail) {
      pthread_mutex_unlock(&p->mu);
      return NULL;
    }
    b = p->work[p->head++];
    rc_now = p->rc;
    pthread_mutex_unlock(&p->mu);
    if (rc_now == VS_OK) {
      const int16_t *src = (const int16_t *)P->d_out[b.buf] + (b.row0 - b.buf_row0) * p->pitch;
      int16_t *dst = p->direct ? p->direct + b.row0 * p->n_samples : (int16_t *)P->staging[t];
      hipError_t e = hipStreamWaitEvent(P->copy_stream[t], P->done[b.buf], 0);
      if (e == hipSuccess) {
        if (p->pitch == p->n_samples) /* rows are contiguous on both sides: one linear DMA */
          e = hipMemcpyAsync(dst, src, b.rows * p->n_samples * 2, hipMemcpyDeviceToHost, P->copy_stream[t]);
        else
          e = hipMemcpy2DAsync(dst, p->n_samples * 2, src, p->pitch * 2, p->n_samples * 2, b.rows,
                               hipMemcpyDeviceToHost, P->copy_stream[t]);
      }
      if (e == hipSuccess) e = hipStreamSynchronize(P->copy_stream[t]);
      if (e != hipSuccess) pipe_fail(p, VS_ERR_HIP, (int)e);
      else if (p->cb && p->cb(p->user, b.row0, b.rows, dst) != 0) pipe_fail(p, VS_ERR_IO, 0);
    }
    pthread_mutex_lock(&p->mu);
    p->pending[b.buf]--;
    pthread_cond_broadcast(&p->cv_free);
    pthread_mutex_unlock(&p->mu);
  }
}

static int vs_synth_rows_impl(vs_ctx *ctx, const vs_lane *lanes, size_t n_lanes, size_t n_samples,
                              vs_rows_cb cb, void *user, int16_t *direct)
{
  if (!ctx || !lanes || n_lanes == 0 || n_samples == 0 || (!cb && !direct)) return VS_ERR_ARG;
  int rc = vs_pool_streams(ctx, direct ? 0 : n_samples * sizeof(int16_t)); /* staging blocks hold whole rows */
  if (rc != VS_OK) return rc;
  VsPool *P = &ctx->pool;
  const size_t pitch = (n_samples + 7) & ~(size_t)7; /* device rows start 16-byte aligned */
  const size_t chunk = n_lanes < VS_COMPUTE_CHUNK ? n_lanes : VS_COMPUTE_CHUNK;
  for (int k = 0; k < 2 && (k == 0 || n_lanes > chunk); k++) {
    rc = vs_pool_device(ctx, &P->d_out[k], &P->d_out_bytes[k], chunk * pitch * sizeof(int16_t));
    if (rc != VS_OK) return rc;
  }
  size_t rows_per_block = P->staging_bytes / (n_samples * sizeof(int16_t));
  if (rows_per_block < 1) rows_per_block = 1;
  if (direct && rows_per_block < 1024) rows_per_block = 1024; /* no staging limit: fewer, larger DMAs */

  const size_t n_chunks = (n_lanes + chunk - 1) / chunk;
  const size_t blocks_per_chunk = (chunk + rows_per_block - 1) / rows_per_block;
  Pipe pipe;
  memset(&pipe, 0, sizeof(pipe));
  pipe.ctx = ctx;
  pipe.n_samples = n_samples;
  pipe.pitch = pitch;
  pipe.cb = cb;
  pipe.user = user;
  pipe.direct = direct;
  pipe.rc = VS_OK;
  pipe.cap = n_chunks * blocks_per_chunk; /* every block of the call is queued exactly once */
  pipe.work = (Block *)malloc((pipe.cap ? pipe.cap : 1) * sizeof(Block));
  vs_plan **plans = (vs_plan **)calloc(n_chunks, sizeof(vs_plan *));
  if (!pipe.work || !plans) {
    free(pipe.work);
    free(plans);
    return VS_ERR_NOMEM;
  }
  pthread_mutex_init(&pipe.mu, NULL);
  pthread_cond_init(&pipe.cv_work, NULL);
  pthread_cond_init(&pipe.cv_free, NULL);
  pthread_t th[VS_DELIVERY_THREADS];
  Worker wk[VS_DELIVERY_THREADS];
  int n_th = 0;
  for (int t = 0; t < VS_DELIVERY_THREADS; t++) {
    wk[t].pipe = &pipe;
    wk[t].t = t;
    if (pthread_create(&th[t], NULL, worker, &wk[t]) != 0) break;
    n_th++;
  }
  size_t n_plans = 0;
  /* the caller's stream if there is one (its work is ordered before ours), else the context's own */
  hipStream_t cs = ctx->stream ? ctx->stream : P->compute_stream;
  hipStream_t saved = ctx->stream;
  ctx->stream = cs; /* vs_plan_launch and vs_plan_status work on the context's stream */
  ctx->upload = P->upload_stream; /* vs_plan_create_impl: the next chunk's records go up beside this chunk's kernel */
  if (n_th < VS_DELIVERY_THREADS) {
    rc = VS_ERR_NOMEM;
  } else {
    int k = 0;
    for (size_t row0 = 0; row0 < n_lanes && pipe_rc(&pipe) == VS_OK; row0 += chunk, k ^= 1) {
      const size_t rows = (n_lanes - row0 < chunk) ? (n_lanes - row0) : chunk;
      vs_plan *plan = NULL;
      /* host work (expansion, sort, cos rows), three small hipMalloc for the plan's own records, the
       * upload on its own stream: all of it while the previous chunk's kernel runs */
      rc = vs_plan_create_impl(ctx, lanes + row0, rows, n_samples, VS_PLAN_POOL_SCRATCH, &plan);
      if (rc != VS_OK) break;
      plans[n_plans++] = plan;
      /* the buffer this chunk goes into must have been delivered */
      pthread_mutex_lock(&pipe.mu);
      while (pipe.pending[k] != 0) pthread_cond_wait(&pipe.cv_free, &pipe.mu);
      pthread_mutex_unlock(&pipe.mu);
      rc = vs_plan_launch(plan, VS_KIND_SYNTH, NULL, 0, (int16_t *)P->d_out[k], pitch, NULL, 0, NULL);
      if (rc != VS_OK) break;
      hipError_t e = hipEventRecord(P->done[k], cs);
      if (e != hipSuccess) {
        ctx->last_hip_error = (int)e;
        rc = VS_ERR_HIP;
        break;
      }
      pthread_mutex_lock(&pipe.mu);
      for (size_t r = 0; r < rows; r += rows_per_block) {
        Block b;
        b.row0 = row0 + r;
        b.rows = (rows - r < rows_per_block) ? (rows - r) : rows_per_block;
        b.buf = k;
        b.buf_row0 = row0;
        pipe.work[pipe.tail++] = b;
        pipe.pending[k]++;
      }
      pthread_cond_broadcast(&pipe.cv_work);
      pthread_mutex_unlock(&pipe.mu);
    }
  }
  pthread_mutex_lock(&pipe.mu);
  while (pipe.pending[0] != 0 || pipe.pending[1] != 0) pthread_cond_wait(&pipe.cv_free, &pipe.mu);
  pipe.closing = true;
  pthread_cond_broadcast(&pipe.cv_work);
  pthread_mutex_unlock(&pipe.mu);
  for (int t = 0; t < n_th; t++) pthread_join(th[t], NULL);
  if (rc == VS_OK && pipe.rc != VS_OK) {
    rc = pipe.rc;
    if (rc == VS_ERR_HIP) ctx->last_hip_error = pipe.hip_error;
  }
  /* the launches' health words (bounded waits of the wave-specialised kernel), then the plans */
  for (size_t i = 0; i < n_plans; i++) {
    const int st = vs_plan_status(plans[i], NULL);
    if (rc == VS_OK && st != VS_OK) rc = st;
  }
  for (size_t i = 0; i < n_plans; i++) vs_plan_destroy(plans[i]);
  ctx->stream = saved;
  ctx->upload = NULL;
  pthread_cond_destroy(&pipe.cv_free);
  pthread_cond_destroy(&pipe.cv_work);
  pthread_mutex_destroy(&pipe.mu);
  free(pipe.work);
  free(plans);
  return rc;
}

int vs_synth_rows(vs_ctx *ctx, const vs_lane *lanes, size_t n_lanes, size_t n_samples, vs_rows_cb cb, void *user)
{
  if (!cb) return VS_ERR_ARG;
  return vs_synth_rows_impl(ctx, lanes, n_lanes, n_samples, cb, user, NULL);
}

typedef struct CopyOut {
  int16_t *pcm;
  size_t n_samples;
} CopyOut;

static int copy_rows(void *user, size_t row0, size_t rows, const int16_t *pcm)
{
  CopyOut *c = (CopyOut *)user;
  memcpy(c->pcm + row0 * c->n_samples, pcm, rows * c->n_samples * sizeof(int16_t));
  return 0;
}

int vs_synth(vs_ctx *ctx, const vs_lane *lanes, size_t n_lanes, size_t n_samples, int16_t *pcm)
{
  if (!pcm) return VS_ERR_ARG;
  if (vs_is_pinned(pcm)) /* DMA straight into the caller's buffer */
    return vs_synth_rows_impl(ctx, lanes, n_lanes, n_samples, NULL, NULL, pcm);
  CopyOut c = {pcm, n_samples};
  return vs_synth_rows_impl(ctx, lanes, n_lanes, n_samples, copy_rows, &c, NULL);
}

/* ------------------------------------------------------------------------------------------
 * vs_source / vs_filter: what the two drop-in programs call (one utterance, or small batches)
 * ---------------------------------------------------------------------------------------- */
/* A handful of utterances -- what the two drop-in programs ask for: nothing is copied.  The plan's records, the flow
 * that comes in, the PCM, the cycle log and the counts all live in pinned, device-mapped host memory; the kernel
 * reads and writes them over PCIe (a few tens of KiB), the CPU moves rows between that block and the caller's
 * buffers.  A process that only ever does this never triggers the runtime's copy-path set-up (27 ms). */
#define VS_ZERO_COPY_LANES 64
#define VS_ZERO_COPY_BYTES ((size_t)2 << 20)
static int vs_run_host_zero_copy(vs_ctx *ctx, int kind, const vs_lane *lanes, size_t n_lanes,
                                 size_t n_samples, const int16_t *in_host, int16_t *out_host,
                                 vs_cycle_rec *recs, size_t recs_pitch, int32_t *ncyc)
{
  VsPool *P = &ctx->pool;
  const size_t pitch = (n_samples + 7) & ~(size_t)7; /* rows start 16-byte aligned */
  const size_t bytes = n_lanes * pitch * sizeof(int16_t);
  const bool want_log = recs && kind != VS_KIND_FILTER;
  const bool want_ncyc = ncyc && kind != VS_KIND_FILTER;
  const size_t log_bytes = want_log ? n_lanes * recs_pitch * sizeof(vs_cycle_rec) : 0;
  const size_t ncyc_bytes = want_ncyc ? n_lanes * sizeof(int32_t) : 0;
  const size_t off_in = (bytes + 63) & ~(size_t)63;
  const size_t off_log = off_in + ((kind == VS_KIND_FILTER) ? ((bytes + 63) & ~(size_t)63) : 0);
  const size_t off_ncyc = (off_log + log_bytes + 63) & ~(size_t)63;
  const size_t total = off_ncyc + ncyc_bytes + 64;
  VS_HIP(ctx, hipSetDevice(ctx->device));
  if (P->zc_io_bytes < total) {
    if (P->zc_io) VS_HIP(ctx, hipHostFree(P->zc_io));
    P->zc_io = NULL;
    P->zc_io_bytes = 0;
    const size_t want = (total + 65535) & ~(size_t)65535;
    VS_HIP(ctx, hipHostMalloc(&P->zc_io, want, hipHostMallocMapped));
    P->zc_io_bytes = want;
  }
  void *devp = NULL;
  VS_HIP(ctx, hipHostGetDevicePointer(&devp, P->zc_io, 0));
  char *hb = (char *)P->zc_io, *db = (char *)devp;
  vs_plan *plan = NULL;
  int rc = vs_plan_create_impl(ctx, lanes, n_lanes, n_samples,
                               (kind == VS_KIND_FILTER ? VS_PLAN_FILTER_ONLY : 0) | VS_PLAN_POOL_SCRATCH | VS_PLAN_ZERO_COPY, &plan);
  if (rc != VS_OK) return rc;
  if (kind == VS_KIND_FILTER)
    for (size_t l = 0; l < n_lanes; l++) memcpy(hb + off_in + l * pitch * 2, in_host + l * n_samples, n_samples * 2);
  if (want_log) memset(hb + off_log, 0, log_bytes);
  rc = vs_plan_launch(plan, kind, (kind == VS_KIND_FILTER) ? (const int16_t *)(db + off_in) : NULL, pitch, (int16_t *)db, pitch,
                      want_log ? (vs_cycle_rec *)(db + off_log) : NULL, recs_pitch, want_ncyc ? (int32_t *)(db + off_ncyc) : NULL);
  if (rc == VS_OK) rc = vs_plan_status(plan, NULL); /* waits for the stream: the device's writes are in host memory */
  if (rc == VS_OK) {
    for (size_t l = 0; l < n_lanes; l++) memcpy(out_host + l * n_samples, hb + l * pitch * 2, n_samples * 2);
    if (want_log) memcpy(recs, hb + off_log, log_bytes);
    if (want_ncyc) memcpy(ncyc, hb + off_ncyc, ncyc_bytes);
  }
  vs_plan_destroy(plan);
  return rc;
}

static int vs_run_host(vs_ctx *ctx, int kind, const vs_lane *lanes, size_t n_lanes,
                       size_t n_samples, const int16_t *in_host, int16_t *out_host,
                       vs_cycle_rec *recs, size_t recs_pitch, int32_t *ncyc)
{
  if (!ctx || !lanes || !out_host || n_lanes == 0 || n_samples == 0) return VS_ERR_ARG;
  if (n_lanes <= VS_ZERO_COPY_LANES && !ctx->copy_warm &&
      n_lanes * ((n_samples + 7) & ~(size_t)7) * 2 * (kind == VS_KIND_FILTER ? 2 : 1) +
              (recs ? n_lanes * recs_pitch * sizeof(vs_cycle_rec) : 0) <= VS_ZERO_COPY_BYTES)
    return vs_run_host_zero_copy(ctx, kind, lanes, n_lanes, n_samples, in_host, out_host, recs, recs_pitch, ncyc);
  VsPool *P = &ctx->pool;
  vs_plan *plan = NULL;
  int rc = vs_plan_create_impl(ctx, lanes, n_lanes, n_samples,
                               (kind == VS_KIND_FILTER ? VS_PLAN_FILTER_ONLY : 0) | VS_PLAN_POOL_SCRATCH, &plan);
  if (rc != VS_OK) return rc;
  const size_t pitch = (n_samples + 7) & ~(size_t)7; /* rows start 16-byte aligned */
  const size_t bytes = n_lanes * pitch * sizeof(int16_t);
  const bool want_log = recs && kind != VS_KIND_FILTER;
  const bool want_ncyc = ncyc && kind != VS_KIND_FILTER;
  const size_t log_bytes = want_log ? n_lanes * recs_pitch * sizeof(vs_cycle_rec) : 0;
  const size_t ncyc_bytes = want_ncyc ? n_lanes * sizeof(int32_t) : 0;
  rc = vs_pool_device(ctx, &P->d_out[0], &P->d_out_bytes[0], bytes);
  if (rc == VS_OK && kind == VS_KIND_FILTER) rc = vs_pool_device(ctx, &P->d_in, &P->d_in_bytes, bytes);
  if (rc == VS_OK && (log_bytes || ncyc_bytes)) rc = vs_pool_device(ctx, &P->d_aux, &P->d_aux_bytes, log_bytes + ncyc_bytes + 16);
  if (rc != VS_OK) {
    vs_plan_destroy(plan);
    return rc;
  }
  int16_t *d_out = (int16_t *)P->d_out[0];
  int16_t *d_in = (kind == VS_KIND_FILTER) ? (int16_t *)P->d_in : NULL;
  vs_cycle_rec *d_log = want_log ? (vs_cycle_rec *)P->d_aux : NULL;
  int32_t *d_ncyc = want_ncyc ? (int32_t *)((char *)P->d_aux + ((log_bytes + 15) & ~(size_t)15)) : NULL;
  hipError_t e = hipSuccess;
  if (d_in)
    e = hipMemcpy2DAsync(d_in, pitch * 2, in_host, n_samples * 2, n_samples * 2, n_lanes,
                         hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess && d_log) e = hipMemsetAsync(d_log, 0, log_bytes, ctx->stream);
  if (e == hipSuccess) {
    rc = vs_plan_launch(plan, kind, d_in, pitch, d_out, pitch, d_log, recs_pitch, d_ncyc);
    if (rc == VS_OK)
      e = hipMemcpy2DAsync(out_host, n_samples * 2, d_out, pitch * 2, n_samples * 2, n_lanes,
                           hipMemcpyDeviceToHost, ctx->stream);
    if (rc == VS_OK && e == hipSuccess && d_log)
      e = hipMemcpyAsync(recs, d_log, log_bytes, hipMemcpyDeviceToHost, ctx->stream);
    if (rc == VS_OK && e == hipSuccess && d_ncyc)
      e = hipMemcpyAsync(ncyc, d_ncyc, ncyc_bytes, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess && rc == VS_OK) rc = vs_plan_status(plan, NULL);
  }
  vs_plan_destroy(plan);
  if (e != hipSuccess) {
    ctx->last_hip_error = (int)e;
    return VS_ERR_HIP;
  }
  return rc;
}

int vs_source(vs_ctx *ctx, const vs_lane *lanes, size_t n_lanes, size_t n_samples,
              int16_t *flow, vs_cycle_rec *recs, size_t recs_pitch, int32_t *ncyc)
{
  if (recs && recs_pitch == 0) return VS_ERR_ARG;
  return vs_run_host(ctx, VS_KIND_SOURCE, lanes, n_lanes, n_samples, NULL, flow, recs, recs_pitch, ncyc);
}

int vs_filter(vs_ctx *ctx, const vs_lane *lanes, size_t n_lanes, size_t n_samples,
              const int16_t *flow, int16_t *pcm)
{
  if (!flow) return VS_ERR_ARG;
  return vs_run_host(ctx, VS_KIND_FILTER, lanes, n_lanes, n_samples, flow, pcm, NULL, 0, NULL);
}
