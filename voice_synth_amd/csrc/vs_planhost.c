/*
 * vs_planhost.c -- the part of plan creation that needs no device: a lane's parameters expanded into the record the
 * kernels read (everything that is a pure function of the parameters, evaluated with the reference's operand
 * types), the cos rows, the ring policy, the order of the lanes.  Plain C without a HIP header in sight, so that it
 * runs under AddressSanitizer / UBSan on the CPU (tests/test_host_sanitizers.py); vs_api.c does the rest.
 *
 *   P  = (int)((float)fs/F0)                     flowgen_shimmer.c:244
 *   T2 = ceil(0.5*cq*P)                          flowgen_shimmer.c:317
 *   cos(PI*k/T2), k = 0..T2-1                    flowgen_shimmer.c:319, 328 (same arguments in
 *                                                both half-pulses; PI is 4.0*atan(1.0), fg:39)
 *   (float)1.2*P, (float)0.8*P                   flowgen_shimmer.c:290
 *   (float)1.8*amp, (float)0.2*amp               flowgen_shimmer.c:306
 * This translation unit is compiled with -ffp-contract=off.
 */
#include <math.h>
#include <pthread.h>
#include <signal.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "vs_planhost.h"

/* ------------------------------------------------------------------------------------------
 * Host expansion of one lane (no device needed; exported for the CPU-side tests)
 * ---------------------------------------------------------------------------------------- */
/* The multiplier the output-noise kernel divides a sample index by its frame length with: for d = Lframe >= 2 and
 * k = ceil(log2 d), m = floor(2^(31+k) / d) + 1 lies in (2^31, 2^32) and floor(i / d) == (i * m) >> (31 + k) for every
 * 0 <= i < 2^31 (the excess of m over 2^(31+k)/d is at most 1, so the product overshoots i/d by less than
 * i / 2^(31+k) < 1/d).  0 for frames the vowel stage cannot have (Lframe < 2). */
uint32_t vs_lframe_magic(int Lframe)
{
  if (Lframe < 2) return 0u;
  int k = 0;
  while (((uint64_t)1 << k) < (uint64_t)Lframe) k++;
  return (uint32_t)((((uint64_t)1 << (31 + k)) / (uint64_t)Lframe) + 1u);
}

/* A[0..40] of the lane's filter: the table or the explicit set, zeros behind its order */
int vs_lane_taps(const vs_lane *lane, double *A)
{
  for (int j = 0; j < VS_MAX_NCOEF; j++) A[j] = 0.0;
  if (lane->vowel != VS_VOWEL_CUSTOM) return vs_vowel_coefficients(lane->vowel, A);
  int order = 0;
  const int rc = vs_lane_order(lane, &order);
  if (rc != VS_OK) return rc;
  for (int j = 0; j <= order; j++) A[j] = lane->A[j];
  return VS_OK;
}

bool vs_lane_is_wide(const vs_lane *lane)
{
  return lane->vowel == VS_VOWEL_CUSTOM && lane->order > VS_ORDER;
}

int vs_expand_lane(const vs_lane *lane, int32_t row, VsDevLane *d)
{
  int rc = vs_lane_validate(lane);
  if (rc != VS_OK) return rc;
  memset(d, 0, sizeof(*d));
  /* the taps are a row of the plan's table: one of the ten tables, or (-1) a row the plan makes for this lane's own
   * set (vs_tap_table_build; vs_lane_validate has been through the set) */
  d->tap_row = (lane->vowel == VS_VOWEL_CUSTOM) ? -1 : vs_vowel_index(lane->vowel);
  if (lane->vowel != VS_VOWEL_CUSTOM && d->tap_row < 0) return VS_ERR_RANGE; /* cannot happen behind vs_lane_validate */
  d->gain = (double)lane->gain;
  d->pre = (double)lane->pre_emphasis;
  d->jitter = lane->jitter;
  d->shimmer = lane->shimmer;
  d->K = lane->K;
  d->Kvar = lane->Kvar;
  d->DC = lane->DC;
  d->noise = lane->noise;
  d->amp = lane->amp;
  const int P = (int)((float)lane->fs / lane->F0);
  d->P = P;
  d->T2 = (int)ceil(0.5 * lane->cq * P);
  d->t_hi = (float)1.2 * P;
  d->t_lo = (float)0.8 * P;
  d->a_hi = (float)1.8 * lane->amp;
  d->a_lo = (float)0.2 * lane->amp;
  d->dcs = (int32_t)(int16_t)(int32_t)lane->DC;
  uint32_t f = 0;
  if ((lane->flags & VS_FLAG_JITTER) && lane->jitter != 0.0) f |= VS_DF_JITTER;
  if ((lane->flags & VS_FLAG_SHIMMER) && lane->shimmer != 0.0) f |= VS_DF_SHIMMER;
  if (lane->flags & VS_FLAG_NOISE) f |= VS_DF_NOISE;
  /* longest admissible period: T is an integer with (float)T <= t_hi */
  d->tbound = (f & VS_DF_JITTER) ? (int)floorf(d->t_hi) : P;
  if (d->tbound < P) d->tbound = P;
  /* for an integer sample x: (float)x < par.DC  <=>  x < ceil(par.DC)   (fg:320, 329) */
  d->thr = (lane->DC < 2147483000.0f) ? (int32_t)ceilf(lane->DC) : 2147483647;
  /* VS_DF_FAST (vs_device.h): bounds under which the generator's short sequences equal the
   * general ones.  Largest admissible amplitude: (float)1.8*amp with shimmer (fg:306), amp
   * without; the rising flank reaches ceil(Amplitude), the falling one Amplitude*(1 - 2*Knew)
   * with Knew <= K*(1 + Kvar) (fg:325). */
  {
    const double amax = (f & VS_DF_SHIMMER) ? (double)d->a_hi : (double)lane->amp;
    const double kmax = (double)lane->K * (1.0 + (double)lane->Kvar) * 1.000001;
    const int tmin = (f & VS_DF_JITTER) ? (int)ceilf(d->t_lo) : P; /* rejection keeps (float)T >= t_lo */
    const bool in_short = amax <= 32767.0 && (2.0 * kmax - 1.0) * amax <= 32767.0 && lane->DC <= 32767.0f;
    if (in_short && d->T2 >= 4 && 2 * d->T2 + VS_TRASH_ROWS <= tmin) f |= VS_DF_FAST;
  }
  d->flags = f;
  d->key0 = (uint32_t)lane->seed;
  d->key1 = (uint32_t)(lane->seed >> 32);
  d->row = row;
  d->out_snr = lane->out_snr;
  {
    /* milisec1 = (int)(header.nSamplesPerSec * 0.001/2.0)*2; Lframe = 50*milisec1 (vowel_new.c:361-363) */
    const unsigned long nSamplesPerSec = (unsigned long)lane->fs;
    const int milisec1 = (int)(nSamplesPerSec * 0.001 / 2.0) * 2;
    d->Lframe = 50 * milisec1;
    d->lframe_magic = vs_lframe_magic(d->Lframe);
  }
  if (lane->out_snr > 0 && d->Lframe <= 0) return VS_ERR_UNSUPPORTED; /* fs < 2000: zero-length frames */
  d->okey0 = (uint32_t)lane->out_seed;
  d->okey1 = (uint32_t)(lane->out_seed >> 32);
  return VS_OK;
}

/* cos(PI*k/T2) with PI the reference's macro 4.0*atan(1.0) (flowgen_shimmer.c:39), which
 * expands textually: PI*i/T2 == ((4.0*atan(1.0))*i)/T2 */
void vs_cos_row(int T2, double *row)
{
  for (int k = 0; k < T2; k++) row[k] = cos(4.0 * atan(1.0) * k / T2);
}

/* Ring capacity in samples per lane and the super-step threshold that goes with it.
 *
 * A lane joins a generator round when its next cycle is certain to fit (fill + tbound <= slots)
 * and takes part in a filter super-step when it holds VS_SS samples, so VS_SS + max(tbound) is
 * the minimum.  More room lets lanes with long periods sit rounds out while the others catch
 * up, which keeps both the rounds and the super-steps well attended; the policy table below
 * comes from replaying real period sequences through the scheduler.
 * The default keeps four 64-lane workgroups resident per CU (160 KiB LDS / 4). */
int vs_ring_policy_for(int group_lanes, int tmax, int cap, int *slots, int *ready_min, int request, double depth)
{
  const int hard_limit = ((VS_LDS_LIMIT - 16 * 1024) / (group_lanes * 2) / VS_SS) * VS_SS; /* keeps 16 KiB for cos rows */
  /* one super-step + the longest cycle + the slots a trip may run past the cycle */
  const int need = ((VS_SS + tmax + VS_TRASH_ROWS + VS_SS - 1) / VS_SS) * VS_SS;
  if (need > hard_limit) return VS_ERR_UNSUPPORTED;
  if (cap <= 0) cap = 288; /* (288 + 8) rows * 128 B = 37 KiB + cos rows + sync words: four workgroups per CU */
  cap = (cap / VS_SS) * VS_SS;
  if (cap > hard_limit) cap = hard_limit;
  int want = ((VS_SS + (int)(depth * tmax) + VS_SS - 1) / VS_SS) * VS_SS;
  if (want < 192) want = 192;
  int c = want < cap ? want : cap;
  if (request > 0) c = cap; /* vs_tuning.ring_slots: the capacity itself, clamped to what fits */
  if (c < need) c = need;
  const double rho = (double)(c - VS_SS) / (double)tmax;
  int thr = 32;                 /* half the live lanes */
  if (rho >= 1.65) thr = 64;    /* all of them */
  else if (rho >= 1.45) thr = 58;
  else if (rho >= 1.33) thr = 48;
  *slots = c;
  if (ready_min) *ready_min = thr;
  return VS_OK;
}

/* a ring of 64 columns (one utterance per lane of a wavefront) */
int vs_ring_policy(int tmax, int cap, int *slots, int *ready_min)
{
  return vs_ring_policy_for(VS_WAVE, tmax, cap, slots, ready_min, 0, 1.7);
}

/* Ring capacity for periods up to tmax: the 64-column ring if it can take them, else the narrow one
 * (VS_NARROW_LANES columns, four times the slots in the same LDS). */
int vs_ring_slots_for(int tmax, int *slots)
{
  int rc = vs_ring_policy_for(VS_WAVE, tmax, 0, slots, NULL, 0, 1.7);
  if (rc == VS_ERR_UNSUPPORTED) rc = vs_ring_policy_for(VS_NARROW_LANES, tmax, 0, slots, NULL, 0, 1.7);
  return rc;
}

/* filter-only lane record: gain, pre-emphasis, coefficients, row, vowel -n fields; everything
 * of the source left zero.  Only the fields the reference's vowel reads are validated, so any
 * sample rate goes (vowel_new.c:196-205 checks the format tag and the bit depth only). */
int vs_expand_filter_lane(const vs_lane *lane, int32_t row, VsDevLane *d)
{
  if (!(lane->pre_emphasis >= 0.0 && lane->pre_emphasis <= 1.0)) return VS_ERR_RANGE; /* vw:127 */
  if (!(lane->gain >= 1)) return VS_ERR_RANGE;                                        /* vw:132 */
  if (lane->vowel == VS_VOWEL_CUSTOM) {
    int order = 0;
    const int rc = vs_lane_order(lane, &order);
    if (rc != VS_OK) return rc;
    for (int j = 0; j <= order; j++)
      if (!isfinite(lane->A[j])) return VS_ERR_RANGE;
    if (lane->A[0] != 1.0) return VS_ERR_RANGE;
  } else if (vs_vowel_index(lane->vowel) < 0) {
    return (lane->vowel == 'A' || lane->vowel == 'I' || lane->vowel == 'U') ? VS_ERR_UNSUPPORTED : VS_ERR_RANGE;
  }
  if (lane->fs <= 0) return VS_ERR_RANGE;
  memset(d, 0, sizeof(*d));
  d->tap_row = (lane->vowel == VS_VOWEL_CUSTOM) ? -1 : vs_vowel_index(lane->vowel);
  d->gain = (double)lane->gain;
  d->pre = (double)lane->pre_emphasis;
  d->row = row;
  d->out_snr = lane->out_snr;
  {
    const unsigned long nSamplesPerSec = (unsigned long)lane->fs;
    const int milisec1 = (int)(nSamplesPerSec * 0.001 / 2.0) * 2;
    d->Lframe = 50 * milisec1; /* vowel_new.c:361-363 */
    d->lframe_magic = vs_lframe_magic(d->Lframe);
  }
  if (lane->out_snr > 0 && d->Lframe <= 0) return VS_ERR_UNSUPPORTED;
  d->okey0 = (uint32_t)lane->out_seed;
  d->okey1 = (uint32_t)(lane->out_seed >> 32);
  return VS_OK;
}

/* expansion of the lane records: independent per lane, so large batches are cut over a few host
 * threads (65536 lanes: 38 ms on one core, the kernel itself takes under 3 ms) */
#define VS_EXPAND_THREADS 16
typedef struct ExpandJob {
  const vs_lane *lanes;
  VsDevLane *dl;
  const uint32_t *order; /* record i is made from lane order[i] (NULL: from lane i) */
  size_t lo, hi;
  int filter_only;
  int rc;          /* the failure of the LOWEST lane met in [lo, hi) */
  size_t bad;      /* ... and that lane */
  VsBatchStats st; /* of the records this job made */
  uint64_t *key;   /* NULL, or where the order key of record i goes */
  uint64_t key_or, key_and; /* over the keys this job wrote */
} ExpandJob;

static void stats_init(VsBatchStats *st)
{
  memset(st, 0, sizeof(*st));
  st->tmax = 1;
  st->pre1 = 1;
}
static void stats_lane(VsBatchStats *st, const vs_lane *lane, const VsDevLane *d)
{
  if (d->T2 > st->max_T2) st->max_T2 = d->T2;
  if (d->tbound > st->tmax) st->tmax = d->tbound;
  if (d->Lframe > 0 && (st->min_lframe == 0 || d->Lframe < st->min_lframe)) st->min_lframe = d->Lframe;
  if (d->Lframe > st->max_lframe) st->max_lframe = d->Lframe;
  if (d->Lframe <= 0) st->no_lframe = 1;
  if (d->out_snr > 0) st->any_onoise = 1;
  if (d->pre != 1.0) st->pre1 = 0;
  if (vs_lane_is_wide(lane)) st->wide = 1;
  if (d->tap_row < 0) st->n_custom++;
  if (d->flags & VS_DF_NOISE) st->n_noisy++;
}
static void stats_merge(VsBatchStats *a, const VsBatchStats *b)
{
  if (b->max_T2 > a->max_T2) a->max_T2 = b->max_T2;
  if (b->tmax > a->tmax) a->tmax = b->tmax;
  if (b->min_lframe > 0 && (a->min_lframe == 0 || b->min_lframe < a->min_lframe)) a->min_lframe = b->min_lframe;
  if (b->max_lframe > a->max_lframe) a->max_lframe = b->max_lframe;
  a->no_lframe |= b->no_lframe;
  a->any_onoise |= b->any_onoise;
  a->pre1 &= b->pre1;
  a->wide |= b->wide;
  a->n_noisy += b->n_noisy;
  a->n_custom += b->n_custom;
}

int vs_tap_table_build(const vs_lane *lanes, VsDevLane *dl, size_t n_lanes, size_t n_custom, double **taps_out, size_t *rows_out)
{
  if (!lanes || !dl || !taps_out || !rows_out) return VS_ERR_ARG;
  const size_t rows = VS_TAP_TABLE_ROWS + n_custom;
  double *taps = (double *)malloc(rows * VS_ORDER * sizeof(double));
  if (!taps) return VS_ERR_NOMEM;
  double A[VS_MAX_NCOEF];
  for (int t = 0; t < VS_TAP_TABLE_ROWS; t++) {
    if (vs_vowel_coefficients(vs_vowel_by_index(t), A) != VS_OK) {
      free(taps);
      return VS_ERR_INTERNAL;
    }
    for (int j = 1; j <= VS_ORDER; j++) taps[(size_t)t * VS_ORDER + (size_t)(j - 1)] = A[j];
  }
  size_t next = VS_TAP_TABLE_ROWS;
  for (size_t i = 0; i < n_lanes && n_custom; i++) {
    if (dl[i].tap_row >= 0) continue;
    if (next >= rows) { /* more sets than the count said: cannot happen */
      free(taps);
      return VS_ERR_INTERNAL;
    }
    const int rc = vs_lane_taps(&lanes[(size_t)dl[i].row], A); /* zeros behind the set's order: acc - 0*y == acc */
    if (rc != VS_OK) {
      free(taps);
      return rc;
    }
    /* (a wide set's first 22 taps: unused, the wide kernel reads all 40 from the plan's other table) */
    for (int j = 1; j <= VS_ORDER; j++) taps[next * VS_ORDER + (size_t)(j - 1)] = A[j];
    dl[i].tap_row = (int32_t)next;
    next++;
  }
  *taps_out = taps;
  *rows_out = rows;
  return VS_OK;
}

/* ------------------------------------------------------------------------------------------
 * The plan workspace: worker threads that live as long as the context, and the host buffers a plan is made in
 * ---------------------------------------------------------------------------------------- */
/* Sixteen threads started and joined per plan cost more than the records they make (0.38 ms of 0.6, 16-core quota
 * of an EPYC 9575F: tools/planhost_time.c), and a plan makes two parallel passes.  The team's threads sleep on a
 * condition variable between passes; a pass is an array of jobs that the threads -- the caller among them -- take one
 * by one. */
typedef void (*vs_team_fn)(void *job);
typedef struct VsTeam {
  int n_workers; /* threads besides the caller */
  pthread_t th[VS_EXPAND_THREADS - 1];
  pthread_mutex_t mu;
  pthread_cond_t go, done;
  unsigned gen; /* passes started so far */
  int pending;  /* workers that have not finished the current pass */
  int stop;
  vs_team_fn fn;
  char *jobs;
  size_t job_size;
  int n_jobs, next; /* next: first job nobody has taken (under mu) */
} VsTeam;

static void team_take_jobs(VsTeam *t)
{
  for (;;) {
    pthread_mutex_lock(&t->mu);
    const int i = (t->next < t->n_jobs) ? t->next++ : -1;
    pthread_mutex_unlock(&t->mu);
    if (i < 0) return;
    t->fn(t->jobs + (size_t)i * t->job_size);
  }
}
static void *team_worker(void *arg)
{
  VsTeam *t = (VsTeam *)arg;
  unsigned seen = 0;
  /* (a library's helper threads take no part in the application's signal handling: they are born with every signal
   * blocked, team_create) */
  for (;;) {
    pthread_mutex_lock(&t->mu);
    while (t->gen == seen && !t->stop) pthread_cond_wait(&t->go, &t->mu);
    if (t->stop) {
      pthread_mutex_unlock(&t->mu);
      return NULL;
    }
    seen = t->gen;
    pthread_mutex_unlock(&t->mu);
    team_take_jobs(t);
    pthread_mutex_lock(&t->mu);
    if (--t->pending == 0) pthread_cond_signal(&t->done);
    pthread_mutex_unlock(&t->mu);
  }
}
static VsTeam *team_create(void)
{
  long nt = sysconf(_SC_NPROCESSORS_ONLN);
  if (nt > VS_EXPAND_THREADS) nt = VS_EXPAND_THREADS;
  if (nt < 1) nt = 1;
  VsTeam *t = (VsTeam *)calloc(1, sizeof(VsTeam));
  if (!t) return NULL;
  pthread_mutex_init(&t->mu, NULL);
  pthread_cond_init(&t->go, NULL);
  pthread_cond_init(&t->done, NULL);
  /* a thread inherits the mask of the thread that makes it: every signal is blocked HERE, around pthread_create, and
   * the caller's own mask put back behind it -- a mask set by the worker itself leaves a window in which a signal of the
   * application's can still be delivered to it */
  sigset_t all, mine;
  sigfillset(&all);
  const int masked = pthread_sigmask(SIG_BLOCK, &all, &mine) == 0;
  for (long k = 0; k + 1 < nt; k++) { /* a thread that cannot be started is simply not there */
    if (pthread_create(&t->th[t->n_workers], NULL, team_worker, t) != 0) break;
    t->n_workers++;
  }
  if (masked) (void)pthread_sigmask(SIG_SETMASK, &mine, NULL);
  return t;
}
static void team_destroy(VsTeam *t)
{
  if (!t) return;
  pthread_mutex_lock(&t->mu);
  t->stop = 1;
  pthread_cond_broadcast(&t->go);
  pthread_mutex_unlock(&t->mu);
  for (int k = 0; k < t->n_workers; k++) pthread_join(t->th[k], NULL);
  pthread_cond_destroy(&t->go);
  pthread_cond_destroy(&t->done);
  pthread_mutex_destroy(&t->mu);
  free(t);
}
/* one pass: fn over jobs[0 .. n_jobs); returns when all of them are done.  No team (NULL), or a team without
 * workers: the caller does them all. */
static void team_run(VsTeam *t, vs_team_fn fn, void *jobs, size_t job_size, int n_jobs)
{
  if (!t || t->n_workers == 0 || n_jobs < 2) {
    for (int i = 0; i < n_jobs; i++) fn((char *)jobs + (size_t)i * job_size);
    return;
  }
  pthread_mutex_lock(&t->mu);
  t->fn = fn;
  t->jobs = (char *)jobs;
  t->job_size = job_size;
  t->n_jobs = n_jobs;
  t->next = 0;
  t->pending = t->n_workers;
  t->gen++;
  pthread_cond_broadcast(&t->go);
  pthread_mutex_unlock(&t->mu);
  team_take_jobs(t);
  pthread_mutex_lock(&t->mu);
  while (t->pending > 0) pthread_cond_wait(&t->done, &t->mu);
  pthread_mutex_unlock(&t->mu);
}

struct VsPlanWs {
  VsTeam *team;      /* made with the first batch of VS_TEAM_MIN_LANES lanes or more */
  VsDevLane *rec[2]; /* the records in input order / in kernel order */
  size_t rec_lanes;  /* lanes either holds */
  int rec_big;       /* the record buffers are asked of big_alloc first (else malloc) */
  int rec_from_big[2]; /* ... and where each one came from in the end: under page-locked-memory pressure either may be ordinary memory */
  vs_planws_alloc_fn big_alloc; /* where record buffers of VS_TEAM_MIN_LANES lanes and more come from (NULL: malloc) */
  vs_planws_free_fn big_free;
  void *big_user;
  uint64_t *key;     /* the order key of every lane */
  uint32_t *idx[2];  /* radix sort of the lane indices */
  size_t key_lanes, idx_lanes;
  size_t *count;     /* 65537 digit counters */
};
#define VS_TEAM_MIN_LANES 8192
#define VS_JOB_LANES 2048 /* lanes per job: 32 jobs for 65536 lanes -- a thread that comes late takes fewer of them */
#define VS_MAX_JOBS 64

VsPlanWs *vs_planws_create(void) { return (VsPlanWs *)calloc(1, sizeof(VsPlanWs)); }
void vs_planws_set_big_allocator(VsPlanWs *ws, vs_planws_alloc_fn alloc, vs_planws_free_fn release, void *user)
{
  if (!ws || ws->rec[0] || ws->rec[1]) return; /* only before the first batch */
  ws->big_alloc = (alloc && release) ? alloc : NULL;
  ws->big_free = (alloc && release) ? release : NULL;
  ws->big_user = user;
}
static void rec_release(VsPlanWs *ws)
{
  for (int b = 0; b < 2; b++) {
    if (ws->rec[b]) {
      if (ws->rec_from_big[b]) ws->big_free(ws->big_user, ws->rec[b]);
      else free(ws->rec[b]);
    }
    ws->rec[b] = NULL;
    ws->rec_from_big[b] = 0;
  }
  ws->rec_lanes = 0;
}
/* buffer b of the workspace: from the big allocator when the batch is big -- page-locked memory, so that the records' way
 * to the device is a DMA transfer that runs NEXT TO a kernel instead of a copy kernel that waits for the chip to be free --
 * and from malloc when there is none to be had (the upload is then staged by the runtime): either buffer, independently */
static VsDevLane *rec_alloc(VsPlanWs *ws, int b, size_t lanes)
{
  VsDevLane *p = NULL;
  ws->rec_from_big[b] = 0;
  if (ws->rec_big) {
    p = (VsDevLane *)ws->big_alloc(ws->big_user, lanes * sizeof(VsDevLane));
    if (p) ws->rec_from_big[b] = 1;
  }
  if (!p) p = (VsDevLane *)malloc(lanes * sizeof(VsDevLane));
  return p;
}
void vs_planws_destroy(VsPlanWs *ws)
{
  if (!ws) return;
  team_destroy(ws->team);
  rec_release(ws);
  free(ws->key);
  free(ws->idx[0]);
  free(ws->idx[1]);
  free(ws->count);
  free(ws);
}
/* the buffers for n lanes (grown, never shrunk: a fresh 8 MB block per plan is two thousand page faults) */
static int ws_reserve(VsPlanWs *ws, size_t n, int need_order)
{
  if (ws->rec_lanes < n) {
    rec_release(ws);
    ws->rec_big = (ws->big_alloc != NULL) && n >= VS_TEAM_MIN_LANES; /* big batches: from the context's allocator */
    ws->rec[0] = rec_alloc(ws, 0, n);
    if (!ws->rec[0]) return VS_ERR_NOMEM;
    ws->rec_lanes = n;
  }
  if (need_order) {
    if (!ws->rec[1]) {
      ws->rec[1] = rec_alloc(ws, 1, ws->rec_lanes);
      if (!ws->rec[1]) return VS_ERR_NOMEM;
    }
    if (ws->idx_lanes < n) {
      free(ws->idx[0]);
      free(ws->idx[1]);
      ws->idx_lanes = 0;
      ws->idx[0] = (uint32_t *)malloc(n * sizeof(uint32_t));
      ws->idx[1] = (uint32_t *)malloc(n * sizeof(uint32_t));
      if (!ws->idx[0] || !ws->idx[1]) return VS_ERR_NOMEM;
      ws->idx_lanes = n;
    }
    if (!ws->count) ws->count = (size_t *)malloc(65537 * sizeof(size_t));
    if (!ws->count) return VS_ERR_NOMEM;
  }
  if (ws->key_lanes < n) {
    free(ws->key);
    ws->key_lanes = 0;
    ws->key = (uint64_t *)malloc(n * sizeof(uint64_t));
    if (!ws->key) return VS_ERR_NOMEM;
    ws->key_lanes = n;
  }
  return VS_OK;
}

/* the kernels' order key of a source record: (P, T2, jitter / shimmer / noise on) -- lane_order_key() below says the
 * same of the lane it was made from */
static uint64_t record_order_key(const VsDevLane *d)
{
  return ((uint64_t)(uint32_t)d->P << 36) | ((uint64_t)(uint32_t)d->T2 << 8) |
         (uint64_t)(d->flags & (VS_DF_JITTER | VS_DF_SHIMMER | VS_DF_NOISE));
}

static void expand_range(void *arg)
{
  ExpandJob *j = (ExpandJob *)arg;
  j->rc = VS_OK;
  j->bad = (size_t)-1;
  j->key_or = 0;
  j->key_and = ~(uint64_t)0;
  stats_init(&j->st);
  for (size_t i = j->lo; i < j->hi; i++) {
    const size_t l = j->order ? (size_t)j->order[i] : i;
    const int rc = j->filter_only ? vs_expand_filter_lane(&j->lanes[l], (int32_t)l, &j->dl[i])
                                  : vs_expand_lane(&j->lanes[l], (int32_t)l, &j->dl[i]);
    if (rc != VS_OK) {
      if (l < j->bad) {
        j->rc = rc;
        j->bad = l;
      }
      if (!j->order) break; /* in input order nothing behind it can be lower */
      continue;
    }
    stats_lane(&j->st, &j->lanes[l], &j->dl[i]);
    if (j->key) {
      const uint64_t k = record_order_key(&j->dl[i]);
      j->key[i] = k;
      j->key_or |= k;
      j->key_and &= k;
    }
  }
}

/* records [lo, hi) of the kernel order: out[i] = in[order[i]] */
typedef struct GatherJob {
  const VsDevLane *in;
  VsDevLane *out;
  const uint32_t *order;
  size_t lo, hi;
} GatherJob;
static void gather_range(void *arg)
{
  GatherJob *j = (GatherJob *)arg;
  for (size_t i = j->lo; i < j->hi; i++) j->out[i] = j->in[j->order[i]];
}

static int jobs_for(size_t n_lanes, int parallel)
{
  if (!parallel) return 1;
  size_t n = (n_lanes + VS_JOB_LANES - 1) / VS_JOB_LANES;
  if (n > VS_MAX_JOBS) n = VS_MAX_JOBS;
  return n ? (int)n : 1;
}

/* one parallel pass of expand_range over all lanes; key: NULL or where every record's order key goes */
static int expand_pass(VsTeam *team, const vs_lane *lanes, VsDevLane *dl, size_t n_lanes, int filter_only,
                       const uint32_t *order, uint64_t *key, uint64_t *varying, VsBatchStats *stats)
{
  ExpandJob jobs[VS_MAX_JOBS];
  const int n_jobs = jobs_for(n_lanes, team != NULL);
  const size_t per = (n_lanes + (size_t)n_jobs - 1) / (size_t)n_jobs;
  int made = 0;
  for (int t = 0; t < n_jobs; t++) {
    const size_t lo = (size_t)t * per, hi = (lo + per < n_lanes) ? lo + per : n_lanes;
    if (lo >= hi) break;
    ExpandJob *j = &jobs[made++];
    j->lanes = lanes;
    j->dl = dl;
    j->order = order;
    j->key = key;
    j->lo = lo;
    j->hi = hi;
    j->filter_only = filter_only;
  }
  team_run(team, expand_range, jobs, sizeof(ExpandJob), made);
  /* the failure of the lowest lane: the same answer whatever the thread count and the order */
  int rc = VS_OK;
  size_t bad = (size_t)-1;
  uint64_t k_or = 0, k_and = ~(uint64_t)0;
  VsBatchStats all;
  stats_init(&all);
  for (int t = 0; t < made; t++) {
    if (jobs[t].rc != VS_OK && jobs[t].bad < bad) {
      rc = jobs[t].rc;
      bad = jobs[t].bad;
    }
    stats_merge(&all, &jobs[t].st);
    k_or |= jobs[t].key_or;
    k_and &= jobs[t].key_and;
  }
  if (stats) *stats = all;
  if (varying) *varying = made ? (k_or ^ k_and) : 0; /* bits of the key that differ somewhere */
  return rc;
}

/* stable LSD radix sort of the lane indices by key, 16 bits per pass, passes whose digit is the same in every key
 * skipped; returns the array that holds the order (a or b) */
static uint32_t *radix_order(const uint64_t *key, uint64_t varying, size_t n_lanes, uint32_t *a, uint32_t *b, size_t *count)
{
  for (size_t l = 0; l < n_lanes; l++) a[l] = (uint32_t)l;
  for (int shift = 0; shift < 64; shift += 16) {
    if (!((varying >> shift) & 0xFFFFu)) continue;
    memset(count, 0, 65537 * sizeof(size_t));
    for (size_t i = 0; i < n_lanes; i++) count[((key[a[i]] >> shift) & 0xFFFFu) + 1]++;
    for (size_t d = 0; d < 65536; d++) count[d + 1] += count[d];
    for (size_t i = 0; i < n_lanes; i++) b[count[(key[a[i]] >> shift) & 0xFFFFu]++] = a[i];
    uint32_t *sw = a;
    a = b;
    b = sw;
  }
  return a;
}

/* The records of a batch in the order the kernels want (source kinds; filter-only: input order), in the workspace's
 * memory: ONE parallel pass over the lanes makes every record where its lane stands and notes its order key -- the
 * 27 MB of a 65536-lane batch are read once, by all threads --, and only a batch whose keys differ is sorted (lane
 * indices, serially: two 16-bit passes for an F0 sweep) and gathered into the second buffer (8 MB, in parallel, out of
 * the caches).  *dl_out stays valid until the next call on this workspace. */
int vs_expand_all_ordered_ws(VsPlanWs *ws, const vs_lane *lanes, size_t n_lanes, int filter_only, VsDevLane **dl_out,
                             int *reordered, VsBatchStats *stats)
{
  if (!ws || !lanes || !dl_out || n_lanes == 0) return VS_ERR_ARG;
  if (reordered) *reordered = 0;
  const int may_order = !filter_only && n_lanes >= 2 && n_lanes <= 0xFFFFFFFFu;
  int rc = ws_reserve(ws, n_lanes, 0);
  if (rc != VS_OK) return rc;
  if (n_lanes >= VS_TEAM_MIN_LANES && !ws->team) ws->team = team_create(); /* NULL: this thread alone */
  VsTeam *team = (n_lanes >= VS_TEAM_MIN_LANES) ? ws->team : NULL;
  uint64_t varying = 0;
  rc = expand_pass(team, lanes, ws->rec[0], n_lanes, filter_only, NULL, may_order ? ws->key : NULL, &varying, stats);
  if (rc != VS_OK) return rc;
  *dl_out = ws->rec[0];
  if (!may_order || !varying) return VS_OK;
  rc = ws_reserve(ws, n_lanes, 1);
  if (rc != VS_OK) return rc;
  const uint32_t *order = radix_order(ws->key, varying, n_lanes, ws->idx[0], ws->idx[1], ws->count);
  GatherJob jobs[VS_MAX_JOBS];
  const int n_jobs = jobs_for(n_lanes, team != NULL);
  const size_t per = (n_lanes + (size_t)n_jobs - 1) / (size_t)n_jobs;
  int made = 0;
  for (int t = 0; t < n_jobs; t++) {
    const size_t lo = (size_t)t * per, hi = (lo + per < n_lanes) ? lo + per : n_lanes;
    if (lo >= hi) break;
    GatherJob *j = &jobs[made++];
    j->in = ws->rec[0];
    j->out = ws->rec[1];
    j->order = order;
    j->lo = lo;
    j->hi = hi;
  }
  team_run(team, gather_range, jobs, sizeof(GatherJob), made);
  *dl_out = ws->rec[1];
  if (reordered) *reordered = 1;
  return VS_OK;
}

/* the same into the caller's memory, with a workspace of its own (tests, small tools) */
static int expand_into(const vs_lane *lanes, VsDevLane *dl, size_t n_lanes, int filter_only, int ordered, int *reordered,
                       VsBatchStats *stats)
{
  if (!lanes || !dl) return VS_ERR_ARG;
  if (reordered) *reordered = 0;
  if (n_lanes == 0) {
    if (stats) stats_init(stats);
    return VS_OK;
  }
  if (!ordered) { /* input order: straight into the caller's memory */
    VsTeam *team = (n_lanes >= VS_TEAM_MIN_LANES) ? team_create() : NULL;
    const int rc = expand_pass(team, lanes, dl, n_lanes, filter_only, NULL, NULL, NULL, stats);
    team_destroy(team);
    return rc;
  }
  VsPlanWs *ws = vs_planws_create();
  if (!ws) return VS_ERR_NOMEM;
  VsDevLane *out = NULL;
  const int rc = vs_expand_all_ordered_ws(ws, lanes, n_lanes, filter_only, &out, reordered, stats);
  if (rc == VS_OK) memcpy(dl, out, n_lanes * sizeof(VsDevLane));
  vs_planws_destroy(ws);
  return rc;
}

int vs_expand_all(const vs_lane *lanes, VsDevLane *dl, size_t n_lanes, int filter_only)
{
  return expand_into(lanes, dl, n_lanes, filter_only, 0, NULL, NULL);
}

int vs_expand_all_stats(const vs_lane *lanes, VsDevLane *dl, size_t n_lanes, int filter_only, VsBatchStats *stats)
{
  return expand_into(lanes, dl, n_lanes, filter_only, 0, NULL, stats);
}

/* The order the kernels want -- wavefronts are formed from lanes with similar periods: by (P, T2, jitter / shimmer /
 * noise on or off), stable -- found BEFORE the records are made, from a 64-bit key per lane that restates the three
 * fields (flowgen_shimmer.c:244, :317 and the option flags) -- what record_order_key() reads off a finished record.
 * vs_kernel_order() answers from the lanes alone (tests hold the plan's order against it); the plan itself takes the
 * keys from the records it has just made (vs_expand_all_ordered_ws).  LSD radix sort of the lane indices, 16 bits per
 * pass, passes whose digit is the same in every key skipped (a homogeneous batch: none run, and *order_out stays
 * NULL = input order). */
static uint64_t lane_order_key(const vs_lane *lane)
{
  uint64_t P = 0, T2 = 0;
  if (lane->fs > 0 && lane->F0 > 0.0f) {
    const float q = (float)lane->fs / lane->F0; /* P = (int)((float)fs/F0), flowgen_shimmer.c:244 */
    if (q >= 0.0f && q < 67108864.0f) P = (uint64_t)(int)q;
  }
  if (lane->cq >= 0.0f && lane->cq <= 1.0f) {
    const double t2 = ceil(0.5 * lane->cq * (double)P); /* T2, flowgen_shimmer.c:317 */
    if (t2 >= 0.0 && t2 < 67108864.0) T2 = (uint64_t)t2;
  }
  uint64_t f = 0;
  if ((lane->flags & VS_FLAG_JITTER) && lane->jitter != 0.0) f |= 1;
  if ((lane->flags & VS_FLAG_SHIMMER) && lane->shimmer != 0.0) f |= 2;
  if (lane->flags & VS_FLAG_NOISE) f |= 4;
  return (P << 36) | (T2 << 8) | f;
}

int vs_kernel_order(const vs_lane *lanes, size_t n_lanes, uint32_t **order_out)
{
  *order_out = NULL;
  if (n_lanes < 2 || n_lanes > 0xFFFFFFFFu) return VS_OK;
  uint64_t *key = (uint64_t *)malloc(n_lanes * sizeof(uint64_t));
  if (!key) return VS_ERR_NOMEM;
  uint64_t all_or = 0, all_and = ~(uint64_t)0;
  for (size_t l = 0; l < n_lanes; l++) {
    key[l] = lane_order_key(&lanes[l]);
    all_or |= key[l];
    all_and &= key[l];
  }
  const uint64_t varying = all_or ^ all_and; /* bits that differ somewhere */
  if (!varying) {
    free(key);
    return VS_OK;
  }
  uint32_t *a = (uint32_t *)malloc(n_lanes * sizeof(uint32_t)), *b = (uint32_t *)malloc(n_lanes * sizeof(uint32_t));
  size_t *count = (size_t *)malloc(65537 * sizeof(size_t));
  if (!a || !b || !count) {
    free(key);
    free(a);
    free(b);
    free(count);
    return VS_ERR_NOMEM;
  }
  for (size_t l = 0; l < n_lanes; l++) a[l] = (uint32_t)l;
  for (int shift = 0; shift < 64; shift += 16) {
    if (!((varying >> shift) & 0xFFFFu)) continue;
    memset(count, 0, 65537 * sizeof(size_t));
    for (size_t i = 0; i < n_lanes; i++) count[((key[a[i]] >> shift) & 0xFFFFu) + 1]++;
    for (size_t d = 0; d < 65536; d++) count[d + 1] += count[d];
    for (size_t i = 0; i < n_lanes; i++) b[count[(key[a[i]] >> shift) & 0xFFFFu]++] = a[i];
    uint32_t *sw = a;
    a = b;
    b = sw;
  }
  free(key);
  free(b);
  free(count);
  *order_out = a;
  return VS_OK;
}

int vs_expand_all_ordered(const vs_lane *lanes, VsDevLane *dl, size_t n_lanes, int *reordered, VsBatchStats *stats)
{
  return expand_into(lanes, dl, n_lanes, 0, 1, reordered, stats);
}

/* mixed rings: group indices by longest period, descending (ties: lower index first).  A merge sort on the index
 * array -- the groups arrive nearly sorted, but "nearly" is not a bound (jitter on half of the lanes interleaves two
 * period scales), and a batch of millions of lanes has tens of thousands of groups */
static void vs_sort_groups_by_period(uint32_t *order, const int *tb, size_t n)
{
  uint32_t *tmp = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
  if (!tmp) { /* no memory for the scratch array: an insertion sort gives the same order, slowly */
    for (size_t i = 1; i < n; i++) {
      const uint32_t v = order[i];
      size_t k = i;
      while (k > 0 && (tb[order[k - 1]] < tb[v] || (tb[order[k - 1]] == tb[v] && order[k - 1] > v))) {
        order[k] = order[k - 1];
        k--;
      }
      order[k] = v;
    }
    return;
  }
  uint32_t *src = order, *dst = tmp;
  for (size_t w = 1; w < n; w *= 2) {
    for (size_t lo = 0; lo < n; lo += 2 * w) {
      const size_t mid = (lo + w < n) ? lo + w : n, hi = (lo + 2 * w < n) ? lo + 2 * w : n;
      size_t a = lo, b = mid, o = lo;
      while (a < mid && b < hi) {
        const bool b_first = tb[src[b]] > tb[src[a]] || (tb[src[b]] == tb[src[a]] && src[b] < src[a]);
        dst[o++] = b_first ? src[b++] : src[a++];
      }
      while (a < mid) dst[o++] = src[a++];
      while (b < hi) dst[o++] = src[b++];
    }
    uint32_t *sw = src;
    src = dst;
    dst = sw;
  }
  if (src != order) memcpy(order, src, n * sizeof(uint32_t));
  free(tmp);
}

int vs_mixed_rings_build(const VsDevLane *dl, size_t n_lanes, int floor_slots, VsGroupSlot **gmap_out, size_t *n_wg_out,
                         size_t *max_lds, int *c_min_out, int *c_max_out)
{
  if (!dl || !gmap_out || !n_wg_out || !max_lds || !c_min_out || !c_max_out || n_lanes == 0) return VS_ERR_ARG;
  const size_t G = VS_WAVE;
  const size_t n_groups = (n_lanes + G - 1) / G, n_wg = (n_groups + 3) / 4;
  int *tb_g = (int *)malloc(n_groups * sizeof(int));
  int *ltab_g = (int *)malloc(n_groups * sizeof(int)); /* every group reserves what ITS cos rows take */
  uint32_t *order = (uint32_t *)malloc(n_groups * sizeof(uint32_t));
  VsGroupSlot *gmap = (VsGroupSlot *)calloc(n_wg * 4, sizeof(VsGroupSlot));
  size_t *used = (size_t *)calloc(n_wg, sizeof(size_t));
  int rc = (tb_g && ltab_g && order && gmap && used) ? VS_OK : VS_ERR_NOMEM;
  int c_min = 0, c_max = 0;
  size_t lds = 0;
  if (rc == VS_OK) {
    for (size_t g = 0; g < n_groups; g++) {
      int tb = 1, seen[VS_WAVE], nseen = 0, sum = 0;
      for (size_t l = g * G; l < n_lanes && l < (g + 1) * G; l++) {
        if ((int)dl[l].tbound > tb) tb = (int)dl[l].tbound;
        bool dup = false;
        for (int k = 0; k < nseen; k++) dup = dup || (seen[k] == dl[l].T2);
        if (!dup) {
          seen[nseen++] = dl[l].T2;
          sum += (dl[l].T2 + 7) & ~7; /* rows are padded to a multiple of 8 doubles (vs_stage_cos_rows) */
        }
      }
      tb_g[g] = tb;
      ltab_g[g] = sum;
      order[g] = (uint32_t)g;
    }
    /* groups by longest period, descending; ties by index, so that the table does not depend on the sort */
    vs_sort_groups_by_period(order, tb_g, n_groups);
    for (size_t i = 0; i < n_wg * 4; i++) gmap[i].group = -1;
    for (size_t i = 0; i < n_groups && rc == VS_OK; i++) {
      const size_t pass = i / n_wg, pos = i % n_wg;
      const size_t wg = (pass & 1) ? (n_wg - 1 - pos) : pos; /* snake: the longest periods meet the shortest */
      const int tb = tb_g[order[i]];
      int c = ((VS_SS + (int)(1.7 * tb) + VS_SS - 1) / VS_SS) * VS_SS;
      const int need = ((VS_SS + tb + VS_TRASH_ROWS + VS_SS - 1) / VS_SS) * VS_SS;
      if (c < floor_slots) c = floor_slots;
      if (c < need) c = need;
      if ((double)(c - VS_SS) / (double)tb < 1.65) c += VS_SS; /* "deep": the filter may wait for all of its lanes */
      const size_t fixed = (size_t)ltab_g[order[i]] * sizeof(double) + VS_SYNC_WORDS_3 * VS_WAVE * sizeof(int);
      const size_t bytes = (((size_t)(c + VS_TRASH_ROWS) * G * sizeof(int16_t) + fixed) + 15) & ~(size_t)15;
      VsGroupSlot *gs = &gmap[wg * 4 + pass];
      gs->group = (int32_t)order[i];
      gs->ring_slots = c;
      gs->ltab_entries = ltab_g[order[i]];
      gs->lds_off = (int32_t)used[wg];
      used[wg] += bytes;
      if (used[wg] > VS_LDS_LIMIT) rc = VS_ERR_UNSUPPORTED;
      if (c_min == 0 || c < c_min) c_min = c;
      if (c > c_max) c_max = c;
    }
    for (size_t w = 0; w < n_wg; w++)
      if (used[w] > lds) lds = used[w];
  }
  free(tb_g);
  free(ltab_g);
  free(order);
  free(used);
  if (rc != VS_OK) {
    free(gmap);
    return rc;
  }
  *gmap_out = gmap;
  *n_wg_out = n_wg;
  *max_lds = lds;
  *c_min_out = c_min;
  *c_max_out = c_max;
  return VS_OK;
}
