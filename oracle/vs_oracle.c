/*
 * oracle/vs_oracle.c -- CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY
 * (see vs_oracle.h for who may call it and how it is pinned).
 *
 * Every arithmetic statement keeps the reference's operand types, evaluation order and
 * float/double conversions.  Build WITHOUT contraction or fast-math
 * (oracle/Makefile: -O2 -ffp-contract=off) on x86-64/SSE2 (FLT_EVAL_METHOD == 0), which is
 * what the reference's own build does.
 *
 * Deviations from the reference, all on inputs where the reference itself is undefined:
 *   - T4 (flowgen_shimmer.c:113) starts at 0 instead of an uninitialised stack value
 *     (SURVEY.md F9);
 *   - double -> short conversions wrap modulo 2^16 through an int32 (what gcc/x86-64 emits)
 *     instead of being undefined when out of range;
 *   - the work buffer is sized from the lane (not 2*fs/Fg, flowgen_shimmer.c:569), so lanes
 *     that would overflow the reference's x[] still compute what the loops say.
 */
#include "vs_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "philox.h"
#include "vs_oracle_tables.h"

#define VS_RAND_MAX 2147483647 /* glibc RAND_MAX, an int */
#define VS_PI (4.0 * atan(1.0)) /* flowgen_shimmer.c:39, textual macro: PI*i/T2 == ((4.0*atan(1.0))*i)/T2 */

static inline int16_t wrap16(double v) { return (int16_t)(int32_t)v; }

/* flowgen_shimmer.c:591-600 */
int16_t vs_oracle_truncate(float aux)
{
  int16_t i;
  if (aux > 32767) i = 32767;
  else if (aux < -32767) i = -32767;
  else i = wrap16(ceil(aux));
  return i;
}

/* vowel_new.c:413-427 */
int16_t vs_oracle_round2int(double x)
{
  double dec;
  dec = x - floor(x);
  if (dec > 0.5) {
    x = x + 1;
  }
  if (x > 32767) x = 32767;
  else if (x < -32767) x = -32767;
  return wrap16(floor(x));
}

void vs_oracle_philox(const uint32_t *ctr, const uint32_t *key, uint32_t *out)
{
  vs_philox4x32_10(ctr, key, out);
}

long vs_oracle_draw(uint64_t seed, uint64_t n)
{
  vs_draw_stream s;
  vs_draw_init(&s, seed);
  s.n = n;
  return vs_draw_next(&s);
}

/* Order of vowel_new.c:172 -- 22 for every table; an explicit set carries its own */
int vs_oracle_order(const vs_lane *lane)
{
  if (lane->vowel != VS_VOWEL_CUSTOM) return VS_ORDER;
  if (lane->order < 0 || lane->order > VS_MAX_ORDER) return -1;
  return lane->order ? lane->order : VS_ORDER;
}

int vs_oracle_coefficients(const vs_lane *lane, double *A)
{
  if (lane->vowel == VS_VOWEL_CUSTOM) {
    /* an explicit set: A[0..order], order = lane->order (0 = 22), at most MAX_ORDER (vowel_new.c:33) */
    const int order = vs_oracle_order(lane);
    if (order < 1) return VS_ERR_RANGE;
    memcpy(A, lane->A, sizeof(double) * (size_t)(order + 1));
    return VS_OK;
  }
  for (int t = 0; t < VS_ORACLE_TAB_NTABLES; t++) {
    if (vs_oracle_tab_ids[t] == (char)lane->vowel) {
      memcpy(A, vs_oracle_tab_A[t], sizeof(double) * VS_NCOEF);
      return VS_OK;
    }
  }
  return VS_ERR_RANGE;
}

/* ------------------------------------------------------------------------------------------
 * Source: flowgen_shimmer.c:242-423
 * ---------------------------------------------------------------------------------------- */
int vs_oracle_source(const vs_lane *lane, size_t n_samples, int16_t *flow, vs_cycle_rec *recs,
                     size_t max_recs, int32_t *ncyc, uint64_t *ndraws)
{
  const float par_jitter = lane->jitter, par_cq = lane->cq, par_K = lane->K, par_F0 = lane->F0,
              par_DC = lane->DC, par_noise = lane->noise, par_Kvar = lane->Kvar,
              par_Shimmer = lane->shimmer;
  const long par_fs = lane->fs;
  const int par_amp = lane->amp;
  const int arg_jitter = (lane->flags & VS_FLAG_JITTER) != 0;
  const int arg_Shimmer = (lane->flags & VS_FLAG_SHIMMER) != 0;
  const int arg_noise = (lane->flags & VS_FLAG_NOISE) != 0;

  int i, k, P, T, T2, T3, T4 = 0;
  int par_NoiseDistWidth;
  unsigned long nSamples = (unsigned long)n_samples, CountSamples;
  float aux, x_pow = 0.0f, w_pow = 0.0f, J, DeltaPer[2] = {0, 0};
  float S = 0.0f, DeltaShimmer[2] = {0, 0};
  vs_draw_stream rng;
  int32_t cycles = 0;

  vs_draw_init(&rng, lane->seed);
  CountSamples = 0L;
  P = T = (int)((float)par_fs / par_F0); /* fg:244 */
  if (P < 1) return VS_ERR_RANGE;

  /* x[] and w[]: the longest cycle is floor(1.2 P); the pulse may reach 2*T2 = P+2 (cq = 1) */
  size_t cap = (size_t)(2 * P + 16);
  int16_t *x = (int16_t *)malloc(cap * sizeof(int16_t));
  int *w = (int *)malloc(cap * sizeof(int));
  if (!x || !w) {
    free(x);
    free(w);
    return VS_ERR_NOMEM;
  }
  memset(x, 0, cap * sizeof(int16_t));

  do {
    /* jitter, fg:248-291 */
    if (arg_jitter && par_jitter != 0.0) {
      DeltaPer[1] = DeltaPer[0];
      do {
        J = (vs_draw_next(&rng) / (VS_RAND_MAX * 10000.0)) * 40000.0 * par_jitter -
            2.0 * par_jitter;
        DeltaPer[0] = DeltaPer[1] * (2.0 + J) / (2.0 - J) + 2.0 * P * J / (2.0 - J);
        T = (int16_t)(int32_t)ceil((float)P + DeltaPer[0]);
      } while ((float)T > (float)1.2 * P || (float)T < (float)0.8 * P);
    }

    /* shimmer, fg:293-313 */
    float Amplitude;
    if (arg_Shimmer && par_Shimmer != 0.0) {
      DeltaShimmer[1] = DeltaShimmer[0];
      do {
        float epsilon = ((float)vs_draw_next(&rng)) / VS_RAND_MAX;
        S = epsilon * 4.0 * par_Shimmer - 2.0 * par_Shimmer;
        DeltaShimmer[0] =
            DeltaShimmer[1] * (2.0 + S) / (2.0 - S) + 2.0 * par_amp * S / (2.0 - S);
        Amplitude = ((float)par_amp + DeltaShimmer[0]);
      } while ((Amplitude > (float)1.8 * par_amp) || (Amplitude < (float)0.2 * par_amp));
    } else {
      Amplitude = (float)par_amp;
    }

    /* glottal flow, fg:317-336 */
    T2 = ceil(0.5 * par_cq * P);
    for (i = 0; i < T2; i++) {
      x[i] = wrap16(ceil(Amplitude * 0.5 * (1.0 - cos(VS_PI * i / T2))));
      if (x[i] < par_DC) {
        x[i] = (int16_t)(int32_t)par_DC;
        T4 = i;
      }
    }
    float Knew =
        par_K * (1 + 2 * par_Kvar * (((1.0 * vs_draw_next(&rng)) / VS_RAND_MAX) - 0.5));
    for (i = T2; i < 2 * T2; i++) {
      x[i] = wrap16(ceil((float)Amplitude * (Knew * cos(VS_PI * (i - T2) / T2) - Knew + 1.0)));
      if (x[i] < par_DC) break;
    }
    T3 = i;
    for (i = T3; i < T; i++) {
      x[i] = (int16_t)(int32_t)par_DC;
    }

    /* closed-phase noise, fg:373-411 */
    if (arg_noise) {
      aux = 0.0;
      for (i = T4; i < T3; i++) {
        aux += (float)x[i] * x[i];
      }
      x_pow = aux / ((float)T3 - T4);

      aux = 1.0 + ((float)T3 - T4) / ((float)T);
      par_NoiseDistWidth = (int32_t)sqrt(12 * aux * (x_pow) / par_noise);

      aux = 0.0;
      for (i = 0; i < T4; i++) {
        w[i] = wrap16(ceil(((1.0 * vs_draw_next(&rng)) / VS_RAND_MAX) * par_NoiseDistWidth -
                           par_NoiseDistWidth / 2.0));
        aux += (float)w[i] * w[i];
        x[i] = vs_oracle_truncate((float)x[i] + w[i]);
      }
      for (i = T3; i < T; i++) {
        w[i] = wrap16(ceil(((1.0 * vs_draw_next(&rng)) / VS_RAND_MAX) * par_NoiseDistWidth -
                           par_NoiseDistWidth / 2.0));
        aux += (float)w[i] * w[i];
        x[i] = vs_oracle_truncate((float)x[i] + w[i]);
      }
      w_pow = aux / ((float)T);
    }

    if (recs && (size_t)cycles < max_recs) {
      recs[cycles].S = (arg_Shimmer && par_Shimmer != 0.0) ? S : 0.0f;
      recs[cycles].x_pow = arg_noise ? x_pow : 0.0f;
      recs[cycles].w_pow = arg_noise ? w_pow : 0.0f;
      recs[cycles].T = T;
    }
    cycles++;

    /* emit, fg:413-421 */
    unsigned long before = CountSamples;
    CountSamples += T;
    if (CountSamples > nSamples) k = T - (CountSamples - nSamples);
    else k = T;
    memcpy(flow + before, x, (size_t)k * sizeof(int16_t));
  } while (CountSamples < nSamples);

  free(x);
  free(w);
  if (ncyc) *ncyc = cycles;
  if (ndraws) *ndraws = rng.n;
  return VS_OK;
}

/* ------------------------------------------------------------------------------------------
 * Filter: vowel_new.c:222-224 (zero state), 266-289 (step), 413-427 (rounding)
 * ---------------------------------------------------------------------------------------- */
int vs_oracle_filter(const vs_lane *lane, size_t n_samples, const int16_t *flow, int16_t *pcm)
{
  double A[VS_MAX_NCOEF], B0 = 1.0;   /* MAX_ORDER + 1, vowel_new.c:33,61-63 */
  double y_double[VS_MAX_NCOEF];
  const float gain = lane->gain, pre_emphasis = lane->pre_emphasis;
  int rc = vs_oracle_coefficients(lane, A);
  if (rc != VS_OK) return rc;
  const int Order = vs_oracle_order(lane); /* vowel_new.c:172 */

  for (int j = 0; j < Order + 1; j++) y_double[j] = 0.0;

  /* frame length, vowel_new.c:361-363 (header.nSamplesPerSec is an unsigned long) */
  const unsigned long nSamplesPerSec = (unsigned long)lane->fs;
  const int milisec1 = (int)(nSamplesPerSec * 0.001 / 2.0) * 2;
  const size_t Lframe = (size_t)(50 * milisec1);
  const float snr = lane->out_snr;
  vs_draw_stream rng;
  vs_draw_init(&rng, lane->out_seed);
  if (snr > 0 && Lframe == 0) return VS_ERR_RANGE;

  size_t f0 = 0;
  while (f0 < n_samples) {
    const size_t ni = (snr > 0) ? ((n_samples - f0 < Lframe) ? n_samples - f0 : Lframe)
                                : n_samples - f0; /* frames only matter to the noise */
    int16_t *y = pcm + f0;
    for (size_t k = 0; k < ni; k++) {
      const size_t i = f0 + k;
      /* zeros: B = {1, 0, ...}; the j >= 1 terms add (+-0)*gain and cannot change y_double[0]
       * (vowel_new.c:266-269, 435-448) */
      y_double[0] = 0.0;
      y_double[0] = y_double[0] + B0 * flow[i] * gain;
      /* poles, vowel_new.c:279-281 */
      for (int j = 1; j < Order + 1; j++) {
        y_double[0] = y_double[0] - A[j] * y_double[j];
      }
      /* pre-emphasis on the output only, vowel_new.c:284 */
      y[k] = vs_oracle_round2int(y_double[0] - pre_emphasis * y_double[1]);
      /* shift, vowel_new.c:287-289 */
      for (int j = Order; j > 0; j--) {
        y_double[j] = y_double[j - 1];
      }
    }
    /* noise added to the filtered frame, vowel_new.c:302-324 */
    if (snr > 0) {
      float aux, sig_power, NoiseDistWidth, noiseval;
      aux = 0.0;
      for (size_t k = 0; k < ni; k++) {
        aux += (float)y[k] * y[k];
      }
      sig_power = aux / (float)ni;
      NoiseDistWidth = sqrt(12 * sig_power / snr);
      for (size_t k = 0; k < ni; k++) {
        noiseval = (1.0 * vs_draw_next(&rng)) / VS_RAND_MAX;
        aux = NoiseDistWidth * (noiseval - 0.5);
        y[k] = vs_oracle_round2int(1.0 * y[k] + 1.0 * aux);
      }
    }
    f0 += ni;
  }
  return VS_OK;
}

/* ------------------------------------------------------------------------------------------
 * Batches
 * ---------------------------------------------------------------------------------------- */
int vs_oracle_max_threads(void)
{
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

int vs_oracle_source_batch(const vs_lane *lanes, size_t n_lanes, size_t n_samples, int16_t *flow,
                           int threads)
{
  int err = VS_OK;
  if (threads < 1) threads = 1;
#pragma omp parallel for num_threads(threads) schedule(dynamic, 4)
  for (long l = 0; l < (long)n_lanes; l++) {
    int rc = vs_oracle_source(&lanes[l], n_samples, flow + (size_t)l * n_samples, NULL, 0, NULL,
                              NULL);
    if (rc != VS_OK) {
#pragma omp critical
      err = rc;
    }
  }
  return err;
}

int vs_oracle_filter_batch(const vs_lane *lanes, size_t n_lanes, size_t n_samples,
                           const int16_t *flow, int16_t *pcm, int threads)
{
  int err = VS_OK;
  if (threads < 1) threads = 1;
#pragma omp parallel for num_threads(threads) schedule(dynamic, 4)
  for (long l = 0; l < (long)n_lanes; l++) {
    int rc = vs_oracle_filter(&lanes[l], n_samples, flow + (size_t)l * n_samples,
                              pcm + (size_t)l * n_samples);
    if (rc != VS_OK) {
#pragma omp critical
      err = rc;
    }
  }
  return err;
}

int vs_oracle_synth_batch(const vs_lane *lanes, size_t n_lanes, size_t n_samples, int16_t *pcm,
                          int threads)
{
  int err = VS_OK;
  if (threads < 1) threads = 1;
#pragma omp parallel num_threads(threads)
  {
    int16_t *flow = (int16_t *)malloc(n_samples * sizeof(int16_t));
#pragma omp for schedule(dynamic, 4)
    for (long l = 0; l < (long)n_lanes; l++) {
      int rc = flow ? vs_oracle_source(&lanes[l], n_samples, flow, NULL, 0, NULL, NULL)
                    : VS_ERR_NOMEM;
      if (rc == VS_OK) rc = vs_oracle_filter(&lanes[l], n_samples, flow, pcm + (size_t)l * n_samples);
      if (rc != VS_OK) {
#pragma omp critical
        err = rc;
      }
    }
    free(flow);
  }
  return err;
}
