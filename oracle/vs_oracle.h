/*
 * oracle/vs_oracle.h -- CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call this.  The
 * product (voice_synth_amd/, include/voice_synth.h) never links, loads or executes it.
 *
 * Parity pin: tests/test_oracle_golden.py compares this restatement byte for byte with
 * fixtures produced by the reference itself, compiled from /root/reference by oracle/Makefile
 * into oracle/_ref/ and linked with oracle/rng_shim.c (script: oracle/gen_golden.py), and
 * with the 17 RNG-independent sha256 known answers of SURVEY.md section 4.
 */
#ifndef VS_ORACLE_H
#define VS_ORACLE_H

#include "../include/voice_synth.h"

#ifdef __cplusplus
extern "C" {
#endif

/* flowgen_shimmer.c:242-423 for one lane.  flow receives n_samples int16.  recs (optional)
 * receives one record per generated cycle, up to max_recs; *ncyc the number of cycles;
 * *ndraws the number of random() calls the reference would have made. */
int vs_oracle_source(const vs_lane *lane, size_t n_samples, int16_t *flow, vs_cycle_rec *recs,
                     size_t max_recs, int32_t *ncyc, uint64_t *ndraws);

/* vowel_new.c:222-224, 266-289, 413-427 for one lane (no vowel -n noise). */
int vs_oracle_filter(const vs_lane *lane, size_t n_samples, const int16_t *flow, int16_t *pcm);

/* The denominator coefficients the lane selects: 23 for a table, order + 1 (at most 41) for an
 * explicit set.  A must hold VS_MAX_NCOEF doubles. */
int vs_oracle_coefficients(const vs_lane *lane, double *A);
/* Order of the lane's filter (vowel_new.c:172): 22, or vs_lane.order of an explicit set; -1 beyond MAX_ORDER */
int vs_oracle_order(const vs_lane *lane);

/* Batches, OpenMP over lanes when threads > 1.  Layout [n_lanes][n_samples]. */
int vs_oracle_source_batch(const vs_lane *lanes, size_t n_lanes, size_t n_samples, int16_t *flow,
                           int threads);
int vs_oracle_filter_batch(const vs_lane *lanes, size_t n_lanes, size_t n_samples,
                           const int16_t *flow, int16_t *pcm, int threads);
int vs_oracle_synth_batch(const vs_lane *lanes, size_t n_lanes, size_t n_samples, int16_t *pcm,
                          int threads);

/* Philox known-answer access for tests. */
void vs_oracle_philox(const uint32_t *ctr, const uint32_t *key, uint32_t *out);
long vs_oracle_draw(uint64_t seed, uint64_t n);

/* round2int() of vowel_new.c:413-427 and truncate() of flowgen_shimmer.c:591-600. */
int16_t vs_oracle_round2int(double x);
int16_t vs_oracle_truncate(float x);

int vs_oracle_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
