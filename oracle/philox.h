/*
 * oracle/philox.h -- Philox4x32-10 counter-based generator (CPU, test infrastructure).
 *
 * TEST INFRASTRUCTURE ONLY: nothing in the shipped product path may include this file.
 * (The device code carries its own statement of the same published algorithm.)
 *
 * Algorithm: Salmon, Moraes, Dror, Shaw, "Parallel random numbers: as easy as 1, 2, 3"
 * (SC'11), Philox-4x32 with 10 rounds.  Known-answer vectors (Random123 kat_vectors) are
 * checked in tests/test_philox.py.
 *
 * Draw contract: draw number n of a lane is
 *     word (n & 3) of philox4x32_10(counter = (n >> 2, 0, 0, 0), key = (seed_lo, seed_hi))
 * shifted right by one bit, so that it lies in [0, 2^31-1] like glibc random().  Every
 * call site of random() in the reference divides by RAND_MAX = 2^31-1
 * (/root/reference/flowgen_shimmer.c:283,298,325,387,398).
 */
#ifndef VS_ORACLE_PHILOX_H
#define VS_ORACLE_PHILOX_H

#include <stdint.h>

#define VS_PHILOX_M0 0xD2511F53u
#define VS_PHILOX_M1 0xCD9E8D57u
#define VS_PHILOX_W0 0x9E3779B9u
#define VS_PHILOX_W1 0xBB67AE85u

static inline void vs_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; r++) {
    uint64_t p0 = (uint64_t)VS_PHILOX_M0 * c0;
    uint64_t p1 = (uint64_t)VS_PHILOX_M1 * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += VS_PHILOX_W0;
    k1 += VS_PHILOX_W1;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* Sequential draw stream of one lane. */
typedef struct {
  uint32_t key[2];
  uint64_t n;        /* index of the next draw */
  uint64_t blk_idx;  /* counter of the cached block, ~0 = none */
  uint32_t blk[4];
} vs_draw_stream;

static inline void vs_draw_init(vs_draw_stream *s, uint64_t seed)
{
  s->key[0] = (uint32_t)seed;
  s->key[1] = (uint32_t)(seed >> 32);
  s->n = 0;
  s->blk_idx = ~(uint64_t)0;
}

static inline long vs_draw_next(vs_draw_stream *s)
{
  uint64_t b = s->n >> 2;
  if (b != s->blk_idx) {
    uint32_t ctr[4] = {(uint32_t)b, (uint32_t)(b >> 32), 0u, 0u};
    vs_philox4x32_10(ctr, s->key, s->blk);
    s->blk_idx = b;
  }
  long r = (long)(s->blk[s->n & 3] >> 1);
  s->n++;
  return r;
}

#endif
