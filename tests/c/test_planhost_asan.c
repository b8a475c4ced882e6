/* The device-free half of plan creation (voice_synth_amd/csrc/vs_planhost.c) and the gather bookkeeping of
 * vs_host.c under AddressSanitizer + UBSan + the thread pool of the expansion.  Built and run by
 * tests/test_host_sanitizers.py:
 *   gcc -fsanitize=address,undefined -fno-sanitize-recover=all -ffp-contract=off tests/c/test_planhost_asan.c \
 *       voice_synth_amd/csrc/vs_planhost.c voice_synth_amd/csrc/vs_host.c -Iinclude -lm -lpthread
 * What it goes through: the expansion of 20000 lanes of mixed periods on 8 threads, the failure of the LOWEST bad
 * lane whatever the thread that met it, the stable order by (P, T2, flags) -- a permutation, sorted, ties in input
 * order --, the mixed-rings table of a batch of many periods (every group once, every ring deep, every workgroup inside
 * the LDS), the ring policy over every period it can be asked for, cos rows, the rounds of a node's gather. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../voice_synth_amd/csrc/vs_planhost.h"

int vs_shard_cut(size_t n_lanes, int n_shards, int shard, size_t *lo, size_t *hi);
size_t vs_gather_rounds(size_t n_lanes, int n_shards, size_t chunk);
int vs_gather_round(size_t n_lanes, int n_shards, int shard, size_t chunk, size_t round, size_t *row0, size_t *rows);

static int fails = 0;
#define CHECK(cond)                                                     \
  do {                                                                  \
    if (!(cond)) {                                                      \
      fprintf(stderr, "FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond);   \
      fails++;                                                          \
    }                                                                   \
  } while (0)

static unsigned rng_state = 12345u;
static unsigned rnd(void)
{
  rng_state = rng_state * 1664525u + 1013904223u;
  return rng_state >> 8;
}

int main(void)
{
  const size_t n = 20000;
  vs_lane *lanes = (vs_lane *)malloc(n * sizeof(vs_lane));
  VsDevLane *dl = (VsDevLane *)malloc(n * sizeof(VsDevLane));
  CHECK(lanes && dl);
  for (size_t l = 0; l < n; l++) {
    CHECK(vs_lane_defaults(&lanes[l]) == VS_OK);
    lanes[l].fs = 16000;
    lanes[l].F0 = 80.0f + (float)(rnd() % 221);         /* periods 53..200: many distinct (P, T2) */
    lanes[l].Fg = lanes[l].F0 * 125.0f / 120.0f + 1.0f;
    lanes[l].jitter = 0.01f;
    lanes[l].shimmer = 0.0576f;
    lanes[l].noise = 100.0f;
    lanes[l].DC = 0.25f;
    lanes[l].flags = (rnd() & 1) ? (VS_FLAG_JITTER | VS_FLAG_SHIMMER | VS_FLAG_NOISE) : VS_FLAG_JITTER;
    lanes[l].vowel = "12467"[l % 5];
    lanes[l].seed = 1 + l;
  }
  CHECK(vs_expand_all(lanes, dl, n, 0) == VS_OK);
  for (size_t l = 0; l < n; l++) CHECK(dl[l].row == (int32_t)l && dl[l].P == (int)((float)lanes[l].fs / lanes[l].F0));

  /* the order: a permutation, sorted by (P, T2, flags), ties in input order */
  VsDevLane *sorted = (VsDevLane *)malloc(n * sizeof(VsDevLane));
  memcpy(sorted, dl, n * sizeof(VsDevLane));
  CHECK(vs_sort_lanes(&sorted, n) == VS_OK);
  unsigned char *seen = (unsigned char *)calloc(n, 1);
  for (size_t l = 0; l < n; l++) {
    CHECK(sorted[l].row >= 0 && (size_t)sorted[l].row < n);
    if (sorted[l].row >= 0 && (size_t)sorted[l].row < n) {
      CHECK(!seen[sorted[l].row]);
      seen[sorted[l].row] = 1;
      CHECK(memcmp(&sorted[l], &dl[sorted[l].row], sizeof(VsDevLane)) == 0);
    }
    if (l > 0) {
      const VsDevLane *a = &sorted[l - 1], *b = &sorted[l];
      const int lt = (a->P != b->P) ? (a->P < b->P) : (a->T2 != b->T2) ? (a->T2 < b->T2) : (a->flags < b->flags);
      const int eq = a->P == b->P && a->T2 == b->T2 && a->flags == b->flags;
      CHECK(lt || eq);
      if (eq) CHECK(a->row < b->row);
    }
  }
  free(seen);
  /* mixed rings over the sorted records (313 groups, the last one ragged): every group exactly once, every ring deep
   * enough for ITS periods and a multiple of the super-step, every workgroup inside the LDS, regions back to back;
   * batches of 1 .. 9 groups; a floor no workgroup can afford is refused and leaves nothing behind */
  for (int pass = 0; pass < 12; pass++) {
    const size_t m = pass == 0 ? n : (pass == 1 ? 64 : (pass == 2 ? 65 : (size_t)(64 * pass - 17)));
    const int floor_slots = (pass & 1) ? 144 : 192;
    VsGroupSlot *gm = NULL;
    size_t n_wg = 0, lds = 0;
    int c_min = 0, c_max = 0;
    const int rc = vs_mixed_rings_build(sorted, m, floor_slots, &gm, &n_wg, &lds, &c_min, &c_max);
    CHECK(rc == VS_OK);
    if (rc != VS_OK) continue;
    const size_t n_groups = (m + 63) / 64;
    CHECK(n_wg == (n_groups + 3) / 4 && lds <= VS_LDS_LIMIT && c_min >= floor_slots && c_max >= c_min);
    unsigned char *hit = (unsigned char *)calloc(n_groups, 1);
    for (size_t w = 0; w < n_wg; w++) {
      size_t off = 0;
      for (int k = 0; k < 4; k++) {
        const VsGroupSlot *gs = &gm[w * 4 + k];
        if (gs->group < 0) continue;
        CHECK((size_t)gs->group < n_groups && !hit[gs->group]);
        hit[gs->group] = 1;
        int tb = 1, need_ltab = 0, seen_t2[64], ns = 0;
        for (size_t l = (size_t)gs->group * 64; l < m && l < ((size_t)gs->group + 1) * 64; l++) {
          if (sorted[l].tbound > tb) tb = sorted[l].tbound;
          int dup = 0;
          for (int q = 0; q < ns; q++) dup |= seen_t2[q] == sorted[l].T2;
          if (!dup) {
            seen_t2[ns++] = sorted[l].T2;
            need_ltab += (sorted[l].T2 + 7) & ~7;
          }
        }
        CHECK(gs->ring_slots % VS_SS == 0 && gs->ring_slots >= VS_SS + tb + VS_TRASH_ROWS && gs->ring_slots >= floor_slots);
        CHECK((double)(gs->ring_slots - VS_SS) / (double)tb >= 1.65);
        CHECK(gs->ltab_entries == need_ltab && (gs->lds_off & 15) == 0 && (size_t)gs->lds_off == off);
        off += (((size_t)(gs->ring_slots + VS_TRASH_ROWS) * 128 + (size_t)gs->ltab_entries * 8 + VS_SYNC_WORDS_3 * 64 * 4) + 15) & ~(size_t)15;
      }
      CHECK(off <= VS_LDS_LIMIT && off <= lds);
    }
    for (size_t g = 0; g < n_groups; g++) CHECK(hit[g]);
    free(hit);
    free(gm);
  }
  {
    VsGroupSlot *gm = (VsGroupSlot *)0x1;
    size_t n_wg = 7, lds = 7;
    int c_min = 7, c_max = 7;
    CHECK(vs_mixed_rings_build(sorted, n, 1200, &gm, &n_wg, &lds, &c_min, &c_max) == VS_ERR_UNSUPPORTED);
    CHECK(gm == (VsGroupSlot *)0x1 && n_wg == 7);
    CHECK(vs_mixed_rings_build(sorted, 0, 192, &gm, &n_wg, &lds, &c_min, &c_max) == VS_ERR_ARG);
  }
  free(sorted);
  /* sizes around the merge widths */
  for (size_t m = 1; m <= 70; m++) {
    VsDevLane *s2 = (VsDevLane *)malloc(m * sizeof(VsDevLane));
    memcpy(s2, dl + 100, m * sizeof(VsDevLane));
    CHECK(vs_sort_lanes(&s2, m) == VS_OK);
    for (size_t l = 1; l < m; l++) CHECK(s2[l - 1].P <= s2[l].P);
    free(s2);
  }

  /* two bad lanes met by different threads: the lowest one's error is the answer */
  lanes[17000].F0 = 10.0f;                   /* below 50: VS_ERR_RANGE */
  lanes[3000].cq = 0.0f;                     /* no pulse: VS_ERR_UNSUPPORTED */
  CHECK(vs_expand_all(lanes, dl, n, 0) == VS_ERR_UNSUPPORTED);
  lanes[3000].cq = 0.55f;
  CHECK(vs_expand_all(lanes, dl, n, 0) == VS_ERR_RANGE);
  lanes[17000].F0 = 120.0f;
  CHECK(vs_expand_all(lanes, dl, n, 1) == VS_OK);  /* the filter-only records */

  /* ring policy: every period, both ring widths, caps and explicit requests */
  for (int tmax = 1; tmax <= 5200; tmax += (tmax < 300 ? 1 : 37)) {
    int slots = 0, thr = 0;
    for (int cap = 0; cap <= 1200; cap += 300) {
      for (int req = 0; req <= 1; req++) {
        const int rc = vs_ring_policy_for(VS_WAVE, tmax, cap, &slots, &thr, req ? cap : 0, req ? 2.4 : 1.7);
        if (rc == VS_OK) CHECK(slots % VS_SS == 0 && slots >= VS_SS + tmax + VS_TRASH_ROWS && (size_t)(slots + VS_TRASH_ROWS) * 128 <= VS_LDS_LIMIT);
        else CHECK(rc == VS_ERR_UNSUPPORTED);
      }
    }
    const int rc = vs_ring_slots_for(tmax, &slots);
    CHECK(rc == VS_OK || rc == VS_ERR_UNSUPPORTED);
  }
  double row[257];
  vs_cos_row(257, row);
  CHECK(row[0] == 1.0 && row[256] < -0.99);

  /* the rounds of a node's gather */
  for (int shards = 1; shards <= 9; shards++) {
    const size_t total = 100003, chunk = 4096;
    size_t covered = 0;
    const size_t rounds = vs_gather_rounds(total, shards, chunk);
    for (int s = 0; s < shards; s++) {
      size_t lo, hi, nxt;
      CHECK(vs_shard_cut(total, shards, s, &lo, &hi) == VS_OK);
      nxt = lo;
      for (size_t k = 0; k < rounds + 2; k++) {
        size_t r0, rows;
        CHECK(vs_gather_round(total, shards, s, chunk, k, &r0, &rows) == VS_OK);
        if (rows) {
          CHECK(r0 == nxt && rows <= chunk);
          nxt += rows;
        }
      }
      CHECK(nxt == hi);
      covered += hi - lo;
    }
    CHECK(covered == total);
  }
  free(lanes);
  free(dl);
  if (fails) return 1;
  printf("ok\n");
  return 0;
}
