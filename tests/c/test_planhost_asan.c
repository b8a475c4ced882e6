/* The device-free half of plan creation (voice_synth_amd/csrc/vs_planhost.c) and the gather bookkeeping of
 * vs_host.c under AddressSanitizer + UBSan + the thread pool of the expansion.  Built and run by
 * tests/test_host_sanitizers.py:
 *   gcc -fsanitize=address,undefined -fno-sanitize-recover=all -ffp-contract=off tests/c/test_planhost_asan.c \
 *       voice_synth_amd/csrc/vs_planhost.c voice_synth_amd/csrc/vs_host.c -Iinclude -lm -lpthread
 * What it goes through: the expansion of 20000 lanes of mixed periods on 8 threads, the failure of the LOWEST bad
 * lane whatever the thread (and the order) that met it, the kernel order by (P, T2, options) found from the lanes and
 * written straight into place -- a permutation, sorted, ties in input order --, the mixed-rings table of a batch of many periods (every group once, every ring deep, every workgroup inside
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

/* a stand-in for the context's page-locked allocator: counts, and refuses the calls whose bit is set in big_refuse_mask */
static int big_calls = 0, big_live = 0;
static unsigned big_refuse_mask = 0;
static void *big_alloc_stub(void *user, size_t bytes)
{
  const int k = big_calls++;
  if (k < 32 && (big_refuse_mask >> k) & 1u) return NULL;
  if (k >= 32 && big_refuse_mask == ~0u) return NULL;
  char *p = (char *)malloc(bytes + 16);
  if (!p) return NULL;
  memcpy(p, "BIGALLOC", 8); /* a block of ours: free() of p + 16 would be caught by the sanitizer */
  (*(int *)user)++;
  return p + 16;
}
static void big_free_stub(void *user, void *ptr)
{
  char *p = (char *)ptr - 16;
  if (memcmp(p, "BIGALLOC", 8) != 0) abort();
  (*(int *)user)--;
  free(p);
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
  {
    /* the batch's statistics, gathered by the threads that make the records, against a walk over the records */
    VsBatchStats st, st2;
    lanes[11].pre_emphasis = 0.5f;
    lanes[12].out_snr = 100.0f;
    CHECK(vs_expand_all_stats(lanes, dl, n, 0, &st) == VS_OK);
    int max_T2 = 0, tmax = 1, minlf = 0, pre1 = 1;
    size_t noisy = 0;
    for (size_t l = 0; l < n; l++) {
      if (dl[l].T2 > max_T2) max_T2 = dl[l].T2;
      if (dl[l].tbound > tmax) tmax = dl[l].tbound;
      if (dl[l].Lframe > 0 && (minlf == 0 || dl[l].Lframe < minlf)) minlf = dl[l].Lframe;
      if (dl[l].pre != 1.0) pre1 = 0;
      if (dl[l].flags & VS_DF_NOISE) noisy++;
    }
    CHECK(st.max_T2 == max_T2 && st.tmax == tmax && st.min_lframe == minlf && st.pre1 == pre1 && pre1 == 0);
    CHECK(st.n_noisy == noisy && st.any_onoise == 1 && st.wide == 0);
    VsDevLane *tmp = (VsDevLane *)malloc(n * sizeof(VsDevLane));
    CHECK(vs_expand_all_ordered(lanes, tmp, n, NULL, &st2) == VS_OK);
    CHECK(memcmp(&st, &st2, sizeof(st)) == 0);       /* the same batch in the kernels' order */
    free(tmp);
    lanes[11].pre_emphasis = 1.0f;
    lanes[12].out_snr = 0.0f;
    CHECK(vs_expand_all_stats(lanes, dl, n, 0, &st) == VS_OK && st.pre1 == 1 && st.any_onoise == 0);
  }

  /* the order: the records of vs_expand_all_ordered are a permutation of vs_expand_all's, sorted by (P, T2, options),
   * ties in input order; vs_kernel_order says the same */
  VsDevLane *sorted = (VsDevLane *)malloc(n * sizeof(VsDevLane));
  int reordered = 0;
  CHECK(vs_expand_all_ordered(lanes, sorted, n, &reordered, NULL) == VS_OK && reordered == 1);
  uint32_t *order = NULL;
  CHECK(vs_kernel_order(lanes, n, &order) == VS_OK && order != NULL);
  unsigned char *seen = (unsigned char *)calloc(n, 1);
  for (size_t l = 0; l < n; l++) {
    CHECK(sorted[l].row >= 0 && (size_t)sorted[l].row < n);
    if (sorted[l].row >= 0 && (size_t)sorted[l].row < n) {
      CHECK(!seen[sorted[l].row]);
      seen[sorted[l].row] = 1;
      CHECK(memcmp(&sorted[l], &dl[sorted[l].row], sizeof(VsDevLane)) == 0);
      CHECK(order && order[l] == (uint32_t)sorted[l].row);
    }
    if (l > 0) {
      const VsDevLane *a = &sorted[l - 1], *b = &sorted[l];
      const unsigned fa = a->flags & 7u, fb = b->flags & 7u; /* jitter / shimmer / noise on: the option bits of the key */
      const int lt = (a->P != b->P) ? (a->P < b->P) : (a->T2 != b->T2) ? (a->T2 < b->T2) : (fa < fb);
      const int eq = a->P == b->P && a->T2 == b->T2 && fa == fb;
      CHECK(lt || eq);
      if (eq) CHECK(a->row < b->row);
    }
  }
  free(order);
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
  /* small batches, and a homogeneous one: input order, no index array */
  for (size_t m = 1; m <= 70; m++) {
    VsDevLane *s2 = (VsDevLane *)malloc(m * sizeof(VsDevLane));
    CHECK(vs_expand_all_ordered(lanes + 100, s2, m, NULL, NULL) == VS_OK);
    for (size_t l = 1; l < m; l++) CHECK(s2[l - 1].P <= s2[l].P);
    free(s2);
  }
  {
    vs_lane same[40];
    VsDevLane out[40];
    uint32_t *ord = (uint32_t *)0x1;
    for (int i = 0; i < 40; i++) {
      same[i] = lanes[7];
      same[i].seed = 100 + (uint64_t)i;
    }
    int re = 1;
    CHECK(vs_kernel_order(same, 40, &ord) == VS_OK && ord == NULL);
    CHECK(vs_expand_all_ordered(same, out, 40, &re, NULL) == VS_OK && re == 0);
    for (int i = 0; i < 40; i++) CHECK(out[i].row == i && out[i].key0 == (uint32_t)(100 + i));
  }

  /* ONE workspace over batches of growing and shrinking size (what a context does: chunks of 16384 lanes, then a
   * whole batch): every answer equals the one-shot function's, whichever buffers the workspace already holds */
  {
    VsPlanWs *ws = vs_planws_create();
    CHECK(ws != NULL);
    const size_t sizes[] = {300, 9000, 64, 20000, 8192, 1, 20000};
    VsDevLane *want = (VsDevLane *)malloc(n * sizeof(VsDevLane));
    CHECK(want != NULL);
    for (size_t k = 0; k < sizeof(sizes) / sizeof(sizes[0]); k++) {
      const size_t m = sizes[k];
      VsDevLane *got = NULL;
      VsBatchStats sa, sb;
      int ra = -1, rb = -1;
      CHECK(vs_expand_all_ordered_ws(ws, lanes, m, 0, &got, &ra, &sa) == VS_OK && got != NULL);
      CHECK(vs_expand_all_ordered(lanes, want, m, &rb, &sb) == VS_OK);
      CHECK(ra == rb && memcmp(&sa, &sb, sizeof(sa)) == 0);
      if (got) CHECK(memcmp(got, want, m * sizeof(VsDevLane)) == 0);
      CHECK(vs_expand_all_ordered_ws(ws, lanes, m, 1, &got, &ra, NULL) == VS_OK && ra == 0); /* filter-only: input order */
      CHECK(vs_expand_all(lanes, want, m, 1) == VS_OK);
      if (got) CHECK(memcmp(got, want, m * sizeof(VsDevLane)) == 0);
    }
    VsDevLane *got = NULL;
    CHECK(vs_expand_all_ordered_ws(ws, lanes, 0, 0, &got, NULL, NULL) == VS_ERR_ARG);
    CHECK(vs_expand_all_ordered_ws(NULL, lanes, 5, 0, &got, NULL, NULL) == VS_ERR_ARG);
    free(want);
    vs_planws_destroy(ws);
    vs_planws_destroy(NULL);
  }

  /* the workspace under page-locked-memory pressure: the context's allocator for big record buffers (hipHostMalloc there,
   * a counting stand-in here) refuses its first, its second, or every request -- a batch whose lanes must be put in
   * kernel order needs TWO buffers, and either may end up in ordinary memory; the answer is the same and every block goes
   * back to where it came from */
  {
    for (int refuse = 0; refuse < 4; refuse++) {
      big_calls = big_live = 0;
      big_refuse_mask = (refuse == 3) ? ~0u : (1u << refuse);
      VsPlanWs *ws = vs_planws_create();
      CHECK(ws != NULL);
      vs_planws_set_big_allocator(ws, big_alloc_stub, big_free_stub, &big_live);
      VsDevLane *want = (VsDevLane *)malloc(n * sizeof(VsDevLane));
      CHECK(want != NULL);
      for (int round = 0; round < 2; round++) {
        VsDevLane *got = NULL;
        int ra = -1, rb = -1;
        CHECK(vs_expand_all_ordered_ws(ws, lanes, n, 0, &got, &ra, NULL) == VS_OK && got != NULL && ra == 1);
        CHECK(vs_expand_all_ordered(lanes, want, n, &rb, NULL) == VS_OK && rb == 1);
        if (got) CHECK(memcmp(got, want, n * sizeof(VsDevLane)) == 0);
      }
      CHECK(big_calls >= 2);             /* both buffers were asked of the big allocator ... */
      free(want);
      vs_planws_destroy(ws);
      CHECK(big_live == 0);              /* ... and what it gave has been given back to it, nothing of it to free() */
    }
  }

  /* two bad lanes met by different threads: the lowest one's error is the answer */
  lanes[17000].F0 = 10.0f;                   /* below 50: VS_ERR_RANGE */
  lanes[3000].cq = 0.0f;                     /* no pulse: VS_ERR_UNSUPPORTED */
  CHECK(vs_expand_all(lanes, dl, n, 0) == VS_ERR_UNSUPPORTED);
  CHECK(vs_expand_all_ordered(lanes, dl, n, NULL, NULL) == VS_ERR_UNSUPPORTED); /* ... in kernel order too: lane 3000 is the lowest */
  lanes[3000].cq = 0.55f;
  CHECK(vs_expand_all(lanes, dl, n, 0) == VS_ERR_RANGE);
  CHECK(vs_expand_all_ordered(lanes, dl, n, NULL, NULL) == VS_ERR_RANGE);
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
