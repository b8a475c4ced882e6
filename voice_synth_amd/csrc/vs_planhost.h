/*
 * vs_planhost.h -- the device-free half of plan creation (csrc/vs_planhost.c), shared with vs_api.c.
 */
#ifndef VS_PLANHOST_H
#define VS_PLANHOST_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/voice_synth.h"
#include "vs_device.h"

#ifdef __cplusplus
extern "C" {
#endif
/* A[0..40] of the lane's filter: the table or the explicit set, zeros behind its order */
int vs_lane_taps(const vs_lane *lane, double *A);
/* the ten tables by number (csrc/vs_host.c): 0..9, -1 / 0 for anything else */
int vs_vowel_index(int vowel);
int vs_vowel_by_index(int index);
/* The plan's tap table: rows 0..9 the ten tables, then one row per record whose tap_row is -1 (a coefficient set of the
 * lane's own, taken from lanes[record.row]; the record learns its row).  *taps is malloc'ed ([*rows][22] doubles, A[1..22]),
 * the caller frees it. */
int vs_tap_table_build(const vs_lane *lanes, VsDevLane *dl, size_t n_lanes, size_t n_custom, double **taps, size_t *rows);
bool vs_lane_is_wide(const vs_lane *lane);
/* VsDevLane.lframe_magic: the multiplier behind "sample index / frame length" in vs_out_noise_kernel */
uint32_t vs_lframe_magic(int Lframe);
/* one lane -> the record the kernels read (validated); the filter-only form fills what vowel reads */
int vs_expand_lane(const vs_lane *lane, int32_t row, VsDevLane *d);
int vs_expand_filter_lane(const vs_lane *lane, int32_t row, VsDevLane *d);
/* what a plan needs to know about its batch as a whole: gathered by the threads that make the records, so that the
 * plan does not walk all of them again for every question */
typedef struct VsBatchStats {
  int max_T2;       /* longest cos row */
  int tmax;         /* longest period any lane's rejection test admits (VsDevLane.tbound) */
  int min_lframe;   /* shortest frame (> 0) of the batch, 0: none */
  int max_lframe;   /* longest frame; == min_lframe (and no_lframe 0): every lane has the same */
  int no_lframe;    /* some lane has no frame at all (fs < 2000) */
  int any_onoise;   /* some lane asks for vowel -n */
  int pre1;         /* every lane has pre_emphasis == 1.0 */
  int wide;         /* some lane carries a coefficient set of 23..40 taps */
  size_t n_noisy;   /* lanes with glottal noise (VS_DF_NOISE) */
  size_t n_custom;  /* lanes that bring a coefficient set of their own (tap_row -1) */
} VsBatchStats;
/* all lanes, cut over up to 16 host threads for batches >= 8192; the failure of the lowest lane wins; stats may be NULL */
int vs_expand_all(const vs_lane *lanes, VsDevLane *dl, size_t n_lanes, int filter_only);
int vs_expand_all_stats(const vs_lane *lanes, VsDevLane *dl, size_t n_lanes, int filter_only, VsBatchStats *stats);
/* the source records in the order the kernels want -- stable by (P, T2, jitter / shimmer / noise on) -- written
 * straight into place: *order (malloc'ed, caller frees; NULL = input order: a homogeneous batch) says which lane
 * record i comes from (vs_kernel_order), vs_expand_all_ordered does both */
int vs_kernel_order(const vs_lane *lanes, size_t n_lanes, uint32_t **order);
int vs_expand_all_ordered(const vs_lane *lanes, VsDevLane *dl, size_t n_lanes, int *reordered, VsBatchStats *stats);
/* What plan creation keeps between calls (one per context): up to 16 worker threads that sleep between plans, the
 * record buffers, keys and sort arrays.  vs_expand_all_ordered_ws makes the records of a batch in it -- source kinds in
 * kernel order, filter-only in input order -- and *dl points into the workspace until the next call. */
typedef struct VsPlanWs VsPlanWs;
VsPlanWs *vs_planws_create(void);
void vs_planws_destroy(VsPlanWs *ws);
/* where the record buffers of big batches (8192 lanes and more) come from -- the context hands in page-locked memory, so
 * that their upload is a DMA transfer that runs next to a kernel; before the first batch only, NULL = malloc */
typedef void *(*vs_planws_alloc_fn)(void *user, size_t bytes);
typedef void (*vs_planws_free_fn)(void *user, void *ptr);
void vs_planws_set_big_allocator(VsPlanWs *ws, vs_planws_alloc_fn alloc, vs_planws_free_fn release, void *user);
int vs_expand_all_ordered_ws(VsPlanWs *ws, const vs_lane *lanes, size_t n_lanes, int filter_only, VsDevLane **dl,
                             int *reordered, VsBatchStats *stats);
void vs_cos_row(int T2, double *row);
/* ring capacity (slots per utterance) and super-step threshold for periods up to tmax */
int vs_ring_policy_for(int group_lanes, int tmax, int cap, int *slots, int *ready_min, int request, double depth);
int vs_ring_policy(int tmax, int cap, int *slots, int *ready_min);
int vs_ring_slots_for(int tmax, int *slots);
/* Mixed rings (vs_device.h, VsGroupSlot): the table of a full grid whose 64-utterance groups (lanes [64 g, 64 g + 64) of
 * the sorted records) differ in period -- every group with the ring depth ITS periods need (1.7 cycles, at least
 * floor_slots) and its own cos-row reservation, the groups dealt to workgroups of four in snake order of their longest
 * period.  VS_OK: *gmap (malloc'ed, [*n_wg][4], caller frees) with every group exactly once, the rest -1; *max_lds = the
 * largest workgroup's LDS bytes (<= VS_LDS_LIMIT), *c_min / *c_max the shallowest / deepest ring.  VS_ERR_UNSUPPORTED:
 * some workgroup would not fit (nothing allocated).  VS_ERR_NOMEM. */
int vs_mixed_rings_build(const VsDevLane *dl, size_t n_lanes, int floor_slots, VsGroupSlot **gmap, size_t *n_wg,
                         size_t *max_lds, int *c_min, int *c_max);
#ifdef __cplusplus
}
#endif
#endif
