/*
 * vs_device.h -- records shared by the host-side plan builder and the gfx950 kernels.
 */
#ifndef VS_DEVICE_H
#define VS_DEVICE_H

#include <stdint.h>

#define VS_WAVE 64 /* lanes per wavefront on gfx950: one utterance per lane */
#define VS_LDS_LIMIT (160 * 1024) /* LDS per CU on gfx950 */
#define VS_NARROW_LANES 16 /* utterances per wavefront of the narrow build (periods beyond the 64-column ring) */
#define VS_SS 24   /* samples per filter super-step == size of the rotating y[] register window */

/* device-side lane flags */
#define VS_DF_JITTER 0x1u  /* -j given and non-zero   (flowgen_shimmer.c:248) */
#define VS_DF_SHIMMER 0x2u /* -s given and non-zero   (flowgen_shimmer.c:295) */
#define VS_DF_NOISE 0x4u   /* -n given                (flowgen_shimmer.c:373) */
/* Host-proved bounds that let the generator take its short instruction sequences:
 *   - every pulse sample fits a signed short before the (signed short) cast, so the cast is the
 *     identity and both half-pulses are monotone in i (amplitude <= 32767 and
 *     (2*K*(1+Kvar) - 1) * amplitude <= 32767 for the largest admissible amplitude);
 *   - the open phase ends at least 8 samples before the shortest admissible period
 *     (2*T2 + 8 <= min T), so stores that run a few slots past a phase are overwritten by the
 *     phases behind it. */
#define VS_DF_FAST 0x8u

/* three-role kernel: order boxes per lane between the open-phase and the noise wavefront, and the
 * LDS words per lane of a group's progress bookkeeping (gpub, npub, oseq, otak, 3 words per box) */
#ifndef VS_ORDER_DEPTH
#define VS_ORDER_DEPTH 2
#endif
#define VS_SYNC_WORDS_3 (4 + 3 * VS_ORDER_DEPTH)

#define VS_TRASH_ROWS 8 /* ring rows [C, C+8): where lanes that must not emit send their 8-sample trips */

/* One utterance as the kernel reads it (128 bytes, 8-byte aligned).  Everything that is a
 * pure function of the lane's parameters is evaluated on the host, in C, with the reference's
 * operand types (vs_expand_lane() in vs_planhost.c).  The 22 taps of its filter are NOT in the record: they are a row of
 * the plan's tap table (VsKernelArgs.taps) -- rows 0..9 the reference's ten tables in the order of vs_vowel_by_index(),
 * one more row per utterance that brings a coefficient set of its own -- so that 65536 utterances on five tables move
 * 8 MB of records through the host and over PCIe instead of 19. */
#define VS_TAP_TABLE_ROWS 10
typedef struct VsDevLane {
  double gain;        /* (double)gain                                  vowel_new.c:268 */
  double pre;         /* (double)pre_emphasis                          vowel_new.c:284 */
  float jitter, shimmer, K, Kvar, DC, noise;
  float t_hi, t_lo;   /* (float)1.2*P, (float)0.8*P                    flowgen_shimmer.c:290 */
  float a_hi, a_lo;   /* (float)1.8*amp, (float)0.2*amp                flowgen_shimmer.c:306 */
  int32_t amp;
  int32_t P;          /* (int)((float)fs/F0)                           flowgen_shimmer.c:244 */
  int32_t T2;         /* ceil(0.5*cq*P)                                flowgen_shimmer.c:317 */
  int32_t tab_off;    /* first entry of this lane's cos(PI*k/T2) row in the table */
  int32_t tbound;     /* longest period the rejection test admits (P without jitter) */
  int32_t dcs;        /* (short)par.DC                                 flowgen_shimmer.c:321,335 */
  uint32_t flags;     /* VS_DF_* */
  uint32_t key0, key1;/* Philox key */
  int32_t row;        /* output row of this lane */
  float out_snr;      /* vowel -n: linear SNR of the noise added to the filtered signal, 0 = off */
  int32_t Lframe;     /* 50 * ((int)(fs*0.001/2.0)*2), the frame the noise power is taken over (vowel_new.c:361-363) */
  uint32_t okey0, okey1; /* Philox key of the vowel stage's draw stream */
  int32_t thr;        /* ceil(par.DC) as an integer: for an integer x, (float)x < par.DC  <=>  x < thr   (fg:320, 329) */
  int32_t ready_min;  /* super-step threshold of this lane's 64-utterance group (the same in all its lanes): ready lanes * 64 >= live lanes * ready_min */
  int32_t tap_row;    /* row of A[1..22] in the plan's tap table (vowel_new.c:279-281); from vs_expand_lane: 0..9 = a table, -1 = a set of the lane's own (the plan gives it a row) */
  uint32_t lframe_magic; /* i / Lframe == umulhi(i, lframe_magic) >> (ceil(log2(Lframe)) - 1) for 0 <= i < 2^31 (vs_lframe_magic) */
} VsDevLane;

/* Wave-specialised launches whose groups differ in period (an F0 sweep): one record per (workgroup, slot) -- which
 * 64-utterance group the slot serves, how many ring slots it has and where its LDS starts.  A workgroup then holds
 * groups from ACROSS the period range, the short-period ones lending LDS to the long-period ones, so that every ring
 * holds >= 1.65 of its group's longest cycles (vs_plan_create_impl: "mixed rings"). */
typedef struct VsGroupSlot {
  int32_t group;      /* index of the 64-utterance group (lanes [64*group, 64*group + 64) of the sorted records); -1: none */
  int32_t ring_slots; /* C of this group's ring (multiple of VS_SS) */
  int32_t lds_off;    /* byte offset of the group's LDS region in the workgroup's allocation (16-byte multiple) */
  int32_t ltab_entries; /* doubles reserved behind THIS group's ring for its cos rows (a multiple of 2: the progress words behind them stay 16-byte aligned) */
} VsGroupSlot;

typedef struct VsKernelArgs {
  const VsDevLane *lanes;
  const double *costab;
  const int16_t *in;
  int16_t *out;
  void *log;          /* vs_cycle_rec* */
  int32_t *ncyc;
  long in_pitch, out_pitch, log_pitch;
  int n_lanes;
  int n_samples;
  int ring_slots;
  int vec_ok;         /* 1: every row start is 4-byte aligned, 16-byte vector stores allowed */
  int ltab_entries;   /* doubles reserved behind the ring for this wavefront's cos rows */
  int ready_min;      /* > 0: super-step threshold for every group (vs_tuning); 0: each group's own VsDevLane.ready_min */
  int ws_pairs;       /* wave-specialised kernels: groups of 64 utterances per workgroup (1, 2 or 4) */
  int ws_roles;       /* wavefronts per group: 2 (generator | filter) or 3 (open phase | noise | filter) */
  int group_lanes;    /* utterances per wavefront of the one-wave kernel: 64 (or 0), or 16 = the narrow build for long periods */
  int ws_pair_bytes;  /* LDS bytes of one pair: ring + trash row + cos rows + progress words, 16-byte multiple */
  int gen_min;        /* wave-specialised kernel: generate when want lanes * 64 >= needing lanes * gen_min */
  float *ondw;        /* vowel -n: NoiseDistWidth of every frame [n_lanes][ondw_pitch] (vs_out_power_kernel -> vs_out_noise_kernel), NULL when no lane asks for it */
  long ondw_pitch;
  int32_t *odone;     /* vowel -n behind a wave-specialised kernel that takes the frame powers along (pow_lframe > 0): frames of row r it has
                         dealt with, [n_lanes]; the streaming pass fills in the rest.  NULL: the streaming pass does every frame */
  int pow_lframe;     /* ... the frame length every lane of the launch shares, 0: lanes differ (or no lane asks for output noise) */
  int gen_low;        /* wave-specialised kernel: a lane with fewer buffered samples than this starts a round at once */
  int *err;           /* device word: bit 0/1 set when a bounded spin of the generator/filter wave ran out */
  int spin_limit;     /* polls before a waiting wave gives up and sets err */
  int fault;          /* VS_FAULT_* (tests only) */
  int ws_filter_prio; /* wave-specialised kernel: s_setprio of the filter wave (0 = leave at 0) */
  unsigned long long *diag; /* VS_DIAG builds only: per-wavefront cycle counters [grid][8] */
  int16_t *sink;       /* one row of n_samples + 32 samples nobody reads: where the lanes beyond n_lanes of the last group store (wave-specialised kernels) */
  const double *awide; /* wide plans (a coefficient set of 23..40 taps): A[1..40] per lane record, zeros behind its order */
  int ws_layout;       /* wave-specialised kernels, how a workgroup's wavefronts map to roles: VS_WS_LAYOUT_* */
  const double *taps;  /* the plan's tap table: [rows][22], A[1..22] of row VsDevLane.tap_row */
  const VsGroupSlot *group_map; /* mixed rings: [workgroups][ws_pairs]; NULL: group = workgroup * ws_pairs + slot, uniform rings */
} VsKernelArgs;

/* Wavefronts of a workgroup are dealt to the CU's four SIMDs four at a time: wavefront w runs on the SIMD of wavefront
 * w % 4, and the first four on four different SIMDs (on MI355X a rotation of 0, 2, 1, 3 -- read from HW_ID by
 * vs_ctx_simd_dealing, which every plan that picks one of these layouts consults; "SIMD k" below = the SIMD of wavefront k).
 *   ROLE_MAJOR  role = w / groups, group = w % groups: with four groups per workgroup the two or three wavefronts
 *               of ONE group share a SIMD (full grids); with one or two groups every wavefront has a SIMD of its own.
 *   SPREAD_2X3  two groups x three roles in EIGHT wavefronts, F0 O0 F1 O1 -- N0 -- N1 (two of them leave at once):
 *               SIMDs 0 and 2 host a filter wavefront ALONE, SIMDs 1 and 3 the open-phase and the noise wavefront of
 *               one group -- half-filled chips (BASELINE config 4's shard), where the lone filter wavefront is the
 *               bound and the generator's work, cut in two, finishes well inside its shadow. */
#define VS_WS_LAYOUT_ROLE_MAJOR 0
#define VS_WS_LAYOUT_SPREAD_2X3 1

#define VS_WIDE_ORDER 40 /* == VS_MAX_ORDER of voice_synth.h */
#define VS_WIDE_SS 48    /* register window and super-step of the wide filter kernel (>= VS_WIDE_ORDER + 1, multiple of 8) */

#endif
