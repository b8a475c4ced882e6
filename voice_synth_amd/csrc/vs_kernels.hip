/*
 * vs_kernels.hip -- gfx950 (MI355X) kernels of the batched vowel synthesiser.
 *
 * Mapping: ONE UTTERANCE PER LANE, 64 utterances per group.  A group is served either by one
 * wavefront that alternates between generating and filtering (vs_synth_kernel) or by a generator
 * wavefront and a filter wavefront (vs_synth_ws_kernel, the default for the fused kind).  Either
 * way the work is CYCLE-MAJOR: lanes do not share a sample clock.  Each lane owns a column of
 * an int16 ring in LDS (layout [slot][lane], 128 B per slot; a lane only ever touches its own
 * column) and its own position n in its own utterance.  The two kinds of work:
 *
 *   generator round (reference flowgen_shimmer.c:246-423): every lane with room produces its
 *       next glottal cycle -- jitter / shimmer recursions with their rejection loops, rising
 *       and falling half-pulse from a host-built cos table (staged in LDS), closed phase,
 *       closed-phase noise from a counter-based Philox stream (4 draws per block) -- and
 *       appends T samples to its ring column.  All lanes walk the SAME phase of their own
 *       cycle together, so branches stay nearly wave-uniform although every lane has its own
 *       period, amplitude and draw counter; and
 *
 *   filter super-steps (reference vowel_new.c:266-289): while a lane holds >= 24 buffered flow
 *       samples it runs 24 steps of the order-22 all-pole recurrence in fp64.  The state
 *       y[n-1..n-22] lives in a rotating window of 24 double registers (no shifting, no LDS);
 *       the 24 int16 results leave as three 16-byte stores per lane at that lane's own n.
 *       The flow itself never reaches HBM.
 *
 * Next to them: vs_out_noise_kernel (vowel -n, second half), vs_filter_wide_kernel (explicit
 * coefficient sets of 23..40 taps: the same recurrence on a 48-sample window, reading a flow row
 * from HBM -- the un-fused path), vs_selftest_kernel (the arithmetic shortcuts against their
 * literal forms, on the device).
 *
 * No MFMA: the path is a scalar recurrence per utterance, not a contraction -- and the block form that would make
 * it one (16 samples x 16 utterances per tile, 40 multiply-adds per sample instead of 22) loses on gfx950, whose
 * fp64 matrix instructions run at the vector rate on the vector issue port (tools/ubench/ubench6.hip).
 *
 * Arithmetic contract: this file is compiled with -ffp-contract=off.  VS_ARITH_EXACT keeps
 * the reference's rounding sequence operation by operation (mul, then sub, j = 1..22), so the
 * double state is bit-identical to the C reference; VS_ARITH_FMA is the explicit, opt-in
 * fused variant.  IEEE fp64 mul/add/fma/div and fp32 mul/add/div are correctly rounded on
 * gfx950 (HIP's default -fhip-fp32-correctly-rounded-divide-sqrt is kept); cos() values come
 * from the host libm table; the device sqrt() only seeds an exact fix-up (an integer search in
 * the source, one exact Newton test in the output-noise kernel).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/voice_synth.h"
#include "vs_device.h"

#include "vs_dev_primitives.h"
#include "vs_dev_generator.h"
#include "vs_dev_filter.h"


/* per-lane constants of the generator from the lane record */
__device__ __forceinline__ void vs_load_cfg(const VsDevLane *__restrict__ L, VsCfg &c, VsGen &s)
{
  c.jitter = L->jitter; c.shimmer = L->shimmer; c.K = L->K; c.Kvar = L->Kvar;
  c.DC = L->DC; c.noise = L->noise; c.t_hi = L->t_hi; c.t_lo = L->t_lo;
  c.a_hi = L->a_hi; c.a_lo = L->a_lo;
  c.amp = L->amp; c.P = L->P; c.T2 = L->T2; c.tab_off = 0;
  c.dcs = L->dcs;
  c.thr = L->thr;
  c.flags = L->flags; c.key0 = L->key0; c.key1 = L->key1;
  s.d = 0u;
  s.dp0 = 0.0f; s.ds0 = 0.0f;
  s.T4 = 0; s.T = c.P; s.g = 0; s.wpos = 0; s.cyc = 0;
  s.amp_next = 0.0f; s.S_next = 0.0f; s.K_next = 0.0f; s.pend = false; s.posted = 0;
}

/* Stage the cos rows this wavefront needs in LDS: one pass per distinct T2 among its lanes
 * (one pass for a homogeneous batch).  A row occupies T2 rounded up to a multiple of 8, padded
 * with 1.0: the 8-sample trips of the short rising sequence read past T2 and get
 * ceil(Ah * (1 - 1)) = 0 there, which adds nothing to the power sum.  The host sized the region
 * for the worst wavefront of the plan (ltab_entries, same rounding).  Sets c.tab_off. */
__device__ __forceinline__ void vs_stage_cos_rows(const VsDevLane *__restrict__ L, VsCfg &c,
                                                  double *ltab, const double *__restrict__ costab,
                                                  int ltab_entries, int lane, bool valid, const VsKernelArgs &args)
{
  const int gtab = L->tab_off;
  int used = 0;
  bool pending = valid;
  /* tests (vs_tuning.fault): a kernel that believes the plan reserved no room for its cos rows */
  if (args.fault == VS_FAULT_SHORT_COS_ROWS) ltab_entries = 0;
  while (__any(pending)) {
    const unsigned long long m = __ballot(pending);
    const int leader = __builtin_ctzll(m);
    const int T2s = __builtin_amdgcn_readlane(c.T2, leader);
    const int gs = __builtin_amdgcn_readlane(gtab, leader);
    const int T2p = (T2s + 7) & ~7;
    if (used + T2p > ltab_entries) {
      /* Plan and kernel disagree about the room for the cos rows ("cannot happen"): like every other
       * one of those, it sets the launch's error word -- vs_plan_status() answers VS_ERR_INTERNAL -- and
       * the launch runs to its end: the lanes that are still pending keep row offset 0 and synthesise
       * from whatever lies there -- staged rows of other lanes, progress words, uninitialised LDS, possibly NaN or
       * infinity: garbage, but harmless garbage (every loop bound of the generator is a lane constant or comes
       * from the draws, none from the cos values; the conversions saturate), and nothing is written outside
       * the staged region.  The caller discards the rows (include/voice_synth.h, vs_plan_status). */
      if (args.err && lane == 0) atomicOr(args.err, 8);
      break;
    }
    for (int k = lane; k < T2p; k += VS_WAVE) ltab[used + k] = (k < T2s) ? costab[gs + k] : 1.0;
    if (pending && c.T2 == T2s) {
      c.tab_off = used;
      pending = false;
    }
    used += T2p;
  }
}

template <int ARITH, int KIND, bool LOG, bool PRE1>
__global__ void __launch_bounds__(VS_WAVE) vs_synth_kernel(VsKernelArgs args)
{
  extern __shared__ __attribute__((aligned(16))) int16_t ring[];

  const int lane = (int)threadIdx.x;
  const long gl = (long)blockIdx.x * VS_GROUP_LANES + lane;
  const bool valid = (lane < VS_GROUP_LANES) && (gl < (long)args.n_lanes);
  const VsDevLane *__restrict__ L = args.lanes + (valid ? gl : (long)args.n_lanes - 1);
  const int N = args.n_samples;
  const int C = args.ring_slots;

  /* filter constants and state in registers */
  double a[VS_ORDER + 1];
  double y[VS_SS];
  vs_load_taps<ARITH>(args.taps, L, a);
#pragma unroll
  for (int j = 0; j < VS_SS; ++j) y[j] = 0.0; /* vowel_new.c:222-224 */
  const double gain = L->gain;
  const double pre = L->pre;
  const long row = (long)L->row;
  /* the group's super-step threshold (the same in all its lanes), unless the launch sets one */
  const int ready_min = (args.ready_min > 0) ? args.ready_min : __builtin_amdgcn_readfirstlane(L->ready_min);

  VsCfg c;
  VsGen s;
  double *ltab = (double *)(ring + (size_t)(C + VS_TRASH_ROWS) * VS_GROUP_LANES); /* slots [C, C+8) are the trash rows */
  if (KIND != VS_KIND_FILTER) {
    vs_load_cfg(L, c, s);
    vs_stage_cos_rows(L, c, ltab, args.costab, args.ltab_entries, lane, valid, args);
    __syncthreads(); /* single-wave workgroup: orders the staging writes before the row reads */
  }
  vs_cycle_rec *logrow = nullptr;
  if (LOG && args.log) logrow = (vs_cycle_rec *)args.log + row * args.log_pitch;

  int16_t *__restrict__ orow = args.out + row * args.out_pitch;
  const int16_t *__restrict__ irow = (KIND == VS_KIND_FILTER) ? args.in + row * args.in_pitch : nullptr;

  VsDiag dg;
#ifdef VS_DIAG
#pragma unroll
  for (int k = 0; k < 8; ++k) dg.acc[k] = 0;
  dg.t = vs_stamp();
#endif
  vs_u32x4 xpre[VS_SS / 8]; /* filter-only kind: the next super-step's input, loaded ahead */
  if (KIND == VS_KIND_FILTER && valid && args.vec_ok && VS_SS <= N) {
#pragma unroll
    for (int k = 0; k < VS_SS / 8; ++k) xpre[k] = *(const vs_u32x4 *)(irow + 8 * k);
  }
  int n = 0;     /* this lane's position in its own utterance */
  int rslot = 0; /* ring slot of sample n */
  bool live = valid;
  /* Scheduling: a super-step costs the same whether 1 or 64 lanes take
   * part, and so does a generator round.  Run a super-step when at least ready_min/64 of the
   * live lanes hold 24 samples; otherwise let every lane with room produce its next cycle
   * (the lanes that are short always have room).  Lanes whose periods run long fill their
   * ring and sit rounds out -- they have fewer cycles to produce anyway.
   * A lane's scalar draws for cycle k+1 are made right behind the samples of cycle k, inside the
   * generator branch: the room check needs the period, and the generator's constants are then
   * not live across the super-step (which needs every register it can get). */
  if (KIND != VS_KIND_FILTER) {
    if (live && (s.g < N)) vs_cycle_scalars(c, s, dg); /* fixes the first period s.T */
  }
  while (__any(live)) {
    bool ready = live;
    if (KIND != VS_KIND_FILTER) {
      ready = live && ((s.g - n >= VS_SS) || (s.g >= N));
      /* room for the whole cycle plus the 8 slots a trip of the short sequences may run past it */
      const bool want = live && s.pend && (s.g - n + s.T + VS_TRASH_ROWS <= C);
      const int n_live = __builtin_popcountll(__ballot(live));
      const int n_ready = __builtin_popcountll(__ballot(ready));
      const bool filter_now = (n_ready > 0) && ((n_ready * 64 >= n_live * ready_min) || !__any(want));
      if (!filter_now) {
        if (want) {
          vs_cycle_emit<LOG>(c, s, ring, C, lane, N, ltab, logrow, (int)args.log_pitch, dg);
          if (s.g < N) vs_cycle_scalars(c, s, dg); /* the next period */
        }
        continue;
      }
    }

    /* ---- filter super-steps: a lane runs while it holds 24 buffered samples (or its tail) ---- */
    if (ready) {
      int outv[VS_SS];
      vs_superstep<ARITH, KIND, PRE1>(a, y, gain, pre, ring + rslot * VS_GROUP_LANES + lane, irow, orow, n, N,
                                      args.vec_ok != 0, outv, xpre);
      if (KIND != VS_KIND_FILTER) {
        rslot += VS_SS;
        if (rslot >= C) rslot = 0;
      }
      n += VS_SS;
      if (n >= N) live = false;
    }
    VS_DIAG_ADD(dg, 6)
  }
#ifdef VS_DIAG
  if (args.diag && lane == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) args.diag[(size_t)blockIdx.x * 8 + k] = dg.acc[k];
  }
#endif

  if (KIND != VS_KIND_FILTER && args.ncyc && valid) args.ncyc[row] = s.cyc;
}

#if VS_GROUP_LANES == VS_WAVE /* the narrow build holds the one-wave kernel only */
/*
 * Wave-specialised fused kernels: the same 64 utterances are served by SEVERAL wavefronts, each with
 * one job, coupled through the LDS ring and a few per-lane progress words.  This is what the fused
 * kind launches by default (vs_plan_create):
 *
 *   two roles (generator | filter) -- whatever does not suit the third role (no glottal noise; rings of
 *     barely one cycle: vs_plan_create_impl).  On grids that leave at least half of the chip's SIMDs empty
 *     every wavefront has a SIMD of its own and a launch takes max(generator, filter) instead of their sum;
 *
 *   three roles (open phase | noise | filter) -- full grids over deep rings (BASELINE config 3: 1024
 *     groups on 1024 SIMDs): four groups per 768-thread workgroup, one workgroup per CU, wavefronts laid out
 *     role-major so that every SIMD hosts the three wavefronts of ONE group (a workgroup's wavefronts
 *     are dealt to the CU's four SIMDs cyclically: w, w+4, w+8 share a SIMD).  Why three: a lone
 *     wavefront issues one instruction -- vector, scalar or LDS alike -- every ~5.25 cycles, and the
 *     filter wavefront at raised priority leaves a second wavefront about a quarter of that
 *     (tools/ubench/ubench3.hip), so a launch of the two-role kernel takes about 0.74 x filter + generator:
 *     whatever the generator cannot do in the filter's shadow it does ALONE on its SIMD while the
 *     filter sleeps.  Two generator wavefronts side by side run at 5.25 and 11 cycles per
 *     instruction (ubench4.hip), i.e. that part goes ~1.5x faster when the generator's work is cut
 *     in two: jitter / shimmer / both flanks here, the closed phase's noise there.
 *     Half-filled chips with glottal noise (BASELINE config 4 sharded over 8 GPUs: 512 groups on 1024 SIMDs; the
 *     16384-utterance chunks of the pipelines) take three roles too, laid out so that the filter wavefront -- the
 *     bound in exact arithmetic -- has a SIMD to itself and the two generator wavefronts share the next one
 *     (VS_WS_LAYOUT_SPREAD_2X3, vs_device.h: TWO groups per workgroup, the launcher refuses anything else; a grid of
 *     at most one group per CU runs role-major with one group per workgroup, a SIMD per wavefront), over rings of 2.4 cycles.
 *
 * Hand-off (workgroup scope, LDS only), per lane l:
 *   gpub[l] = samples of l that are complete in the ring     (written by the generator / by the noise wavefront)
 *   npub[l] = samples of l the filter has read from the ring  (written by the filter wavefront)
 *   three roles: oseq[l] / otak[l] = orders posted by the open-phase wavefront / taken by the noise
 *   wavefront, ord[0..2][l] = the order (vs_post_order): which draws, which samples, what width.
 * The LDS executes one wavefront's operations in order, so "ring writes, then gpub" on one side and
 * "gpub read, then ring reads" on the other is a release/acquire pair; the fences keep the compiler
 * from reordering.  A cycle's slots [g, g+T) are written only when g - npub + T <= C (T = the cycle's
 * period, fixed by vs_cycle_scalars), i.e. never over samples the filter has not consumed.  In the
 * three-role kernel two wavefronts write the ring at the same time -- cycle c's noise and cycle
 * c+1's flanks -- so the noise wavefront stops exactly at the end of its cycle (vs_noise_trips<TAIL>),
 * while the open-phase wavefront only ever runs past a phase INSIDE its own cycle and finishes all
 * of that before it posts the order.
 *
 * Progress: a lane that is short of 24 samples always has room for its next cycle, and a generator
 * round starts whenever such a lane exists; if no lane has room every lane holds more than 24 samples
 * and the filter runs.  Spins are bounded (args.spin_limit polls, then the error word of the launch
 * is set and the wavefront leaves) so that a protocol bug cannot hang the device.
 */
/* s_sleep units of 64 cycles between two polls of a waiting wavefront.  A polling wavefront takes issue slots from the
 * working ones, a sleeping one reacts late.  Two roles on a full grid (BASELINE config 5): 32 (A/B 1, 2, 8, 32: best by
 * ~2 %; 8 is 1.2 % slower).  Three roles: 8 -- each wavefront has half the work per sample, hand-offs are twice as
 * frequent, and a late reaction costs more than the polls (config 3: 2.81-2.86 against 2.88-2.91 ms at 32, four and
 * five same-box repetitions; 4 .. 16 are alike; profiles/r04_poll_sleep_ab.txt).  Half-filled chips do not care.
 * -DVS_POLL_SLEEP=n sets both (A/B builds). */
#ifdef VS_POLL_SLEEP
#define VS_POLL_SLEEP_2 VS_POLL_SLEEP
#define VS_POLL_SLEEP_3 VS_POLL_SLEEP
#else
#define VS_POLL_SLEEP_2 32
#define VS_POLL_SLEEP_3 8
#endif
/* (s_sleep takes an immediate: one call site per value) */
template <bool THREE>
__device__ __forceinline__ void vs_poll_sleep()
{
  if (THREE) __builtin_amdgcn_s_sleep(VS_POLL_SLEEP_3);
  else __builtin_amdgcn_s_sleep(VS_POLL_SLEEP_2);
}

/* what every role of a group needs to find its lane, its ring and its progress words */
struct VsGroup {
  const VsDevLane *L;
  int16_t *ring;
  double *ltab;
  int *gpub, *npub;
  VsOrderBox ord;
  long group, row;
  int lane, N, C;
  int ltab_entries; /* doubles reserved for this group's cos rows (the launch's, or the group's own over mixed rings) */
  bool valid;
};

/* The generator role of the two-role kernel (SPLIT = false: whole cycles) and the open-phase role of
 * the three-role kernel (SPLIT = true: jitter, shimmer and both flanks; the cycle's noise leaves as
 * an order, and progress is published by the noise wavefront only). */
template <bool SPLIT>
__device__ __forceinline__ void vs_generator_wave(const VsKernelArgs &args, const VsGroup &g)
{
  const int lane = g.lane, N = g.N, C = g.C;
  VsCfg c;
  VsGen s;
  VsDiag dg;
#ifdef VS_DIAG
#pragma unroll
  for (int k = 0; k < 8; ++k) dg.acc[k] = 0;
  dg.rounds = dg.attend = 0;
  dg.t = vs_stamp();
#endif
  vs_load_cfg(g.L, c, s);
  vs_stage_cos_rows(g.L, c, g.ltab, args.costab, g.ltab_entries, lane, g.valid, args);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); /* own staging writes before own row reads */
  VsRoundKeys rk; /* two roles: this wavefront draws the noise itself (three: the noise wavefront has its own) */
  if (!SPLIT) vs_round_keys(c.key0, c.key1, rk);
  int spins = 0;
  /* A poll that cannot end differently from the last one is cut short: whether a round starts depends
   * on this wavefront's own state (which only a round changes) and on two words per lane written by
   * the others -- npub and, three roles, otak.  While neither has moved the wavefront goes back to
   * sleep after two LDS reads and a compare instead of the ~30 instructions of the full evaluation;
   * an idle open-phase wavefront polls ~2000 times per launch, and every instruction it issues is
   * taken from the two that work (profiles/r03_kernel_experiments.txt). */
  int prev_seen = -1, prev_taken = -1;
  bool dirty = true; /* own state changed since the last full evaluation */
  /* tests (vs_tuning.fault): a generator that never publishes -- the bounded waits of the other
   * wavefronts must run out and reach the caller as VS_ERR_INTERNAL */
  const bool withhold = args.fault == VS_FAULT_WITHHOLD_PROGRESS;
  while (!withhold) {
#ifdef VS_TIMING_GENERATOR_ONLY
    const int n_seen = s.g; /* timing build: the generator alone, never short of room */
#else
    const int n_seen = __hip_atomic_load(&g.npub[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
    int taken = 0;
    if (SPLIT) taken = __hip_atomic_load(&g.ord.otak[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (!dirty && !__any((n_seen != prev_seen) || (taken != prev_taken))) {
      vs_poll_sleep<SPLIT>();
      VS_DIAG_ADD(dg, 6)
      if (++spins > args.spin_limit) {
        if (args.err && lane == 0) atomicOr(args.err, 1);
        break;
      }
      continue;
    }
    prev_seen = n_seen;
    prev_taken = taken;
    dirty = false;
    const bool need = g.valid && (s.g < N);
    if (!__any(need)) break;
    if (need && !s.pend) vs_cycle_scalars(c, s, dg); /* fixes the next period s.T */
    bool want = need && (s.g - n_seen + s.T + VS_TRASH_ROWS <= C);
    /* ... and, three roles, one of the lane's order boxes is free */
    if (SPLIT) want = want && (s.posted - taken < VS_ORDER_DEPTH);
    const bool hungry = want && (s.g - n_seen < args.gen_low); /* its filter would run dry during a round */
    const int n_need = __builtin_popcountll(__ballot(need));
    const int n_want = __builtin_popcountll(__ballot(want));
    if ((n_want > 0) && ((n_want * 64 >= n_need * args.gen_min) || __any(hungry))) {
#ifdef VS_DIAG
      dg.rounds += 1;
      dg.attend += (unsigned long long)n_want;
#endif
      if (SPLIT) {
        if (want) vs_cycle_emit<false, false, true>(c, s, g.ring, C, lane, N, g.ltab, nullptr, 0, dg, nullptr, g.ord);
      } else {
        if (want) vs_cycle_emit<false, true, false>(c, s, g.ring, C, lane, N, g.ltab, nullptr, 0, dg, &g.gpub[lane], VsOrderBox(), &rk);
        VS_LDS_RELEASE();
        __hip_atomic_store(&g.gpub[lane], s.g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      spins = 0;
      dirty = true;
    } else {
      vs_poll_sleep<SPLIT>();
      VS_DIAG_ADD(dg, 6)
      if (++spins > args.spin_limit) {
        if (args.err && lane == 0) atomicOr(args.err, 1);
        break;
      }
    }
  }
#ifdef VS_DIAG
  if (args.diag && lane == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) args.diag[(size_t)g.group * 16 + k] = dg.acc[k];
    args.diag[(size_t)g.group * 16 + 9] = dg.rounds; /* slots 9, 10: unused by the filter wavefront */
    args.diag[(size_t)g.group * 16 + 10] = dg.attend;
  }
#endif
  if (args.ncyc && g.valid) args.ncyc[g.row] = s.cyc;
}

/* The noise role of the three-role kernel: takes the orders of the open-phase wavefront, adds the
 * closed phase's noise (flowgen_shimmer.c:385-406, vs_noise_trips) and is the one that publishes a
 * lane's progress to the filter: the open phase when it takes the order, the closed phase trip by
 * trip.  The round keys of a lane's Philox stream are made once per launch here. */
__device__ __forceinline__ void vs_noise_wave(const VsKernelArgs &args, const VsGroup &g)
{
  const int lane = g.lane, N = g.N, C = g.C;
  VsRoundKeys rk;
  vs_round_keys(g.L->key0, g.L->key1, rk);
  const int dcs = g.L->dcs;
  int taken = 0; /* orders of this lane dealt with */
  int gend = 0;  /* samples of this lane that are complete = where its next cycle starts */
  int wpos = 0;  /* ring slot of sample gend */
  int spins = 0;
#ifdef VS_DIAG
  VsDiag dg;
#pragma unroll
  for (int k = 0; k < 8; ++k) dg.acc[k] = 0;
  dg.rounds = dg.attend = 0;
  dg.t = vs_stamp();
#endif
  for (;;) {
    const bool open = g.valid && (gend < N);
    if (!__any(open)) break;
    const int posted = __hip_atomic_load(&g.ord.oseq[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const bool have = open && (posted != taken);
    if (__any(have)) {
      if (have) {
        const int *box = g.ord.w + (taken & (VS_ORDER_DEPTH - 1)) * (3 * VS_WAVE) + lane;
        const uint32_t d0 = (uint32_t)box[0];
        const int w1 = box[VS_WAVE];
        const int NDW = box[2 * VS_WAVE];
        taken += 1;
        /* the three reads above precede this store in the LDS queue: the box is free again */
        VS_LDS_RELEASE();
        __hip_atomic_store(&g.ord.otak[lane], taken, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const int T3 = w1 & 0xFFFF, T = (int)((unsigned)w1 >> 16);
        const int m = T - T3;
        /* the open phase [0, T3) of this cycle is in the ring (it was before the order was posted) */
        __hip_atomic_store(&g.gpub[lane], gend + T3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (__any(m > 0)) {
          vs_noise_trips<true, true>(g.ring, C, lane, rk, vs_noise_consts(NDW, dcs), d0, m, wpos, T3, gend,
                                     &g.gpub[lane]);
        }
        gend += T;
        wpos += T; /* T <= ring_slots */
        if (wpos >= C) wpos -= C;
      }
      spins = 0;
      VS_DIAG_ADD(dg, 4)
    } else {
      vs_poll_sleep<true>();
      VS_DIAG_ADD(dg, 6)
      if (++spins > args.spin_limit) {
        if (args.err && lane == 0) atomicOr(args.err, 4);
        break;
      }
    }
  }
#ifdef VS_DIAG
  if (args.diag && lane == 0) { /* slots 11, 12: unused by the filter wavefront */
    args.diag[(size_t)g.group * 16 + 11] = dg.acc[4];
    args.diag[(size_t)g.group * 16 + 12] = dg.acc[6];
  }
#endif
}

/* The filter role: super-steps of 24 samples for every lane that holds them (vs_superstep).
 * PARTIAL: groups whose threshold is below 64 lanes run super-steps with the ready lanes only (the
 * two-role kernel); without it every group waits for all of its live lanes (the three-role kernel:
 * the second copy of the window that a partial super-step needs does not fit its 168 registers). */
template <int ARITH, bool PRE1, bool PARTIAL, bool POW>
__device__ __forceinline__ void vs_filter_wave(const VsKernelArgs &args, const VsGroup &g)
{
  const int lane = g.lane, N = g.N, C = g.C;
  const VsDevLane *__restrict__ L = g.L;
  int16_t *ring = g.ring;
  int *gpub = g.gpub, *npub = g.npub;
  const bool valid = g.valid;
  /* When wavefronts share a SIMD (full grids), VALU issue goes to the higher priority first: the
   * filter is the longest of the instruction streams, so it issues as if it were alone and the
   * others fill the slots it leaves.  Without this the hardware prefers the OLDEST wavefront -- the
   * generator -- and the launch takes longer (profiles/r02_ws_full_grid_sweep.txt). */
  if (args.ws_filter_prio >= 3) __builtin_amdgcn_s_setprio(3);
  else if (args.ws_filter_prio == 2) __builtin_amdgcn_s_setprio(2);
  else if (args.ws_filter_prio == 1) __builtin_amdgcn_s_setprio(1);
  /* VS_ARITH_F32: the window and the taps are the packed single-precision ones (VsF32Filter, vs_dev_filter.h); the double
   * ones below are never touched and compile away */
  constexpr bool F32 = (ARITH == VS_ARITH_F32);
  constexpr int DARITH = F32 ? VS_ARITH_FMA : ARITH;
  VsF32Filter f32;
  double a[VS_ORDER + 1];
  if (F32) vs_f32_load(args.taps, L, f32);
  else vs_load_taps<DARITH>(args.taps, L, a);
  const double gain = L->gain;
  const double pre = L->pre;
  int16_t *orow = args.out + g.row * args.out_pitch;
  const int ready_min = (args.ready_min > 0) ? args.ready_min : __builtin_amdgcn_readfirstlane(L->ready_min);
#ifdef VS_DIAG
  VsDiag dg;
#pragma unroll
  for (int k = 0; k < 8; ++k) dg.acc[k] = 0;
  dg.t = vs_stamp();
#endif
  if (!PARTIAL || ready_min >= VS_WAVE) {
    /* Every live lane must be ready (the threshold of deep rings, BASELINE config 3): all lanes
     * of the group then share one position n, the loop is wave-uniform, and the super-step runs
     * under the FULL exec mask -- lanes beyond n_lanes filter whatever their ring column holds
     * and only their stores are masked.  What that buys: the window y[] is updated in place.
     * Under a divergent "if (ready)" the compiler has to keep the old window alive for the
     * lanes that sit out and copies all 24 doubles in and out of every super-step (2 of 55
     * vector instructions per sample). */
    double y[VS_SS];
#pragma unroll
    for (int j = 0; j < VS_SS; ++j) y[j] = 0.0; /* vowel_new.c:222-224 */
    /* A wait that runs out (a protocol bug, or the fault injected by the tests) sets the error word
     * and stops waiting: the remaining super-steps run on whatever the ring holds, the launch ends
     * and vs_plan_status() reports it.  No second way out of the loop -- a "break" here would make
     * the old and the new window meet at the loop latch, and the compiler would copy it again. */
    bool gave_up = false;
    int rslot = 0;
    /* waits until every live lane holds the 24 samples from n on */
    auto await = [&](int n) {
      for (int polls = 0; !gave_up; ++polls) {
#ifdef VS_TIMING_FILTER_ONLY
        const int g_seen = N; /* timing build: the filter alone, never short of input */
#else
        const int g_seen = __hip_atomic_load(&gpub[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
        const bool ready = !valid || (g_seen - n >= VS_SS) || (g_seen >= N);
        if (__all(ready)) break;
        vs_poll_sleep<!PARTIAL>();
        VS_DIAG_ADD(dg, 6)
        if (polls > args.spin_limit) {
          if (args.err && lane == 0) atomicOr(args.err, 2);
          gave_up = true;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    /* the ring reads of the super-step precede this store in the LDS queue: the slots are free */
    auto release = [&](int n_done) {
      rslot += VS_SS;
      if (rslot >= C) rslot = 0;
      VS_LDS_RELEASE();
      __hip_atomic_store(&npub[lane], n_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    /* Lanes beyond n_lanes store to a row of their own that nobody reads, and the super-steps that
     * store sample by sample (the last one of a length that is no multiple of 24; all of them when rows
     * are not 4-byte aligned) run in a loop of their own: no branch is left inside a super-step, and
     * without one the compiler keeps each ring read together with its sign extension (DS_READ_I16
     * instead of DS_READ_U16 + V_BFE_I32: 16 instructions per super-step) and drops the exec masks
     * around the 16-byte stores. */
    if (!valid) orow = args.sink;
    const int n_whole = (args.vec_ok != 0) ? (N / VS_SS) * VS_SS : 0;
    /* POW: vowel -n's frame powers on the way (VsFramePower, vs_dev_filter.h) -- every lane of the launch has frames of
     * args.pow_lframe samples, and the lanes of this wavefront share one position, so where a frame ends is a scalar */
    VsFramePower fp;
    int f_len = 0, f_left = 0; /* length of the frame the next super-step starts in; what is left of it */
    if (POW) {
      fp.sum = fp.done = 0.0f;
      fp.frame = 0;
      fp.bad = fp.requirk = false;
      fp.lanes = args.lanes;
      fp.ondw = args.ondw;
      fp.ondw_pitch = args.ondw_pitch;
      fp.first_lane = g.group * VS_WAVE;
      fp.n_lanes = args.n_lanes;
      f_len = f_left = (args.pow_lframe < N) ? args.pow_lframe : N;
    }
    int n = 0;
    for (; n < n_whole; n += VS_SS) {
      VS_DIAG_ADD(dg, 7)
      await(n);
      int outv[VS_SS];
      vs_u32x4 xpre[VS_SS / 8]; /* only the filter-only kind prefetches */
      if (POW) {
        fp.tb = (f_left <= VS_SS) ? f_left : 0;
        fp.len = f_len;
        fp.force = (args.fault == VS_FAULT_REROUND) && ((n / VS_SS) % 7 == 3);
      }
      if constexpr (F32)
        vs_superstep_f32<PRE1, 1, POW>(f32, ring + rslot * VS_WAVE + lane, orow, n, N, true, &fp);
      else
        vs_superstep<DARITH, VS_KIND_SYNTH, PRE1, true, 1, PARTIAL, POW>(a, y, gain, pre, ring + rslot * VS_WAVE + lane, nullptr, orow, n,
                                                                         N, true, outv, xpre, true, &fp);
      if (POW) {
        if (fp.tb) { /* a frame ended behind sample tb - 1; the rest of the super-step went to the next one's sum */
          vs_frame_power_store(fp, fp.bad || fp.requirk);
          fp.frame += 1;
          fp.bad = fp.requirk;
          const int rest = N - (n + fp.tb);
          f_len = (args.pow_lframe < rest) ? args.pow_lframe : rest;
          f_left = f_len - (VS_SS - fp.tb);
        } else {
          fp.bad = fp.bad || fp.requirk;
          f_left -= VS_SS;
        }
        fp.requirk = false;
      }
      release(n + VS_SS);
      VS_DIAG_ADD(dg, 0)
    }
    /* frames [0, fp.frame) have their width (or NaN) in the table; the one in progress and whatever the sample-by-sample
     * super-steps below complete are the streaming pass's */
    if (POW && valid) args.odone[g.row] = fp.frame;
    for (; n < N; n += VS_SS) {
      VS_DIAG_ADD(dg, 7)
      await(n);
      int outv[VS_SS];
      vs_u32x4 xpre[VS_SS / 8];
      if constexpr (F32)
        vs_superstep_f32<PRE1, 0, false>(f32, ring + rslot * VS_WAVE + lane, orow, n, N, true);
      else
        vs_superstep<DARITH, VS_KIND_SYNTH, PRE1, true, 0>(a, y, gain, pre, ring + rslot * VS_WAVE + lane, nullptr, orow, n,
                                                           N, false, outv, xpre, true);
      release(n + VS_SS);
      VS_DIAG_ADD(dg, 0)
    }
  } else {
    /* shallower rings: a super-step as soon as ready_min/64 of the live lanes hold 24 samples; every
     * lane has its own position n (no frame powers on the way: all of them are the streaming pass's) */
    if (POW && valid) args.odone[g.row] = 0;
    double y[VS_SS];
#pragma unroll
    for (int j = 0; j < VS_SS; ++j) y[j] = 0.0; /* vowel_new.c:222-224 */
    int n = 0, rslot = 0, spins = 0;
    bool live = valid;
    while (__any(live)) {
      VS_DIAG_ADD(dg, 7)
#ifdef VS_TIMING_FILTER_ONLY
      const int g_seen = N;
#else
      const int g_seen = __hip_atomic_load(&gpub[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      const bool ready = live && ((g_seen - n >= VS_SS) || (g_seen >= N));
      const int n_live = __builtin_popcountll(__ballot(live));
      const int n_ready = __builtin_popcountll(__ballot(ready));
      if ((n_ready > 0) && (n_ready * 64 >= n_live * ready_min)) {
        /* Every lane has its own n, so "all 24 samples lie inside the row" is a per-lane question -- but
         * one that only a lane's LAST super-step answers with no (N is no multiple of 24), or every one
         * when rows are not 4-byte aligned.  Asked once per super-step for the whole wavefront it leaves
         * the common case without the 24 per-sample bounds tests and the exec masks around them. */
        const bool inside = (args.vec_ok != 0) && (n + VS_SS <= N);
        if (__all(!ready || inside)) {
          if (ready) {
            int outv[VS_SS];
            vs_u32x4 xpre[VS_SS / 8]; /* only the filter-only kind prefetches */
            if constexpr (F32)
              vs_superstep_f32<PRE1, 1, false>(f32, ring + rslot * VS_WAVE + lane, orow, n, N, true);
            else
              vs_superstep<DARITH, VS_KIND_SYNTH, PRE1, true, 1, PARTIAL>(a, y, gain, pre, ring + rslot * VS_WAVE + lane, nullptr, orow,
                                                                          n, N, true, outv, xpre);
          }
        } else {
          if (ready) {
            int outv[VS_SS];
            vs_u32x4 xpre[VS_SS / 8];
            if constexpr (F32)
              vs_superstep_f32<PRE1, 0, false>(f32, ring + rslot * VS_WAVE + lane, orow, n, N, true);
            else
              vs_superstep<DARITH, VS_KIND_SYNTH, PRE1, true>(a, y, gain, pre, ring + rslot * VS_WAVE + lane, nullptr, orow,
                                                              n, N, args.vec_ok != 0, outv, xpre);
          }
        }
        if (ready) {
          rslot += VS_SS;
          if (rslot >= C) rslot = 0;
          n += VS_SS;
          if (n >= N) live = false;
        }
        /* the ring reads above precede this store in the LDS queue: the slots are free */
        VS_LDS_RELEASE();
        __hip_atomic_store(&npub[lane], n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        spins = 0;
        VS_DIAG_ADD(dg, 0)
      } else {
        vs_poll_sleep<!PARTIAL>();
        VS_DIAG_ADD(dg, 6)
        if (++spins > args.spin_limit) {
          if (args.err && lane == 0) atomicOr(args.err, 2);
          break;
        }
      }
    }
  }
#ifdef VS_DIAG
  if (args.diag && lane == 0) {
    args.diag[(size_t)g.group * 16 + 8] = dg.acc[0];
    args.diag[(size_t)g.group * 16 + 14] = dg.acc[6];
    args.diag[(size_t)g.group * 16 + 15] = dg.acc[7];
  }
#endif
}

/* ROLES wavefronts per group of 64 utterances, args.ws_pairs groups per workgroup, wavefronts laid
 * out role-major: role = wavefront / groups.  Two roles: 0 generator, 1 filter.  Three roles: 0 open
 * phase, 1 noise, 2 filter (the hardware prefers the older of two wavefronts of equal priority:
 * open phase before noise is the better order, tools/ubench/ubench4.hip). */
template <int ARITH, bool PRE1, int ROLES, bool POW>
__device__ __forceinline__ void vs_synth_ws_body(const VsKernelArgs &args)
{
  extern __shared__ __attribute__((aligned(16))) int16_t lds_base[];

  const int widx = (int)threadIdx.x >> 6;
  int ngroups, role, slot;
  if (ROLES == 3 && args.ws_layout == VS_WS_LAYOUT_SPREAD_2X3) {
    /* eight wavefronts: F0 O0 F1 O1 -- N0 -- N1 (vs_device.h); role -1 = nothing to do */
    const int simd = widx & 3, second = widx >> 2;
    ngroups = 2;
    slot = simd >> 1;
    role = (simd & 1) ? (second ? 1 : 0) : (second ? -1 : 2);
  } else {
    ngroups = (int)blockDim.x / (ROLES * VS_WAVE);
    role = widx / ngroups;
    slot = widx - role * ngroups;
  }
  VsGroup g;
  g.lane = (int)threadIdx.x & (VS_WAVE - 1);
  g.group = (long)blockIdx.x * (long)ngroups + slot;
  g.C = args.ring_slots;
  g.ltab_entries = args.ltab_entries;
  g.ring = lds_base + (size_t)slot * (size_t)(args.ws_pair_bytes / sizeof(int16_t));
  if (args.group_map) {
    /* mixed rings (vs_device.h, VsGroupSlot): this slot's group, ring depth and LDS region come from the plan's
     * table -- one scalar load, everything stays wave-uniform */
    const VsGroupSlot gs = args.group_map[(size_t)blockIdx.x * (size_t)ngroups + (size_t)slot];
    g.group = (gs.group >= 0) ? (long)gs.group : (((long)args.n_lanes + VS_WAVE - 1) / VS_WAVE); /* none: beyond the batch */
    g.C = __builtin_amdgcn_readfirstlane(gs.ring_slots);
    g.ring = lds_base + (size_t)__builtin_amdgcn_readfirstlane(gs.lds_off) / sizeof(int16_t);
    g.ltab_entries = __builtin_amdgcn_readfirstlane(gs.ltab_entries);
  }
  const long gl = g.group * VS_WAVE + g.lane;
  g.valid = gl < (long)args.n_lanes;
  g.L = args.lanes + (g.valid ? gl : (long)args.n_lanes - 1);
  g.N = args.n_samples;
  g.ltab = (double *)(g.ring + (size_t)(g.C + VS_TRASH_ROWS) * VS_WAVE);
  g.gpub = (int *)(g.ltab + g.ltab_entries);
  g.npub = g.gpub + VS_WAVE;
  g.ord.oseq = g.npub + VS_WAVE;
  g.ord.otak = g.ord.oseq + VS_WAVE;
  g.ord.w = g.ord.otak + VS_WAVE;
  g.row = (long)g.L->row;

  if (role == 0) {
    g.gpub[g.lane] = 0;
    g.npub[g.lane] = 0;
    if (ROLES == 3) {
      g.ord.oseq[g.lane] = 0;
      g.ord.otak[g.lane] = 0;
    }
  }
  __syncthreads();
  if (role < 0) return; /* the two spare wavefronts of the spread layout */
  /* a slot without a group (the last workgroup of a grid that is no multiple of its groups; an empty slot of the
   * mixed-rings table): nothing to synthesise -- its filter wavefront used to run all N samples into the sink row */
  if (g.group * VS_WAVE >= (long)args.n_lanes) return;

#ifdef VS_TIMING_GENERATOR_ONLY
  if (role == ROLES - 1) return;
#endif
#ifdef VS_TIMING_FILTER_ONLY
  if (role != ROLES - 1) return;
#endif
  if (role == ROLES - 1) vs_filter_wave<ARITH, PRE1, ROLES == 2, POW>(args, g);
  else if (ROLES == 3 && role == 1) vs_noise_wave(args, g);
  else vs_generator_wave<ROLES == 3>(args, g);
}
template <int ARITH, bool PRE1, int ROLES>
__global__ void __launch_bounds__(ROLES * 4 * VS_WAVE) vs_synth_ws_kernel(VsKernelArgs args)
{
  vs_synth_ws_body<ARITH, PRE1, ROLES, false>(args);
}
/* the same with vowel -n's frame powers taken along by the filter wavefronts (VsFramePower, vs_dev_filter.h) */
template <int ARITH, bool PRE1, int ROLES>
__global__ void __launch_bounds__(ROLES * 4 * VS_WAVE) vs_synth_ws_pow_kernel(VsKernelArgs args)
{
  vs_synth_ws_body<ARITH, PRE1, ROLES, true>(args);
}

/*
 * vowel -n (reference vowel_new.c:302-324): white noise added to the filtered signal, frame by frame
 * (Lframe = 800 samples at 16 kHz, 1100 at 22.05 kHz, vw:361-363):
 *     aux = 0; for every sample of the frame: aux += (float)y[i]*y[i];                (float, in sample order)
 *     sig_power = aux / (float)ni;  NoiseDistWidth = sqrt(12*sig_power/snr);          (float)
 *     noiseval = (1.0*random())/RAND_MAX;  aux = NoiseDistWidth*(noiseval - 0.5);     (float <- double)
 *     y[i] = round2int(1.0*y[i] + 1.0*aux);
 * The power of a WHOLE frame stands in front of its first noise sample, so this is two streaming passes over the
 * finished PCM, behind whatever kernel wrote it (the fused wave-specialised kernels, the one-wave kernel, the wide
 * filter kernel -- none of them knows about frames):
 *   vs_out_power_kernel  one thread per (utterance, frame): the sequential float sum, then NoiseDistWidth -> ondw[row][frame];
 *   vs_out_noise_kernel  one thread per 8 samples: the vowel process draws once per sample, in order, so draw n belongs
 *                        to sample n -- two Philox blocks per thread, every sample independent.
 * 2 B/sample read + 4 B/sample read and written, only when asked for.
 */
/* aux += (float)y*y over the 8 samples of 16 bytes of a row, in sample order (vowel_new.c:304-306) */
__device__ __forceinline__ void vs_power8(const vs_u32x4 v, float &aux)
{
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float lo = (float)(int)(int16_t)(v[e] & 0xFFFFu), hi = (float)((int)v[e] >> 16);
    aux += lo * lo;
    aux += hi * hi;
  }
}

/* NoiseDistWidth of frame fr of the utterance of record L (row `row`), or of nothing (NaN is never read) behind its end */
template <bool VEC>
__device__ __forceinline__ float vs_frame_width(const VsKernelArgs &args, const VsDevLane *__restrict__ L, long row, int fr, float snr)
{
  const int N = args.n_samples;
  const int Lframe = L->Lframe;
  const long f0 = (long)fr * Lframe;
  if (f0 >= N) return __uint_as_float(VS_ONDW_UNKNOWN);
  const int left = N - (int)f0;
  const int ni = (left < Lframe) ? left : Lframe;
  const int16_t *__restrict__ fp = args.out + row * args.out_pitch + f0;
  float aux = 0.0f;
  int i = 0;
  if (VEC) {
    /* frames start on even samples (Lframe is a multiple of 100) of rows that start on 4-byte boundaries: 16-byte
     * loads, eight of them -- one 128-byte line of this thread's frame -- in flight before the first is used, so a
     * line is asked for once although 64 threads of a wavefront read 64 different frames */
    const vs_u32x4 *__restrict__ vp = (const vs_u32x4 *)fp;
    const int nvec = ni >> 3;
    int c = 0;
    for (; c + 8 <= nvec; c += 8) {
      vs_u32x4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = vp[c + j];
#pragma unroll
      for (int j = 0; j < 8; ++j) vs_power8(v[j], aux);
    }
    for (; c < nvec; ++c) vs_power8(vp[c], aux);
    i = nvec << 3;
  }
  for (; i < ni; ++i) {
    const float f = (float)fp[i];
    aux += f * f;
  }
  return vs_noise_width(aux, ni, snr);
}

/* every frame: one thread per (utterance, frame) */
template <bool VEC>
__global__ void __launch_bounds__(256) vs_out_power_kernel(VsKernelArgs args)
{
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  const long l = gid / args.ondw_pitch;
  if (l >= (long)args.n_lanes) return;
  const int fr = (int)(gid - l * args.ondw_pitch);
  const VsDevLane *__restrict__ L = args.lanes + l;
  const float snr = L->out_snr;
  if (!(snr > 0.0f)) return;
  const long row = (long)L->row;
  args.ondw[row * args.ondw_pitch + fr] = vs_frame_width<VEC>(args, L, row, fr, snr);
}

/* Behind a fused kernel that took the frame powers along (vs_synth_ws_pow_kernel): one thread per utterance looks through
 * its row of the table and does what that kernel left -- the frames behind its last whole super-step (args.odone[row] says
 * where they start) and the ones it marked unknown because the quirk path of round2int() ran during them (VsFramePower). */
template <bool VEC>
__global__ void __launch_bounds__(64) vs_out_power_fill_kernel(VsKernelArgs args)
{
  const long l = (long)blockIdx.x * 64 + threadIdx.x;
  if (l >= (long)args.n_lanes) return;
  const VsDevLane *__restrict__ L = args.lanes + l;
  const float snr = L->out_snr;
  if (!(snr > 0.0f)) return;
  const long row = (long)L->row;
  float *__restrict__ wrow = args.ondw + row * args.ondw_pitch;
  const int done = args.odone[row];
  const int frames = (int)(((long)args.n_samples + L->Lframe - 1) / L->Lframe);
  for (int fr = 0; fr < frames; ++fr)
    if (fr >= done || __float_as_uint(wrow[fr]) == VS_ONDW_UNKNOWN) wrow[fr] = vs_frame_width<VEC>(args, L, row, fr, snr);
}

/* One noise sample, literally (vowel_new.c:315-317); r = the draw, what random() returns. */
__device__ __forceinline__ int vs_onoise_literal(int y, uint32_t r, float ndw)
{
  const float noiseval = (float)vs_unit_of_draw(r);
  const float aux = (float)((double)ndw * ((double)noiseval - 0.5));
  return vs_round2int(1.0 * (double)y + 1.0 * (double)aux);
}
/* ... and as the noise kernel computes it, 19 vector instructions per sample instead of 34 (and the kernel is bound by them:
 * Philox alone is 9.4).
 *   noiseval: (float)((1.0*r)/RAND_MAX) is the float next to r/(2^31 - 1) = r*2^-31*(1 + 2^-31 + ...): an integer r of
 *     up to 31 bits rounded to 24, where the tail only ever decides an exact tie (upwards).  r*inv, inv = RN(1/RAND_MAX),
 *     is within an ulp of the quotient -- 2^-52 relative, where no integer r lies closer than 2^-31 relative to a point
 *     that rounds the other way -- so the two correction steps of vs_unit_of_draw() cannot matter behind the
 *     conversion (device self-test [7]: all 2^31 draws).
 *   aux: noiseval - 0.5 is exact in double, so RN(NoiseDistWidth * (noiseval - 0.5)) is ONE fused multiply-add,
 *     fma(NoiseDistWidth, noiseval, -NoiseDistWidth/2): the same exact quantity, rounded once, to double; then to float.
 *   rounding: y is an integer, so round2int(y + aux) = clamp(y + ceil(aux - 0.5)) for every float aux outside
 *     (-2^-38, 0): where the double sum y + aux rounds at all (|aux| < 2^-14) it stays strictly between y - 1 and y + 1 on
 *     aux's side of y, at least 2^-38 - 2^-40 away from the integers the reference's "x + 1" could round up to.  Inside
 *     that interval round2int has its quirks (x + 1 rounds to an integer: vs_dev_primitives.h).  A nonzero aux is at
 *     least NoiseDistWidth * 2^-25 in magnitude (the floats next to one half), so a frame whose width is 0 or >= 2^-13
 *     never meets the interval: the test is per FRAME, not per sample.  ceil(aux - 0.5) = -floor(-aux + 0.5) is ONE
 *     instruction on gfx950, V_CVT_RPI_I32_F32.
 *   clamp: widths below 65534 keep |aux| <= 32767, the rounded value fits 16 bits, and two samples are subtracted,
 *     saturated and lifted from -32768 to the reference's -32767 as a pair (V_CVT_PK_I16_I32, V_PK_SUB_I16 clamp,
 *     V_PK_MAX_I16) straight from the packed words of the row.
 * Frames outside [2^-13, 65534) (an SNR beyond ~90 dB, or below 0 dB on a clipped signal: the ABI takes any out_snr > 0)
 * take the literal form.  Self-test [7]: every float outside (-2^-38, 0) at nine values of y. */
__device__ __forceinline__ bool vs_onoise_width_is_plain(float ndw) { return (ndw == 0.0f) || (ndw >= 0x1p-13f && ndw < 65534.0f); }
__device__ __forceinline__ bool vs_onoise_aux_is_plain(float aux)
{
  return (__float_as_uint(aux) - 0x80000001u) > 0x2C7FFFFEu; /* not in (-2^-38, -0) */
}
/* floor(-aux + 0.5) = -ceil(aux - 0.5) */
__device__ __forceinline__ int vs_onoise_neg_round(float aux)
{
  int r;
  asm("v_cvt_rpi_i32_f32_e64 %0, -%1" : "=v"(r) : "v"(aux));
  return r;
}
__device__ __forceinline__ int vs_onoise_neg_round_of_draw(uint32_t o, double ndw, double neg_half_ndw)
{
  const float noiseval = (float)((double)(o >> 1) * 0x1.00000002p-31);
  return vs_onoise_neg_round((float)__builtin_fma(ndw, (double)noiseval, neg_half_ndw));
}
/* {y.lo - r0, y.hi - r1}, each saturated to int16 and lifted to >= -32767; |r0|, |r1| <= 32767 */
__device__ __forceinline__ uint32_t vs_onoise_sub2(uint32_t ypair, int r0, int r1)
{
  const vs_i16x2 r = __builtin_amdgcn_cvt_pk_i16(r0, r1);
  uint32_t d;
  asm("v_pk_sub_i16 %0, %1, %2 clamp" : "=v"(d) : "v"(ypair), "v"(__builtin_bit_cast(uint32_t, r)));
  const vs_i16x2 floor_ = {(short)-32767, (short)-32767};
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(vs_i16x2, d), floor_));
}

/* VS_ONOISE_OCTETS octets of 8 samples per thread, 256 threads apart (measurement builds: -DVS_ONOISE_OCTETS=1 ...) */
#ifndef VS_ONOISE_OCTETS
#define VS_ONOISE_OCTETS 1
#endif
#ifndef VS_ONOISE_ROWMAJOR
#define VS_ONOISE_ROWMAJOR 1
#endif
template <bool VEC>
__global__ void __launch_bounds__(256) vs_out_noise_kernel(VsKernelArgs args, unsigned segs, unsigned seg_magic, unsigned seg_shift)
{
  constexpr int OCT = VS_ONOISE_OCTETS;
  /* one workgroup = 256 * OCT octets of ONE utterance: its key, frame length and row are wave-uniform */
#if VS_ONOISE_ROWMAJOR
  /* consecutive workgroups walk along a row: utterance = block / segs by the launcher's multiplier (scalar) */
  const unsigned lane_idx = (segs == 1u) ? blockIdx.x : ((unsigned)(((unsigned long long)blockIdx.x * seg_magic) >> 32) >> seg_shift);
  const unsigned seg = blockIdx.x - lane_idx * segs;
#else
  const unsigned lane_idx = blockIdx.x, seg = blockIdx.y;
#endif
  const VsDevLane *__restrict__ L = args.lanes + lane_idx;
  const float snr = L->out_snr;
  if (!(snr > 0.0f)) return;
  const int N = args.n_samples;
  const int base = (int)((seg * (256u * OCT) + threadIdx.x) * 8u);
  if (base >= N) return;
  const long row = (long)L->row;
  int16_t *__restrict__ orow = args.out + row * args.out_pitch;
  const float *__restrict__ wrow = args.ondw + row * args.ondw_pitch;
  const uint32_t magic = L->lframe_magic;
  const int sh = 31 - __builtin_clz((unsigned)(L->Lframe - 1)); /* ceil(log2(Lframe)) - 1 */
  const uint32_t k0 = L->okey0, k1 = L->okey1;
  vs_u32x4 yp[OCT]; /* 8 samples each, packed as they lie in the row */
  float w0[OCT], w1[OCT];
#pragma unroll
  for (int c = 0; c < OCT; ++c) {
    const int i0 = base + c * 2048;
    if (i0 >= N) continue;
    const int16_t *op = orow + i0;
    if (VEC && (i0 + 8 <= N)) {
      yp[c] = *(const vs_u32x4 *)op;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        yp[c][e] = vs_pack16((i0 + 2 * e < N) ? (int)op[2 * e] : 0, (i0 + 2 * e + 1 < N) ? (int)op[2 * e + 1] : 0);
    }
    /* the frames of the two quads (Lframe is a multiple of 4, not of 8): i / Lframe by the record's multiplier */
    const int fr0 = (int)(__umulhi((uint32_t)i0, magic) >> sh);
    const int fr1 = (i0 + 4 < N) ? (int)(__umulhi((uint32_t)(i0 + 4), magic) >> sh) : fr0;
    w0[c] = wrow[fr0];
    w1[c] = wrow[fr1];
  }
#pragma unroll
  for (int c = 0; c < OCT; ++c) {
    const int i0 = base + c * 2048;
    if (i0 >= N) continue;
    int16_t *op = orow + i0;
    uint32_t o[8];
    vs_philox((uint32_t)(i0 >> 2), k0, k1, o[0], o[1], o[2], o[3]);
    vs_philox((uint32_t)(i0 >> 2) + 1u, k0, k1, o[4], o[5], o[6], o[7]);
    vs_u32x4 v = yp[c];
    if (vs_onoise_width_is_plain(w0[c]) && vs_onoise_width_is_plain(w1[c])) {
      const double d0 = (double)w0[c], d1 = (double)w1[c], h0 = -0.5 * d0, h1 = -0.5 * d1;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r0 = vs_onoise_neg_round_of_draw(o[2 * e], (e < 2) ? d0 : d1, (e < 2) ? h0 : h1);
        const int r1 = vs_onoise_neg_round_of_draw(o[2 * e + 1], (e < 2) ? d0 : d1, (e < 2) ? h0 : h1);
        v[e] = vs_onoise_sub2(v[e], r0, r1);
      }
    } else {
      int y[8];
      vs_unpack8(v, y);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        v[e] = vs_clamp_pack16(vs_onoise_literal(y[2 * e], o[2 * e] >> 1, (e < 2) ? w0[c] : w1[c]),
                               vs_onoise_literal(y[2 * e + 1], o[2 * e + 1] >> 1, (e < 2) ? w0[c] : w1[c]));
    }
    if (VEC && (i0 + 8 <= N)) {
      *(vs_u32x4 *)op = v;
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (i0 + k < N) op[k] = (int16_t)((k & 1) ? (v[k >> 1] >> 16) : (v[k >> 1] & 0xFFFFu));
    }
  }
}

extern "C" hipError_t vs_launch_out_noise(const VsKernelArgs *args, hipStream_t stream)
{
  if (!args->ondw || args->ondw_pitch <= 0) return hipErrorInvalidValue;
  const long frames = (long)args->n_lanes * args->ondw_pitch;
  const long pblocks = (frames + 255) / 256;
  const long per_block = 2048L * VS_ONOISE_OCTETS; /* 256 threads x 8 samples x octets */
  const long segs = ((long)args->n_samples + per_block - 1) / per_block;
  if (pblocks > 0x7FFFFFFFL) return hipErrorInvalidValue;
#if VS_ONOISE_ROWMAJOR
  const long nblocks = segs * (long)args->n_lanes;
  if (nblocks > 0x7FFFFFFFL) return hipErrorInvalidValue;
  /* block / segs for every block < 2^31: the multiplier of vs_lframe_magic (csrc/vs_planhost.c); segs = 1: the block itself */
  unsigned k = 0;
  while ((1ull << k) < (unsigned long long)segs) k++;
  const unsigned seg_magic = segs > 1 ? (unsigned)(((1ull << (31 + k)) / (unsigned long long)segs) + 1ull) : 0u;
  const unsigned seg_shift = segs > 1 ? k - 1 : 0u;
  const dim3 grid((unsigned)nblocks);
#else
  if (segs > 65535L) return hipErrorInvalidValue;
  const unsigned seg_magic = 0, seg_shift = 0;
  const dim3 grid((unsigned)args->n_lanes, (unsigned)segs);
#endif
  const bool fill = args->odone != nullptr; /* the fused kernel took the frame powers along */
  const dim3 fgrid((unsigned)((args->n_lanes + 63) / 64));
  if (args->vec_ok) {
    if (fill) hipLaunchKernelGGL(vs_out_power_fill_kernel<true>, fgrid, dim3(64), 0, stream, *args);
    else hipLaunchKernelGGL(vs_out_power_kernel<true>, dim3((unsigned)pblocks), dim3(256), 0, stream, *args);
    hipLaunchKernelGGL(vs_out_noise_kernel<true>, grid, dim3(256), 0, stream, *args, (unsigned)segs, seg_magic, seg_shift);
  } else {
    if (fill) hipLaunchKernelGGL(vs_out_power_fill_kernel<false>, fgrid, dim3(64), 0, stream, *args);
    else hipLaunchKernelGGL(vs_out_power_kernel<false>, dim3((unsigned)pblocks), dim3(256), 0, stream, *args);
    hipLaunchKernelGGL(vs_out_noise_kernel<false>, grid, dim3(256), 0, stream, *args, (unsigned)segs, seg_magic, seg_shift);
  }
  return hipGetLastError();
}

/*
 * Wide filter kernel: explicit coefficient sets of 23..40 taps (MAX_ORDER of vowel_new.c:33).
 * The same recurrence as vs_superstep -- vowel_new.c:266-289 with a larger Order -- on a register
 * window of 48 doubles and 40 coefficients per lane; the input is a flow row in HBM (the source
 * kernel wrote it, or the caller supplied it), so this path is NOT fused.  Lanes of lower order
 * carry zeros in the missing taps (acc - 0*y == acc).  One lane per thread, 64-thread workgroups.
 */
template <int ARITH>
__global__ void __launch_bounds__(VS_WAVE) vs_filter_wide_kernel(VsKernelArgs args)
{
  const int lane = (int)threadIdx.x;
  const long gl = (long)blockIdx.x * VS_WAVE + lane;
  if (gl >= (long)args.n_lanes) return;
  const VsDevLane *__restrict__ L = args.lanes + gl;
  const int N = args.n_samples;
  double a[VS_WIDE_ORDER + 1];
  double y[VS_WIDE_SS];
  a[0] = 1.0;
  const double *__restrict__ aw = args.awide + gl * VS_WIDE_ORDER;
#pragma unroll
  for (int j = 1; j <= VS_WIDE_ORDER; ++j) a[j] = aw[j - 1];
#pragma unroll
  for (int j = 0; j < VS_WIDE_SS; ++j) y[j] = 0.0; /* vowel_new.c:222-224 */
  const double gain = L->gain;
  const double pre = L->pre;
  const long row = (long)L->row;
  const int16_t *__restrict__ irow = args.in + row * args.in_pitch;
  int16_t *__restrict__ orow = args.out + row * args.out_pitch;
  const bool vec = args.vec_ok != 0;

  for (int n = 0; n < N; n += VS_WIDE_SS) {
    const bool whole = vec && (n + VS_WIDE_SS <= N);
#pragma unroll
    for (int g = 0; g < VS_WIDE_SS / 8; ++g) {
      int xin[8];
      if (whole) {
        const vs_u32x4 v = *(const vs_u32x4 *)(irow + n + 8 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          xin[2 * e] = (int)(int16_t)(v[e] & 0xFFFFu);
          xin[2 * e + 1] = (int)(int16_t)(v[e] >> 16);
        }
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) xin[k] = (n + 8 * g + k < N) ? (int)irow[n + 8 * g + k] : 0;
      }
      int outv[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int t = 8 * g + k;
        /* y_double[0] = 0.0 + B[0]*x[i]*gain, B = {1, 0, ...} (vowel_new.c:266-269) */
        double acc = (double)xin[k] * gain;
        const double y1 = y[(t + VS_WIDE_SS - 1) % VS_WIDE_SS];
        if (ARITH == VS_ARITH_EXACT) {
#pragma unroll
          for (int j = 1; j <= VS_WIDE_ORDER; ++j) acc = acc - a[j] * y[(t + VS_WIDE_SS - j) % VS_WIDE_SS];
        } else {
          double p0 = acc, p1 = -(a[2] * y[(t + VS_WIDE_SS - 2) % VS_WIDE_SS]);
#pragma unroll
          for (int j = 3; j <= VS_WIDE_ORDER; ++j) {
            const double yj = y[(t + VS_WIDE_SS - j) % VS_WIDE_SS];
            if (j & 1) p0 = __builtin_fma(-a[j], yj, p0);
            else p1 = __builtin_fma(-a[j], yj, p1);
          }
          acc = __builtin_fma(-a[1], y1, p0 + p1);
        }
        const double o = (ARITH == VS_ARITH_EXACT) ? (acc - pre * y1) : __builtin_fma(-pre, y1, acc);
        outv[k] = vs_round2int(o); /* vowel_new.c:284 */
        y[t] = acc;                /* the window rotates by renaming, vowel_new.c:287-289 */
        __builtin_amdgcn_sched_barrier(0);
      }
      if (whole) {
        vs_u32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          v[e] = ((uint32_t)outv[2 * e] & 0xFFFFu) | ((uint32_t)outv[2 * e + 1] << 16);
        *(vs_u32x4 *)(orow + n + 8 * g) = v;
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (n + 8 * g + k < N) orow[n + 8 * g + k] = (int16_t)outv[k];
      }
    }
  }
}

extern "C" hipError_t vs_launch_filter_wide(int arith, const VsKernelArgs *args, unsigned grid, hipStream_t stream)
{
  if (!args->awide || !args->in) return hipErrorInvalidValue;
  if (arith == VS_ARITH_EXACT)
    hipLaunchKernelGGL(vs_filter_wide_kernel<VS_ARITH_EXACT>, dim3(grid), dim3(VS_WAVE), 0, stream, *args);
  else /* (VS_ARITH_F32 as well: the single-precision filter is the wave-specialised kernels' alone) */
    hipLaunchKernelGGL(vs_filter_wide_kernel<VS_ARITH_FMA>, dim3(grid), dim3(VS_WAVE), 0, stream, *args);
  return hipGetLastError();
}

/*
 * Device self-test (vs_ctx_selftest): the shortcuts this file takes instead of the reference's
 * library calls are checked against the straightforward form ON THE DEVICE.
 *   [0] vs_unit_of_draw(r) == (double)r / 2147483647.0 (the compiler's IEEE division) for ALL
 *       2^31 possible draws;
 *   [1] Philox4x32-10 known answers (Random123 kat_vectors) and the counter/key wiring;
 *   [2] vs_isqrt_floor(v) == floor(sqrt(v)) for float-valued v on a grid that includes every
 *       perfect square up to 2^24 and its two float neighbours;
 *   [3] vs_round2int(x) against a literal transcription of vowel_new.c:413-427 on a grid around
 *       every half-integer and the clamp edges, tiny negative values and the neighbours of -0.5;
 *   [4] the one-fma noise sample (vs_noise_sample) against the reference's
 *       (short)DC + (short)ceil((r/RAND_MAX)*N - N/2.) for ALL 2^31 draws at 16 (width, DC) pairs,
 *       and for every width 0..VS_NDW_FAST at the edge draws;
 *   [5] vs_philox2 (prepared round keys, two blocks) against vs_philox;
 *   [7] the output-noise sample (vowel -n, vs_out_noise_kernel): (float)(r*inv) against (float)((1.0*r)/RAND_MAX) for ALL
 *       2^31 draws, and y - V_CVT_RPI_I32_F32(-aux), clamped, against round2int(1.0*y + 1.0*aux) for EVERY float aux
 *       (all 2^32 bit patterns but the NaNs and (-2^-38, -0)) at nine y, the packed 16-bit form wherever |aux| <= 32767.
 * bad[k] counts failures of check k ([6] is the host's: the wave-to-SIMD probe).
 */
__device__ __forceinline__ int vs_round2int_literal(double x)
{
  double dec = x - floor(x);
  if (dec > 0.5) x = x + 1;
  if (x > 32767) x = 32767;
  else if (x < -32767) x = -32767;
  return (int)(int16_t)(int)floor(x);
}
/* both forms of the super-step's rounding against the literal one: vs_round2int() always, the
 * half-down form wherever the super-step would not fall back (vs_superstep); returns the failures */
__device__ __forceinline__ int vs_round2int_check(double x)
{
  const int want = vs_round2int_literal(x);
  int bad = (vs_round2int(x) != want) ? 1 : 0;
  const bool flagged = (__double2hiint(x) <= VS_R2I_Q1_HI) || ((uint32_t)__double2loint(x) == 0xFFFFFFFFu);
  if (!flagged && vs_round2int_half_down(x) != want) bad += 1;
  return bad;
}

/* flowgen_shimmer.c:387, 398, literally */
__device__ __forceinline__ int vs_noise_w_literal(uint32_t r, int N)
{
  return (int)(int16_t)(int)ceil(((1.0 * (double)r) / 2147483647.0) * (double)N - (double)N / 2.);
}

__global__ void __launch_bounds__(256) vs_selftest_kernel(unsigned long long *bad)
{
  const unsigned long long tid = (unsigned long long)blockIdx.x * 256ull + threadIdx.x;
  const unsigned long long nthreads = (unsigned long long)gridDim.x * 256ull;
  unsigned long long b0 = 0, b2 = 0, b3 = 0, b4 = 0, b5 = 0, b7 = 0;
  for (unsigned long long r = tid; r < (1ull << 31); r += nthreads) {
    const double ref = (1.0 * (double)(uint32_t)r) / 2147483647.0;
    if (vs_unit_of_draw((uint32_t)r) != ref) b0++;
  }
  for (unsigned long long k = tid; k < (1ull << 24); k += nthreads) {
    const float sq = (float)((double)k * (double)k);
    const uint32_t sb = __float_as_uint(sq); /* sq >= 0: neighbours are the adjacent bit patterns */
    const float cand[3] = {sq, __uint_as_float(sb > 0u ? sb - 1u : 0u), __uint_as_float(sb + 1u)};
    for (int j = 0; j < 3; ++j) {
      const double v = (double)cand[j];
      long long want = (long long)k - 2;
      if (want < 0) want = 0;
      while ((double)(want + 1) * (double)(want + 1) <= v) ++want; /* floor(sqrt(v)) by definition */
      if ((long long)vs_isqrt_floor(v) != want) b2++;
    }
  }
  for (unsigned long long k = tid; k < 140000ull * 64ull; k += nthreads) {
    const int base = (int)(k / 64ull) - 70000;             /* integers -70000 .. 69999 */
    const int j = (int)(k % 64ull);
    const double frac = (j < 32) ? 0.5 + (double)(j - 16) * 0x1p-50 : (double)(j - 32) / 32.0;
    const double x = (double)base + frac;
    b3 += vs_round2int_check(x);
    /* the doubles around the integer itself, 16 each way (the largest double below a power of
     * two is where x + 1 rounds up to the next integer) */
    const double xi = __longlong_as_double(__double_as_longlong((double)base) + (long long)(j - 32));
    b3 += vs_round2int_check(xi);
  }
  for (unsigned long long k = tid; k < 4096ull; k += nthreads) {
    /* -2^-e and -0.5 +- j ulps, e = 1..1074: where x - floor(x) rounds */
    const int e = (int)(k % 1075ull);
    const int j = (int)(k / 1075ull);
    const double tiny = -ldexp(1.0, -e) * (1.0 + 0.25 * (double)j);
    const double near = __longlong_as_double(__double_as_longlong(-0.5) + (long long)(e % 9) - 4 + 16 * j);
    b3 += vs_round2int_check(tiny) + vs_round2int_check(near) + vs_round2int_check(-tiny);
    /* the neighbours of +-2^-e: -2^-54 is the last member of the quirk set */
    const double pw = ldexp(1.0, -e);
    for (int d = -2; d <= 2; ++d) {
      const double u = __longlong_as_double(__double_as_longlong(pw) + (long long)d);
      b3 += vs_round2int_check(u) + vs_round2int_check(-u);
    }
  }
  if (tid == 0) {
    /* the quirk set is where the two forms differ, and the super-step's test catches all of it */
    const double q[6] = {-0x1p-54, -0x1p-60, -5e-324, 0x1.fffffffffffffp-1, 0x1.fffffffffffffp+0, 0x1.fffffffffffffp+13};
    for (int i = 0; i < 6; ++i) {
      const bool flagged = (__double2hiint(q[i]) <= VS_R2I_Q1_HI) || ((uint32_t)__double2loint(q[i]) == 0xFFFFFFFFu);
      if (!flagged || vs_round2int(q[i]) != vs_round2int_literal(q[i]) ||
          vs_round2int_half_down(q[i]) + 1 != vs_round2int_literal(q[i]))
        b3++;
    }
  }
  {
    const int widths[16] = {1, 2, 3, 7, 100, 2801, 2802, 4095, 4096, 12345, 32767, 32768, 45001, 65534, 45533, 45534};
    const int dcv[16] = {0, 1, -1, 9830, 0, 0, 3, -3, 0, 0, 16000, -16000, 9830, 0, 9830, -9830};
    for (int wi = 0; wi < 16; ++wi) {
      const int N = widths[wi], dc = dcv[wi];
      const VsNoiseK nk = vs_noise_consts(N, dc);
      for (unsigned long long r = tid; r < (1ull << 31); r += nthreads)
        if ((int)(int16_t)vs_noise_sample(nk, (uint32_t)r) != dc + vs_noise_w_literal((uint32_t)r, N)) b4++;
    }
    const uint32_t edge[12] = {0u, 1u, 2u, 3u, 0x3FFFFFFFu, 0x40000000u, 0x40000001u, 0x7FFFFFFCu, 0x7FFFFFFDu, 0x7FFFFFFEu, 0x7FFFFFFFu, 0x12345678u};
    for (unsigned long long k = tid; k < (unsigned long long)(VS_NDW_FAST + 1) * 12ull; k += nthreads) {
      const int N = (int)(k / 12ull);
      const uint32_t r = edge[k % 12ull];
      const int dc = (N & 1) ? 0 : 7;
      const VsNoiseK nk = vs_noise_consts(N, dc);
      /* widths near the limit leave no room for DC: compare modulo 2^16, as the store does */
      if ((int)(int16_t)vs_noise_sample(nk, r) != (int)(int16_t)(dc + vs_noise_w_literal(r, N))) b4++;
    }
  }
  for (unsigned long long k = tid; k < 65536ull; k += nthreads) {
    const uint32_t blk = (uint32_t)(k * 2654435761ull), k0 = (uint32_t)(k * 40503ull + 1ull), k1 = (uint32_t)(~k * 97ull);
    VsRoundKeys rk;
    vs_round_keys(k0, k1, rk);
    uint32_t o[8], p[8];
    vs_philox2(blk, rk, o);
    vs_philox(blk, k0, k1, p[0], p[1], p[2], p[3]);
    vs_philox(blk + 1u, k0, k1, p[4], p[5], p[6], p[7]);
    for (int w = 0; w < 8; ++w)
      if (o[w] != p[w]) b5++;
  }
  for (unsigned long long r = tid; r < (1ull << 31); r += nthreads) {
    const float want = (float)((1.0 * (double)(uint32_t)r) / 2147483647.0);
    const float got = (float)((double)(uint32_t)r * 0x1.00000002p-31);
    if (got != want) b7++;
  }
  {
    const int ys[9] = {0, 1, -1, 2, 4096, 16384, 32767, -32767, -32768};
    for (unsigned long long k = tid; k < (1ull << 32); k += nthreads) {
      const float aux = __uint_as_float((uint32_t)k);
      if (aux != aux || !vs_onoise_aux_is_plain(aux)) continue;
      const int r = vs_onoise_neg_round(aux);
      const bool narrow = (aux >= -32767.0f) && (aux <= 32767.0f); /* what widths below 65534 give: the packed form */
      for (int j = 0; j < 9; ++j) {
        const int want = vs_round2int_literal(1.0 * (double)ys[j] + 1.0 * (double)aux);
        const long long v = (long long)ys[j] - (long long)r;
        if ((int)((v > 32767) ? 32767 : ((v < -32767) ? -32767 : v)) != want) b7++;
        if (narrow && (int)(int16_t)(vs_onoise_sub2((uint32_t)ys[j] & 0xFFFFu, r, 0) & 0xFFFFu) != want) b7++;
      }
    }
  }
  if (b7) atomicAdd(&bad[7], b7);
  if (b0) atomicAdd(&bad[0], b0);
  if (b2) atomicAdd(&bad[2], b2);
  if (b3) atomicAdd(&bad[3], b3);
  if (b4) atomicAdd(&bad[4], b4);
  if (b5) atomicAdd(&bad[5], b5);
  if (tid == 0) {
    unsigned long long b1 = 0;
    uint32_t o0, o1, o2, o3;
    vs_philox(0u, 0u, 0u, o0, o1, o2, o3);
    if (o0 != 0x6627E8D5u || o1 != 0xE169C58Du || o2 != 0xBC57AC4Cu || o3 != 0x9B00DBD8u) b1++;
    /* counter (n, 0, 0, 0) with a non-trivial key against the host-computed value is covered by
     * every bit-exact parity test; here: the key words are not swapped */
    vs_philox(1u, 2u, 3u, o0, o1, o2, o3);
    uint32_t p0, p1, p2, p3;
    vs_philox(1u, 3u, 2u, p0, p1, p2, p3);
    if (o0 == p0 && o1 == p1) b1++;
    if (b1) atomicAdd(&bad[1], b1);
  }
}

/*
 * Where the hardware puts the wavefronts of a workgroup (vs_ctx_simd_dealing): every wavefront of a workgroup of
 * `blockDim.x / 64` wavefronts that has a CU to itself (the launcher asks for most of the LDS, as the fused launches
 * do) writes its HW_ID register.  The wave-specialised layouts are built on wavefront w running on the SIMD of wavefront w % 4
 * (vs_device.h): correctness does not depend on it, the launch time does (role-major against the wrong order:
 * 2.6 against 6.4 ms, profiles/r04_config4_roles.txt) -- so the plan asks instead of assuming.
 * HW_ID (gfx9 family): wave_id [3:0], simd_id [5:4], pipe_id [7:6], cu_id [11:8], sh_id [12], se_id [15:13].
 */
__global__ void __launch_bounds__(1024) vs_simd_probe_kernel(unsigned *out)
{
  extern __shared__ __attribute__((aligned(16))) int16_t probe_lds[];
  if ((threadIdx.x & (VS_WAVE - 1)) == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4); /* hwreg(HW_REG_HW_ID, 0, 32) */
    out[(size_t)blockIdx.x * 16 + (threadIdx.x >> 6)] = hw | 0x80000000u;     /* bit 31: "written" */
    if (threadIdx.x == 0) probe_lds[0] = (int16_t)hw;                         /* the LDS is really allocated */
  }
  __syncthreads(); /* every wavefront of the workgroup is resident at the same time */
}

/* vs_plan_reseed: record i takes the seeds of the lane it was made from (VsDevLane.row) */
__global__ void __launch_bounds__(256) vs_reseed_kernel(VsDevLane *lanes, const unsigned long long *seeds,
                                                        const unsigned long long *out_seeds, int n_lanes)
{
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)n_lanes) return;
  VsDevLane *L = lanes + i;
  const unsigned long long s = seeds[L->row], o = out_seeds[L->row];
  L->key0 = (uint32_t)s;
  L->key1 = (uint32_t)(s >> 32);
  L->okey0 = (uint32_t)o;
  L->okey1 = (uint32_t)(o >> 32);
}

extern "C" hipError_t vs_launch_reseed(VsDevLane *lanes, const unsigned long long *seeds, const unsigned long long *out_seeds,
                                       int n_lanes, hipStream_t stream)
{
  hipLaunchKernelGGL(vs_reseed_kernel, dim3((unsigned)((n_lanes + 255) / 256)), dim3(256), 0, stream, lanes, seeds, out_seeds, n_lanes);
  return hipGetLastError();
}

extern "C" hipError_t vs_launch_simd_probe(int waves, unsigned grid, size_t lds_bytes, unsigned *out_dev, hipStream_t stream)
{
  if (waves < 1 || waves > 16) return hipErrorInvalidValue;
  if (lds_bytes > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void *)vs_simd_probe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(vs_simd_probe_kernel, dim3(grid), dim3((unsigned)waves * VS_WAVE), lds_bytes, stream, out_dev);
  return hipGetLastError();
}

extern "C" hipError_t vs_launch_selftest(unsigned long long *bad_dev, hipStream_t stream)
{
  hipLaunchKernelGGL(vs_selftest_kernel, dim3(4096), dim3(256), 0, stream, bad_dev);
  return hipGetLastError();
}

#endif /* VS_GROUP_LANES == VS_WAVE */

/* ------------------------------------------------------------------------------------------
 * launch table
 * ---------------------------------------------------------------------------------------- */
typedef void (*vs_kernel_fn)(VsKernelArgs);

template <int ARITH, int KIND, bool PRE1>
static vs_kernel_fn vs_pick_log(bool log)
{
  return log ? (vs_kernel_fn)vs_synth_kernel<ARITH, KIND, true, PRE1>
             : (vs_kernel_fn)vs_synth_kernel<ARITH, KIND, false, PRE1>;
}

#if VS_GROUP_LANES == VS_WAVE
template <int ARITH, bool PRE1>
static vs_kernel_fn vs_pick_ws2(bool three, bool pow)
{
  if (three) return pow ? (vs_kernel_fn)vs_synth_ws_pow_kernel<ARITH, PRE1, 3> : (vs_kernel_fn)vs_synth_ws_kernel<ARITH, PRE1, 3>;
  return pow ? (vs_kernel_fn)vs_synth_ws_pow_kernel<ARITH, PRE1, 2> : (vs_kernel_fn)vs_synth_ws_kernel<ARITH, PRE1, 2>;
}
static vs_kernel_fn vs_pick_ws(int arith, bool pre1, bool three, bool pow)
{
  if (arith == VS_ARITH_EXACT) return pre1 ? vs_pick_ws2<VS_ARITH_EXACT, true>(three, pow) : vs_pick_ws2<VS_ARITH_EXACT, false>(three, pow);
  if (arith == VS_ARITH_F32) return pre1 ? vs_pick_ws2<VS_ARITH_F32, true>(three, pow) : vs_pick_ws2<VS_ARITH_F32, false>(three, pow);
  return pre1 ? vs_pick_ws2<VS_ARITH_FMA, true>(three, pow) : vs_pick_ws2<VS_ARITH_FMA, false>(three, pow);
}
extern "C" hipError_t vs_launch_kernel_narrow(int arith, int kind, bool log, bool pre1, const VsKernelArgs *args,
                                              unsigned grid, size_t lds_bytes, hipStream_t stream);
#define VS_LAUNCH_NAME vs_launch_kernel
#else
#define VS_LAUNCH_NAME vs_launch_kernel_narrow_impl
#endif

/* pre1: every lane has pre_emphasis == 1.0 (the plan knows); only the exact filter has a
 * shorter sequence for it, the other kinds share one instantiation */
extern "C" hipError_t VS_LAUNCH_NAME(int arith, int kind, bool log, bool wave_specialised, bool pre1,
                                     const VsKernelArgs *args, unsigned grid, size_t lds_bytes,
                                     hipStream_t stream)
{
  vs_kernel_fn fn = nullptr;
  unsigned block = VS_WAVE;
#if VS_GROUP_LANES == VS_WAVE
  if (args->group_lanes != 0 && args->group_lanes != VS_WAVE) {
    /* periods beyond the 64-column ring: the narrow build of this file (16 utterances per wavefront) */
    return vs_launch_kernel_narrow(arith, kind, log, pre1, args, grid, lds_bytes, stream);
  }
  if (wave_specialised && kind == VS_KIND_SYNTH && !log) {
    const bool three = args->ws_roles == 3;
    /* vowel -n with one frame length for the whole launch: the filter wavefronts take the frame powers along */
    const bool pow = args->ondw && args->odone && args->pow_lframe > 0;
    fn = vs_pick_ws(arith, pre1, three, pow);
    /* lds_bytes arrives as the bytes of ONE group (ring + cos rows + progress words); args->ws_pairs
     * groups share a workgroup, args->ws_roles wavefronts serve each */
    block = (unsigned)args->ws_roles * VS_WAVE * (unsigned)args->ws_pairs;
    if (three && args->ws_layout == VS_WS_LAYOUT_SPREAD_2X3) {
      if (args->ws_pairs != 2) return hipErrorInvalidValue;
      block = 8 * VS_WAVE;
    }
    /* mixed rings (args->group_map): the rings of a workgroup differ in depth and lds_bytes arrives as the largest
     * workgroup's sum */
    if (!args->group_map) lds_bytes = (size_t)args->ws_pair_bytes * (size_t)args->ws_pairs;
    else if (args->ws_pairs != 4 || (three && args->ws_layout != VS_WS_LAYOUT_ROLE_MAJOR)) return hipErrorInvalidValue;
    grid = (grid + (unsigned)args->ws_pairs - 1) / (unsigned)args->ws_pairs;
  }
#endif
  if (!fn) { /* the one-wave kernel */
    if (arith == VS_ARITH_F32) arith = VS_ARITH_FMA; /* only the wave-specialised kernels have the single-precision filter */
    if (arith == VS_ARITH_EXACT) {
      if (kind == VS_KIND_SYNTH) fn = pre1 ? vs_pick_log<VS_ARITH_EXACT, VS_KIND_SYNTH, true>(log) : vs_pick_log<VS_ARITH_EXACT, VS_KIND_SYNTH, false>(log);
      else if (kind == VS_KIND_SOURCE) fn = vs_pick_log<VS_ARITH_EXACT, VS_KIND_SOURCE, false>(log);
      else if (kind == VS_KIND_FILTER) fn = pre1 ? vs_pick_log<VS_ARITH_EXACT, VS_KIND_FILTER, true>(false) : vs_pick_log<VS_ARITH_EXACT, VS_KIND_FILTER, false>(false);
    } else if (arith == VS_ARITH_FMA) {
      if (kind == VS_KIND_SYNTH) fn = vs_pick_log<VS_ARITH_FMA, VS_KIND_SYNTH, false>(log);
      else if (kind == VS_KIND_SOURCE) fn = vs_pick_log<VS_ARITH_EXACT, VS_KIND_SOURCE, false>(log);
      else if (kind == VS_KIND_FILTER) fn = vs_pick_log<VS_ARITH_FMA, VS_KIND_FILTER, false>(false);
    }
  }
  if (!fn) return hipErrorInvalidValue;
  if (kind == VS_KIND_FILTER) lds_bytes = 0;
  if (lds_bytes > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds_bytes);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(fn, dim3(grid), dim3(block), lds_bytes, stream, *args);
  return hipGetLastError();
}

#if VS_GROUP_LANES != VS_WAVE
extern "C" hipError_t vs_launch_kernel_narrow(int arith, int kind, bool log, bool pre1, const VsKernelArgs *args,
                                              unsigned grid, size_t lds_bytes, hipStream_t stream)
{
  return vs_launch_kernel_narrow_impl(arith, kind, log, false, pre1, args, grid, lds_bytes, stream);
}
#endif
