/*
 * vs_dev_generator.h -- the source: one glottal cycle of every lane of a wavefront (flowgen_shimmer.c:246-423)
 * in two halves, vs_cycle_scalars() and vs_cycle_emit(), the closed phase's noise in 8-draw trips
 * (vs_noise_trips) and the order boxes of the three-role kernel
 * Included by vs_kernels.hip only (device code, one translation unit per build: the 64-column build and
 * the narrow one, -DVS_GROUP_LANES=16).
 */
#ifndef VS_DEV_GENERATOR_H
#define VS_DEV_GENERATOR_H

/*
 * One glottal cycle of a lane is produced in two halves, vs_cycle_scalars() and vs_cycle_emit().
 */
/* First half of a cycle: every draw of the cycle except the noise -- the jitter and shimmer
 * recursions with their rejection loops (flowgen_shimmer.c:248-313) and the closing-speed draw
 * (fg:325; the rising half-pulse between them consumes no draws, so the draw index is the same).
 * They fix the cycle's period T, amplitude and Knew -- no ring space is needed yet, so a lane
 * runs them as soon as its previous cycle is written, and the room check for the samples can
 * use the ACTUAL period instead of the worst case 1.2*P.  The Philox block the draws come from
 * lives only inside this function. */
__device__ __forceinline__ void vs_cycle_scalars(const VsCfg &c, VsGen &s, VsDiag &dg)
{
  VS_DIAG_ADD(dg, 7)
#if defined(VS_TIMING_STUB) && (VS_TIMING_STUB & 1)
  /* measurement builds only (tools/insts.sh): what the per-cycle draws and recursions cost in instructions -- wrong samples */
  s.T = c.P;
  s.d += 3u;
  s.K_next = c.K;
  s.amp_next = (float)c.amp;
  s.S_next = 0.0f;
  s.pend = true;
  return;
#endif
  VsBlk blk;
  blk.idx = 0xFFFFFFFFu; blk.b0 = blk.b1 = blk.b2 = blk.b3 = 0u;
  /* ---- jitter: fg:248-291 ---- */
  if (c.flags & VS_DF_JITTER) {
    const float dp1 = s.dp0; /* DeltaPer[1] = DeltaPer[0] */
    for (;;) {
      const uint32_t r = vs_draw(c, s, blk);
      const float J = (float)(((double)r / (2147483647 * 10000.0)) * 40000.0 * (double)c.jitter -
                              2.0 * (double)c.jitter);
      const double Jd = (double)J;
      s.dp0 = (float)((double)dp1 * (2.0 + Jd) / (2.0 - Jd) + 2.0 * (double)c.P * Jd / (2.0 - Jd));
      s.T = vs_short_of(ceil((double)((float)c.P + s.dp0)));
      if (!(((float)s.T > c.t_hi) || ((float)s.T < c.t_lo))) break;
    }
  }

  /* ---- shimmer: fg:293-313 ---- */
  float Amplitude = (float)c.amp;
  float S = 0.0f;
  if (c.flags & VS_DF_SHIMMER) {
    const float ds1 = s.ds0;
    for (;;) {
      const uint32_t r = vs_draw(c, s, blk);
      const float epsilon = (float)r / 2147483648.0f; /* (float)RAND_MAX == 2^31 */
      S = (float)((double)epsilon * 4.0 * (double)c.shimmer - 2.0 * (double)c.shimmer);
      const double Sd = (double)S;
      s.ds0 = (float)((double)ds1 * (2.0 + Sd) / (2.0 - Sd) + 2.0 * (double)c.amp * Sd / (2.0 - Sd));
      Amplitude = (float)c.amp + s.ds0;
      if (!((Amplitude > c.a_hi) || (Amplitude < c.a_lo))) break;
    }
  }
  /* ---- closing speed: fg:325 (one draw per cycle, always) ---- */
  {
    const uint32_t r = vs_draw(c, s, blk);
    s.K_next = (float)((double)c.K * (1.0 + (double)(2.0f * c.Kvar) * (vs_unit_of_draw(r) - 0.5)));
  }
  s.amp_next = Amplitude;
  s.S_next = S;
  s.pend = true;
  VS_DIAG_ADD(dg, 0)
}

/* (float)x*(float)x of flowgen_shimmer.c:376 for an integer sample |x| <= 32767, literally: the conversion is exact
 * (16 bits into 24), the product rounds once.  (The integer form -- (float)(x*x), same value -- costs a V_MUL_I32_I24
 * of the expensive class where V_MUL_F32 is a plain 32-bit operation: tools/ubench/ubench5.) */
__device__ __forceinline__ float vs_sq_f(int x)
{
  const float f = (float)x;
  return f * f;
}

/* Publishing progress through LDS: the LDS performs the operations of ONE wavefront in the order
 * they were issued, so a progress word stored after the data is seen after the data by whoever
 * reads the word first and the data second -- no wait for the data stores to come back is needed,
 * only the compiler must not move the accesses across each other (a fence with workgroup scope
 * would add an s_waitcnt lgkmcnt(0), a full LDS round trip, to every noise trip). */
#define VS_LDS_RELEASE() __atomic_signal_fence(__ATOMIC_SEQ_CST)


/* Largest noise width the short noise sequence takes (see vs_noise_fast()). */
#define VS_NDW_FAST 65534

/*
 * One noise sample of the closed phase (flowgen_shimmer.c:387-389, 398-400):
 *     w    = (signed short)ceil(((1.0*random())/RAND_MAX)*NoiseDistWidth - NoiseDistWidth/2.)
 *     x[i] = truncate((float)x[i] + w)            with x[i] = (short)par.DC on [T3, T)
 * for a draw r in [0, 2^31), a width N <= VS_NDW_FAST and |(short)DC| + N/2 + 2 <= 32767 (no clamp,
 * no wrap), as the LOW 16 BITS of
 *     trunc(fma(r, N*inv, I - N/2 + 1 - 1e-10)),      I = (short)DC + 65536,  inv = 0x1.00000002p-31
 * -- one conversion, one fused multiply-add, one conversion, and the store takes the low half.
 * Why this is the reference's value for every r: with M = 2^31 - 1 (a prime) the exact quantity
 * V = r*N/M - N/2 is an integer only for r = 0 and r = M; for every other r it lies at least
 * 1/(2M) = 2.3e-10 from an integer.  ceil(V) = floor(V + 1 - d) for any 0 < d <= 2.3e-10 then (and
 * for integer V), and adding the integer I makes the argument positive, so truncation is the
 * floor.  The reference's three roundings (quotient, product, difference) stay within N*2^-52 <
 * 1.5e-11 of V; here N*inv is exact (N < 2^21), r*N*inv = r*N/M*(1 - 2^-62), the constant and the
 * fma round once each at magnitude < 2^18 (<= 1.5e-11 each): 2.2e-11 in all, against margins of
 * 1e-10 below and 1.3e-10 above the integer boundaries.  Adding 65536 does not change the low 16
 * bits.  Checked exhaustively over r for a set of N (and DC values), and over all N at the edge
 * draws, by tests/test_noise_shortcut.py (CPU) and by vs_ctx_selftest() on the device.
 */
struct VsNoiseK {
  double c, k2;
};
__device__ __forceinline__ VsNoiseK vs_noise_consts(int NDW, int dcs)
{
  VsNoiseK k;
  k.c = (double)NDW * 0x1.00000002p-31;
  k.k2 = ((double)(dcs + 65536) - (double)NDW / 2.0 + 1.0) - 1e-10;
  return k;
}
/* the sample's int16 value is the low half of the result */
__device__ __forceinline__ int vs_noise_sample(const VsNoiseK &k, uint32_t r)
{
  return (int)__builtin_fma((double)r, k.c, k.k2);
}

/*
 * The noise of one closed phase on the short sequence (T4 == 0, a width the one-fma form is proved
 * for): draw ordinal q = 0..m-1 of the cycle belongs to cycle sample T3 + q, the draws start at
 * index d0 of the lane's stream.  Two Philox blocks (8 draws) per trip; word 0 of the first block
 * has ordinal q0 in -3..0 (the scalar draws of this cycle sit in front of it), so the first trip
 * masks its leading words.  A lane that is done (q0 >= m) sends its trips to the trash rows.
 * The trips follow the Philox blocks, not the ring: a trip's 8 slots start anywhere and may wrap
 * (compare + select per sample); the run itself moves on by 8 slots per trip.  T3 + q0 >= 1:
 * VS_DF_FAST lanes have T2 >= 4.
 * TAIL: the trip in which a lane ends stops AT the end.  Without it that trip runs up to 7 slots
 * into the next cycle, which is fine when the same wavefront writes the next cycle afterwards, and
 * not when another wavefront is already writing it (three-role kernel).
 * wpos: ring slot of cycle sample 0; gbase: the utterance's sample count at cycle sample 0.
 */
template <bool PUB, bool TAIL>
__device__ __forceinline__ void vs_noise_trips(int16_t *ring, int C, int lane, const VsRoundKeys &rk,
                                               const VsNoiseK nk, uint32_t d0, int m, int wpos, int T3,
                                               int gbase, int *gpub_lane)
{
  const uint32_t bfirst = d0 >> 2;
  int q0 = (int)(4u * bfirst - d0);
  uint32_t b = bfirst;
  VsRun8 run = vs_run8(ring, wpos, C, T3 + q0, lane);
  char *const trashA = vs_run8_trash(ring, C, lane).A;
  {
    uint32_t o[8];
    vs_philox2(b, rk, o);
    int xv[8];
#pragma unroll
    for (int w = 0; w < 8; ++w) xv[w] = vs_noise_sample(nk, o[w] >> 1);
    /* words in front of the cycle's first noise draw (q0 + w < 0) go to the trash rows */
    char *A = (m > 0) ? run.A : trashA, *B = (m > 0) ? run.B : trashA;
    uint32_t pw[8];
    vs_wrap_select8(vs_lds_addr(A), vs_lds_addr(B), run.kw, pw);
    const uint32_t trash32 = vs_lds_addr(trashA);
#pragma unroll
    for (int w = 0; w < 3; ++w) pw[w] = (q0 + w >= 0) ? pw[w] : trash32;
    if (TAIL && __any(q0 + 8 > m)) { /* a cycle whose noise ends inside its first trip */
#pragma unroll
      for (int w = 0; w < 8; ++w) pw[w] = (q0 + w < m) ? pw[w] : trash32;
    }
    vs_lds_store16<0>(pw[0], xv[0]); vs_lds_store16<1>(pw[1], xv[1]); vs_lds_store16<2>(pw[2], xv[2]); vs_lds_store16<3>(pw[3], xv[3]);
    vs_lds_store16<4>(pw[4], xv[4]); vs_lds_store16<5>(pw[5], xv[5]); vs_lds_store16<6>(pw[6], xv[6]); vs_lds_store16<7>(pw[7], xv[7]);
    q0 += 8;
    b += 2u;
    vs_run8_advance(run, C);
    if (PUB) {
      const int done = (q0 < m) ? q0 : m;
      VS_LDS_RELEASE();
      __hip_atomic_store(gpub_lane, gbase + T3 + ((done > 0) ? done : 0), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  while (__any(q0 < m)) {
    uint32_t o[8];
    vs_philox2(b, rk, o);
    int xv[8];
#pragma unroll
    for (int w = 0; w < 8; ++w) xv[w] = vs_noise_sample(nk, o[w] >> 1);
    /* a lane that is done sends the trip to the trash rows */
    char *A = (q0 < m) ? run.A : trashA, *B = (q0 < m) ? run.B : trashA;
    /* the eight store addresses first, once (one compare and one select each: the ring may wrap inside
     * the trip), then -- rarely -- the end of the cycle, then the stores */
    uint32_t pw[8];
    vs_wrap_select8(vs_lds_addr(A), vs_lds_addr(B), run.kw, pw);
    /* some lane ends inside this trip (1 <= m - q0 <= 7; as ONE compare: an AND of two makes the compiler
     * turn the lane mask into 0/1 per lane and compare that against zero again): its slots behind the end
     * go to the trash rows too */
    if (TAIL && __any((unsigned)(m - q0 - 1) < 7u)) {
      const uint32_t trash32 = vs_lds_addr(trashA);
#pragma unroll
      for (int w = 0; w < 8; ++w) pw[w] = (q0 + w < m) ? pw[w] : trash32;
    }
    vs_lds_store16<0>(pw[0], xv[0]); vs_lds_store16<1>(pw[1], xv[1]); vs_lds_store16<2>(pw[2], xv[2]); vs_lds_store16<3>(pw[3], xv[3]);
    vs_lds_store16<4>(pw[4], xv[4]); vs_lds_store16<5>(pw[5], xv[5]); vs_lds_store16<6>(pw[6], xv[6]); vs_lds_store16<7>(pw[7], xv[7]);
    q0 += 8;
    b += 2u;
    vs_run8_advance(run, C);
    if (PUB) { /* every trip: the filter may be waiting for exactly these eight samples (and the last trip of a cycle must publish) */
      const int done = (q0 < m) ? q0 : m;
      VS_LDS_RELEASE();
      __hip_atomic_store(gpub_lane, gbase + T3 + done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
}

/*
 * Three-role kernel: what the open-phase wavefront hands to the noise wavefront for one cycle of
 * one lane -- three LDS words per lane and a sequence number.
 *   word 0 = d0                 draw index of the cycle's first noise draw
 *   word 1 = T3 | T << 16       the noise covers cycle samples [T3, T); T3 == T: nothing to add
 *   word 2 = NoiseDistWidth     (<= VS_NDW_FAST)
 *   oseq   = orders posted so far (written last; the LDS keeps it behind the three words)
 * VS_ORDER_DEPTH orders per lane may be outstanding (order k lives in box k % depth): the open-phase
 * wavefront posts the next one only when oseq - otak < depth, otak being the orders the noise
 * wavefront has taken (copied).  With one box the two wavefronts fall into lockstep and lanes that
 * just missed a batch of the noise wavefront sit out the next round (rounds at 58 % attendance
 * instead of 86 %, profiles/r03_kernel_experiments.txt).
 */
struct VsOrderBox {
  int *w, *oseq, *otak; /* w: [VS_ORDER_DEPTH][3][64] ints in LDS; oseq, otak: [64] */
};
__device__ __forceinline__ void vs_post_order(const VsOrderBox &ob, int lane, VsGen &s, uint32_t d0, int T3, int T, int NDW)
{
  int *box = ob.w + (s.posted & (VS_ORDER_DEPTH - 1)) * (3 * VS_WAVE) + lane;
  box[0] = (int)d0;
  box[VS_WAVE] = T3 | (T << 16);
  box[2 * VS_WAVE] = NDW;
  s.posted += 1;
  VS_LDS_RELEASE();
  __hip_atomic_store(&ob.oseq[lane], s.posted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

/*
 * Second half of a cycle, for every lane that is ACTIVE in the EXEC mask (the caller wraps the
 * call in "if (want)"): the samples -- statement-by-statement restatement of
 * flowgen_shimmer.c:317-423 (scalar form: oracle/vs_oracle.c).
 *
 * Two instruction sequences per phase, same results:
 *   - the general one follows the reference sample by sample (float compare against DC, the
 *     (signed short) wrap, stores masked by the end of the cycle);
 *   - the short one runs when every active lane carries VS_DF_FAST (see vs_device.h): eight
 *     samples per trip, integer compare against ceil(DC), no wrap, and stores that may run up
 *     to 7 slots past a phase -- those slots belong to a later phase of the same cycle or to the
 *     next cycle and are written again before the filter may read them (the room check of the
 *     caller leaves 8 spare slots).  One wavefront per SIMD pays ~5.3 ticks per instruction
 *     whatever its type, so instructions are what is being saved.
 * ltab is this wavefront's copy of the cos rows in LDS (rows padded to a multiple of 8 with
 * 1.0), c.tab_off the lane's row in it.
 */
template <bool LOG, bool PUB = false, bool SPLIT = false>
__device__ __forceinline__ void vs_cycle_emit(const VsCfg &c, VsGen &s, int16_t *ring, int C,
                                              int lane, int N, const double *ltab,
                                              vs_cycle_rec *logrow, int log_cap, VsDiag &dg,
                                              int *gpub_lane = nullptr, const VsOrderBox ord = VsOrderBox(),
                                              const VsRoundKeys *keys = nullptr)
{
  VS_DIAG_ADD(dg, 7)
  const float Amplitude = s.amp_next;
  const float S = s.S_next;
  s.pend = false;
  VS_DIAG_ADD(dg, 0)
  const int T = s.T;
  const int T2 = c.T2;
  const int room = N - s.g; /* samples of this cycle that still belong to the utterance */
  const int lim = (T < room) ? T : room; /* samples of this cycle that are emitted */
  const double Ad = (double)Amplitude;
  const double Ah = Ad * 0.5; /* "Amplitude * 0.5 * (...)" evaluates (Amplitude*0.5) first */
  const float dcsf = (float)c.dcs;
  const double *trow = ltab + c.tab_off;
  float psum = 0.0f; /* aux of fg:374-377, accumulated from T4 on */
  int T4 = s.T4;
  /* the short sequences: every active lane proved in range on the host, no per-cycle log */
  const bool fast = !LOG && __all((c.flags & VS_DF_FAST) != 0);
  constexpr bool PREFETCH = !SPLIT;

  /* ---- rising half-pulse: fg:318-324 ---- */
  if (fast) {
    const float dcs2 = dcsf * dcsf;
    /* PREFETCH (a wavefront with a SIMD of its own, or one that does whole cycles): the cos values of
     * trip i+8 are read while trip i computes -- nothing else would hide the LDS round trip.  The
     * open-phase wavefront of the three-role kernel reads them where it needs them instead: it shares
     * its SIMD with two others that issue while it waits, and carrying the next trip's values costs a
     * register copy per value and trip (every instruction counts, see vs_generator_wave). */
    double cv[8];
    if (PREFETCH) {
#pragma unroll
      for (int k = 0; k < 8; ++k) cv[k] = trow[k];
    }
    for (int i = 0; __any(i < T2); i += 8) {
      if (i < T2) {
        double nv[8];
        if (PREFETCH) {
          if (i + 8 < T2) {
#pragma unroll
            for (int k = 0; k < 8; ++k) nv[k] = trow[i + 8 + k];
          }
        } else {
#pragma unroll
          for (int k = 0; k < 8; ++k) cv[k] = trow[i + k];
        }
        int x[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = (int)ceil(Ah * (1.0 - cv[k])); /* pad: cos = 1 -> 0 */
        if (PREFETCH) {
#pragma unroll
          for (int k = 0; k < 8; ++k) cv[k] = nv[k];
        }
        const VsRun8 run = vs_run8(ring, s.wpos, C, i, lane);
        /* monotone flank: if the trip's first sample is not below DC none of it is.  (The stores are
         * written out in both branches: joined behind them, the samples of the common branch would be
         * copied into the registers the rare one leaves them in.) */
        if (__any(x[0] < c.thr)) {
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const bool lt = (x[k] < c.thr) && (i + k < T2); /* if(x[i] < par.DC) { x[i] = par.DC; T4 = i; } */
            x[k] = lt ? c.dcs : x[k];
            T4 = lt ? (i + k) : T4;
            const float acc = psum + vs_sq_f(x[k]);
            psum = lt ? dcs2 : acc;
          }
          vs_run8_store_all(run, x);
          asm volatile("; rising trip with samples below DC"); /* two different tails: nothing to merge again */
        } else {
#pragma unroll
          for (int k = 0; k < 8; ++k) psum += vs_sq_f(x[k]);
          vs_run8_store_all(run, x);
          asm volatile("; rising trip");
        }
      }
    }
  } else {
    const int nE = (T2 < lim) ? T2 : lim; /* rising samples that are emitted */
    int i = 0;
    for (; i + 4 <= nE; i += 4) {
      double v[4];
      int xs0[4];
      float xf0[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = trow[i + k];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = 1.0 - v[k];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = Ah * v[k];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = ceil(v[k]);
#pragma unroll
      for (int k = 0; k < 4; ++k) xs0[k] = vs_short_of(v[k]);
#pragma unroll
      for (int k = 0; k < 4; ++k) xf0[k] = (float)xs0[k];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool lt = xf0[k] < c.DC; /* if(x[i] < par.DC) { x[i] = par.DC; T4 = i; } */
        const int xs = lt ? c.dcs : xs0[k];
        const float xf = lt ? dcsf : xf0[k];
        T4 = lt ? (i + k) : T4;
        /* aux of fg:375 runs over [T4, T3) with the FINAL T4 -- which is a T4 carried over from an
         * earlier cycle when this cycle never goes below DC (the variable is never reset, fg:114):
         * samples in front of it do not count */
        psum = lt ? (xf * xf) : ((i + k >= T4) ? (psum + xf * xf) : psum);
        ring[vs_ring_at(s.wpos, C, i + k, lane)] = (int16_t)xs;
      }
    }
    for (; i < T2; ++i) { /* remainder, and (last cycle of the utterance) samples past the end */
      const int xs0 = vs_short_of(ceil(Ah * (1.0 - trow[i])));
      const float xf0 = (float)xs0;
      const bool lt = xf0 < c.DC;
      const int xs = lt ? c.dcs : xs0;
      const float xf = lt ? dcsf : xf0;
      T4 = lt ? i : T4;
      psum = lt ? (xf * xf) : ((i >= T4) ? (psum + xf * xf) : psum);
      if (i < lim) ring[vs_ring_at(s.wpos, C, i, lane)] = (int16_t)xs;
    }
  }
  s.T4 = T4;

  VS_DIAG_ADD(dg, 1)
  /* ---- closing speed: fg:325, drawn in vs_cycle_scalars ---- */
  const double Kd = (double)s.K_next;

  /* ---- falling half-pulse: fg:327-332 ---- */
  int T3 = 2 * T2;
  {
    bool run = true;
    int kdone = 0; /* falling samples this lane has been through (lanes of a wave may differ in T2) */
    if (fast) {
      /* trips of 8, the last one possibly partial (the cos rows are padded to a multiple of 8); the
       * flank falls monotonically, so a whole trip whose last sample is not below DC holds no break.
       * Stores are unconditional: slots at and behind the break, and behind the flank, are written
       * again by the closed phase (2*T2 + 8 <= T). */
      double cv[8];
      if (PREFETCH) {
#pragma unroll
        for (int k = 0; k < 8; ++k) cv[k] = trow[k];
      }
      for (int k0 = 0; __any(run && (k0 < T2)); k0 += 8) {
        if (run && (k0 < T2)) {
          kdone = k0 + 8; /* >= T2 behind the last trip: nothing is left for the general sequence below */
          double nv[8];
          if (PREFETCH) {
            if (k0 + 8 < T2) { /* next trip's cos values, read behind this trip's arithmetic */
#pragma unroll
              for (int k = 0; k < 8; ++k) nv[k] = trow[k0 + 8 + k];
            }
          } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) cv[k] = trow[k0 + k];
          }
          int x[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) x[k] = (int)ceil(Ad * ((Kd * cv[k] - Kd) + 1.0));
          if (PREFETCH) {
#pragma unroll
            for (int k = 0; k < 8; ++k) cv[k] = nv[k];
          }
          const VsRun8 r8 = vs_run8(ring, s.wpos, C, T2 + k0, lane);
          if (__any((x[7] < c.thr) || (k0 + 8 > T2))) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              const bool in = k0 + k < T2;                   /* for(i = par.T2; i < 2*par.T2; i++) */
              const bool brk = run && in && (x[k] < c.thr); /* if(x[i] < par.DC) break; */
              T3 = brk ? (T2 + k0 + k) : T3;
              run = run && !brk;
              psum = (run && in) ? (psum + vs_sq_f(x[k])) : psum;
            }
            vs_run8_store_all(r8, x);
            asm volatile("; falling trip with the break");
          } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) psum += vs_sq_f(x[k]);
            vs_run8_store_all(r8, x);
            asm volatile("; falling trip");
          }
        }
      }
    }
    /* general sequence: everything when !fast, nothing otherwise */
    for (int k0 = kdone; run && (k0 < T2); k0 += 4) {
      double v[4];
      int xsv[4];
      float xfv[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = trow[(k0 + k < T2) ? (k0 + k) : 0];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = Kd * v[k];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = v[k] - Kd;
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = v[k] + 1.0;
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = Ad * v[k];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = ceil(v[k]);
#pragma unroll
      for (int k = 0; k < 4; ++k) xsv[k] = vs_short_of(v[k]);
#pragma unroll
      for (int k = 0; k < 4; ++k) xfv[k] = (float)xsv[k];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = T2 + k0 + k;
        const bool act = run && (k0 + k < T2);
        const bool brk = act && (xfv[k] < c.DC); /* if(x[i] < par.DC) break; */
        T3 = brk ? i : T3;
        run = run && !brk;
        const bool keep = act && !brk;
        psum = keep ? (psum + xfv[k] * xfv[k]) : psum;
        ring[(keep && (i < lim)) ? vs_ring_at(s.wpos, C, i, lane) : vs_ring_idx(C, lane)] =
            (int16_t)xsv[k];
      }
    }
  }

  VS_DIAG_ADD(dg, 2)
  if (PUB && !(((c.flags & VS_DF_NOISE) != 0) && T4 > 0)) {
    /* wave-specialised kernel: the open phase [0, T3) is in the ring -- let the filter wave have
     * it while the closed phase is still being written (the LDS keeps this store behind the
     * ring writes above).  Not when noise will still be added to [0, T4) below. */
    VS_LDS_RELEASE();
    __hip_atomic_store(gpub_lane, s.g + ((T3 < lim) ? T3 : lim), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  float x_pow = 0.0f, w_pow = 0.0f;
  const bool noisy = (c.flags & VS_DF_NOISE) != 0;
  bool handed = false; /* three-role kernel: the cycle's noise went to the noise wavefront as an order */

  if (!noisy) {
    /* ---- closed phase without noise: fg:334-336 ---- */
    if (fast) {
      /* trips of 8 from T3; the last one may run up to 7 slots into the next cycle */
      for (int i = T3; __any(i < T); i += 8) {
        const VsRun8 r8 = vs_run8_or_trash(i < T, ring, s.wpos, C, i, lane);
        const int x[8] = {c.dcs, c.dcs, c.dcs, c.dcs, c.dcs, c.dcs, c.dcs, c.dcs};
        vs_run8_store_all(r8, x);
      }
    } else {
      for (int i = T3; i < lim; ++i) ring[vs_ring_at(s.wpos, C, i, lane)] = (int16_t)c.dcs;
    }
    VS_DIAG_ADD(dg, 3)
  } else {
    /* ---- closed phase with noise: fg:373-411 ---- */
    x_pow = psum / ((float)T3 - (float)T4);
    const float aux = (float)(1.0 + (double)(((float)T3 - (float)T4) / ((float)T)));
    const float arg = 12.0f * aux * x_pow / c.noise;
#if defined(VS_TIMING_STUB) && (VS_TIMING_STUB & 2)
    const int NDW = 2000 + (int)(arg * 0.0f); /* measurement builds only: what the noise width costs */
#else
    const int NDW = vs_isqrt_floor((double)arg);
#endif
    const double NDWd = (double)NDW;
    const double half = NDWd / 2.0;
    const int ntail = (T > T3) ? (T - T3) : 0;
    const int m = T4 + ntail; /* draws this cycle: [0,T4) then [T3,T) */
    const uint32_t d0 = s.d;
    const uint32_t bfirst = d0 >> 2;
    const int nblk = (m > 0) ? (int)(((d0 + (uint32_t)m - 1u) >> 2) - bfirst) + 1 : 0;
    float wsum = 0.0f;
    /* short sequence: noise only behind the pulse (T4 == 0, the usual case: DC flow 0.25 after
     * -n), a width the one-fma form is proved for, and samples that cannot reach the clamp */
    const int absdc = (c.dcs < 0) ? -c.dcs : c.dcs;
    const bool nfast = fast && __all((T4 == 0) && (NDW <= VS_NDW_FAST) && ((NDW >> 1) + 2 + absdc <= 32767));
    if (nfast) {
      /* draw ordinal q = 0..m-1 belongs to sample T3 + q.  Two Philox blocks (8 draws) per
       * trip; word 0 of the first block has ordinal q0 in -3..0 (the scalar draws of this cycle
       * sit in front of it), so the first trip masks its leading words.  A lane that is done
       * (q0 >= m) sends its trips to the trash rows; the trip in which a lane ends may run up to
       * 7 slots into the next cycle. */
      if (SPLIT) {
        /* three-role kernel: the noise wavefront does this part -- post the order (the open phase
         * is in the ring: the LDS keeps the order behind those stores) */
        vs_post_order(ord, lane, s, d0, T3, T, NDW);
        handed = true;
      } else {
        /* the lane's ten round keys: the caller's, made once per launch (the generator wavefront of the
         * two-role kernel has the registers), or made here per cycle (the one-wave kernel has not) */
        if (keys) {
          vs_noise_trips<PUB, false>(ring, C, lane, *keys, vs_noise_consts(NDW, c.dcs), d0, m, s.wpos, T3, s.g, gpub_lane);
        } else {
          VsRoundKeys rk;
          vs_round_keys(c.key0, c.key1, rk);
          vs_noise_trips<PUB, false>(ring, C, lane, rk, vs_noise_consts(NDW, c.dcs), d0, m, s.wpos, T3, s.g, gpub_lane);
        }
      }
    } else if (T4 == 0) {
      /* T4 == 0 on the general sequence: the draws map to i = T3 + q, q = 0..m-1 */
      int mlim = lim - T3;
      mlim = (mlim < m) ? mlim : m;
      mlim = (mlim > 0) ? mlim : 0;
      int slot0 = s.wpos + T3; /* ring slot of q == 0; T3 <= P + 2 < C */
      if (slot0 >= C) slot0 -= C;
      int q0 = (int)(4u * bfirst - d0); /* ordinal of word 0 of the first block, -3..0 */
      /* two Philox blocks (8 draws) per trip: their dependency chains interleave */
      for (int bi = 0; bi < nblk; bi += 2) {
        const uint32_t b = bfirst + (uint32_t)bi;
        uint32_t o[8];
        vs_philox(b, c.key0, c.key1, o[0], o[1], o[2], o[3]);
        vs_philox(b + 1u, c.key0, c.key1, o[4], o[5], o[6], o[7]);
        double u8[8];
        int wv8[8];
#pragma unroll
        for (int w = 0; w < 8; ++w) u8[w] = vs_unit_of_draw(o[w] >> 1);
#pragma unroll
        for (int w = 0; w < 8; ++w) u8[w] = u8[w] * NDWd;
#pragma unroll
        for (int w = 0; w < 8; ++w) u8[w] = u8[w] - half;
#pragma unroll
        for (int w = 0; w < 8; ++w) u8[w] = ceil(u8[w]);
#pragma unroll
        for (int w = 0; w < 8; ++w) wv8[w] = vs_short_of(u8[w]);
#pragma unroll
        for (int w = 0; w < 8; ++w) {
          const int q = q0 + w;
          if (LOG) {
            if ((unsigned)q < (unsigned)m) wsum += (float)wv8[w] * (float)wv8[w];
          }
          /* w[i] = (short)ceil(((1.0*random())/RAND_MAX)*NDW - NDW/2.0); x[i] = truncate(DC + w) */
          int xv = c.dcs + wv8[w];
          xv = (xv > 32767) ? 32767 : ((xv < -32767) ? -32767 : xv);
          const bool ok = (unsigned)q < (unsigned)mlim;
          const unsigned sl = (unsigned)(slot0 + q), wr = sl - (unsigned)C; /* (slot0 + q) mod C */
          const int slot = (int)((sl < wr) ? sl : wr);
          ring[vs_ring_idx(ok ? slot : C, lane)] = (int16_t)xv;
        }
        q0 += 8;
        if (PUB) {
          const int done = (q0 < mlim) ? q0 : mlim; /* noise samples [T3, T3 + done) are written */
          VS_LDS_RELEASE();
          __hip_atomic_store(gpub_lane, s.g + T3 + ((done > 0) ? done : 0), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
    } else {
      /* general case: draws cover [0,T4) then [T3,T) */
      for (int bi = 0; bi < nblk; ++bi) {
        const uint32_t b = bfirst + (uint32_t)bi;
        uint32_t o0, o1, o2, o3;
        vs_philox(b, c.key0, c.key1, o0, o1, o2, o3);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const uint32_t ow = (w == 0) ? o0 : (w == 1) ? o1 : (w == 2) ? o2 : o3;
          const int q = (int)(4u * b + (uint32_t)w - d0); /* ordinal of this draw in the cycle */
          const bool act = (q >= 0) && (q < m);
          const int i = (q < T4) ? q : (T3 + (q - T4));
          /* w[i] = (short)ceil(((1.0*random())/RAND_MAX)*NDW - NDW/2.0), fg:387,398 */
          const double u = vs_unit_of_draw(ow >> 1);
          const int wv = vs_short_of(ceil(u * NDWd - half));
          if (act) wsum += (float)wv * (float)wv;
          /* truncate((float)x[i] + w[i]).  x[i] is (short)DC on [T3,T) by construction and, for
           * a monotone rising flank, on [0,T4) too -- but an amplitude above 32767 wraps the
           * (short) conversion and leaves genuine pulse samples below T4, so those are read
           * back from the ring */
          if (act && (i < lim)) {
            const int idx = vs_ring_at(s.wpos, C, i, lane);
            const int base = (q < T4) ? (int)ring[idx] : c.dcs;
            int xv = base + wv;
            xv = (xv > 32767) ? 32767 : ((xv < -32767) ? -32767 : xv);
            ring[idx] = (int16_t)xv;
          }
        }
      }
    }
    s.d = d0 + (uint32_t)m;
    w_pow = wsum / (float)T;
    VS_DIAG_ADD(dg, 4)
  }

  if (LOG) {
    if (logrow && s.cyc < log_cap) {
      vs_cycle_rec rec;
      rec.S = S;
      rec.x_pow = noisy ? x_pow : 0.0f;
      rec.w_pow = noisy ? w_pow : 0.0f;
      rec.T = T;
      logrow[s.cyc] = rec;
    }
  }

  /* three-role kernel: a cycle this wavefront has written in full still goes through the noise
   * wavefront, which is the one that publishes progress to the filter -- as an empty order */
  if (SPLIT && !handed) vs_post_order(ord, lane, s, s.d, T, T, 0);

  /* ---- emit bookkeeping: fg:413-423 ---- */
  s.cyc += 1;
  s.g += T;
  int wp = s.wpos + T;
  if (wp >= C) wp -= C;
  s.wpos = wp;
  VS_DIAG_ADD(dg, 5)
}

#endif
