/*
 * vs_dev_filter.h -- the filter: one super-step of 24 samples of vowel_new.c:266-289 per lane (vs_superstep)
 * Included by vs_kernels.hip only (device code, one translation unit per build: the 64-column build and
 * the narrow one, -DVS_GROUP_LANES=16).
 */
#ifndef VS_DEV_FILTER_H
#define VS_DEV_FILTER_H

/* the lane's 22 taps as vs_superstep wants them: A[1..22] for VS_ARITH_EXACT, -A[1..22] for VS_ARITH_FMA */
template <int ARITH>
__device__ __forceinline__ void vs_load_taps(const double *__restrict__ taps, const VsDevLane *__restrict__ L,
                                             double (&a)[VS_ORDER + 1])
{
  const double *__restrict__ row = taps + (size_t)L->tap_row * VS_ORDER; /* the lane's row of the plan's tap table */
  a[0] = 1.0;
#pragma unroll
  for (int j = 1; j <= VS_ORDER; ++j) {
    a[j] = (ARITH == VS_ARITH_FMA) ? -row[j - 1] : row[j - 1];
    /* the sign goes INTO the register: left to itself the compiler keeps +A and folds the negation back into every
     * multiply-add as a source modifier -- free, but only the 8-byte encoding has modifiers */
    if (ARITH == VS_ARITH_FMA) asm volatile("" : "+v"(a[j]));
  }
}

/* the 8 samples of one granule / of 16 bytes of a PCM row, as integers */
__device__ __forceinline__ void vs_unpack8(const vs_u32x4 v, int *x)
{
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    x[2 * e] = (int)(int16_t)(v[e] & 0xFFFFu);
    x[2 * e + 1] = (int)v[e] >> 16;
  }
}

/*
 * vowel -n inside the fused kernels (vowel_new.c:303-309): the filter wavefront of a wave-specialised kernel whose lanes
 * share one position and one frame length sums (float)y*y as its results are packed -- in sample order, so the float
 * rounding sequence is the reference's.  Frames end on multiples of 4 samples (Lframe is a multiple of 100, a super-step
 * starts on a multiple of 24): behind every fourth sample the sum is set aside and started afresh IF the frame ends there --
 * a scalar compare and two selects, no branch inside the super-step (a branch there makes the old and the new register
 * assignment of everything that is live meet behind it: the first version cost 2.9 vector instructions per sample in
 * copies) -- and the caller turns what was set aside into that frame's NoiseDistWidth between two super-steps.  Three
 * vector instructions per sample in place of a 2 B/sample streaming pass over the finished PCM.  What it cannot vouch for
 * it leaves to that pass (vs_out_power_kernel, fill mode): a frame during which round2int()'s quirk path rounded a
 * super-step again gets NaN, and frames behind the last whole super-step are not counted in done[].
 */
#define VS_ONDW_UNKNOWN 0x7FC00000u /* NaN: "this frame's power is the streaming pass's to find" */
struct VsFramePower {
  float sum;   /* aux of vowel_new.c:303-306 for the frame this lane is in */
  float done;  /* ... of the frame that ended inside the super-step just run (tb != 0) */
  int tb;      /* wave-uniform: the frame ends behind sample tb - 1 of the super-step being run (4, 8, .. 24), 0: not in it */
  int len;     /* wave-uniform: samples of the frame that ends there (ni of vw:307) */
  int frame;   /* wave-uniform: index of the frame the super-step starts in */
  bool bad;    /* wave-uniform: the frame the super-step starts in has seen the quirk path */
  bool requirk; /* wave-uniform: ... and so has this super-step (the caller folds it into `bad` of the frame it ends in) */
  bool force;  /* wave-uniform, tests (VS_FAULT_REROUND): this super-step takes the quirk path whatever its arguments */
  /* what a lane needs at a frame's end -- its record (out_snr, row), the table -- is looked up THERE, once per 800 or 1100
   * samples: the super-step has no register to spare for it (168 per wavefront in the three-role kernels) */
  const VsDevLane *lanes;
  float *ondw;
  long ondw_pitch;
  long first_lane; /* wave-uniform: record index of this wavefront's lane 0 */
  int n_lanes;
};
/* sqrt(v) correctly rounded to double whatever the last bit of the device sqrt: s is at most
 * one ulp off, the residual r = v - s*s is exact in one fma, and the true root lies beyond
 * s + ulp/2 exactly when r > s*ulp (a root of a double is never a rounding midpoint). */
__device__ __forceinline__ double vs_sqrt_rn(double v)
{
  double s = sqrt(v);
  if (!(v > 0.0) || !(s > 0.0)) return s;
  const double r = __builtin_fma(-s, s, v);
  const double up = __longlong_as_double(__double_as_longlong(s) + 1) - s; /* ulp above s */
  const double dn = s - __longlong_as_double(__double_as_longlong(s) - 1); /* ulp below s */
  if (r > s * up) s = s + up;
  else if (-r > s * dn) s = s - dn;
  return s;
}
/* NoiseDistWidth = sqrt(12*sig_power/snr), sig_power = aux/(float)ni: float arithmetic, the root through double (vw:307-309) */
__device__ __forceinline__ float vs_noise_width(float aux, int ni, float snr)
{
  const float sig_power = aux / (float)ni;
  return (float)vs_sqrt_rn((double)(12.0f * sig_power / snr));
}
/* two results as they are packed: aux += (float)y*y, low sample first */
__device__ __forceinline__ void vs_power_pair(uint32_t w, float &aux)
{
  const float lo = (float)(int)(int16_t)(w & 0xFFFFu), hi = (float)((int)w >> 16);
  aux += lo * lo;
  aux += hi * hi;
}
/* the entry of the frame that has just ended: its width from the sum that was set aside, or "unknown" */
__device__ __forceinline__ void vs_frame_power_store(const VsFramePower &fp, bool unknown)
{
  const long gl = fp.first_lane + (long)(threadIdx.x & (VS_WAVE - 1));
  if (gl < (long)fp.n_lanes) {
    const VsDevLane *__restrict__ L = fp.lanes + gl;
    const float snr = L->out_snr;
    if (snr > 0.0f)
      fp.ondw[(long)L->row * fp.ondw_pitch + fp.frame] = unknown ? __uint_as_float(VS_ONDW_UNKNOWN) : vs_noise_width(fp.done, fp.len, snr);
  }
}
/* behind sample t of a super-step (t + 1 a multiple of 4): if the frame ends here, its sum is set aside and the next one's
 * starts -- tb is wave-uniform, so `ends` is a scalar condition and the two selects take it as a mask */
__device__ __forceinline__ void vs_frame_power_mark(VsFramePower &fp, int t)
{
  const bool ends = fp.tb == t + 1;
  fp.done = ends ? fp.sum : fp.done;
  fp.sum = ends ? 0.0f : fp.sum;
}

/*
 * One filter super-step of one lane: 24 samples of vowel_new.c:266-289 starting at the lane's
 * own position n.  x comes from the lane's ring column (rp = &ring[rslot][lane], never wraps
 * inside a super-step because ring_slots is a multiple of VS_SS and rslot advances by VS_SS
 * from 0) or, for VS_KIND_FILTER, from HBM.  The 24 int16 results leave as three
 * 16-byte stores (store_ok: lanes beyond the batch run along in the all-lanes loop of the
 * wave-specialised kernel and must not store).  y[] is the rotating window of the last 24 outputs
 * in double (y[t] = y at n+t-24 on entry, = y at n+t on exit).
 */
template <int ARITH, int KIND, bool PRE1 = false, bool PACKED = false, int WHOLE = -1, bool LATE = false, bool POW = false>
__device__ __forceinline__ void vs_superstep(const double (&a)[VS_ORDER + 1], double (&y)[VS_SS],
                                             double gain, double pre, const int16_t *rp,
                                             const int16_t *__restrict__ irow,
                                             int16_t *__restrict__ orow, int n, int N, bool vec_ok,
                                             int (&outv)[VS_SS], vs_u32x4 (&xnext)[VS_SS / 8],
                                             bool store_ok = true, VsFramePower *fp = nullptr)
{
  static_assert(!POW || (PACKED && WHOLE == 1), "frame powers ride on the eager packing of the branch-free super-step");
  /* WHOLE: the caller has taken this decision out of its loop (1: 16-byte stores, 0: sample by sample).
   * With it -- and store_ok a constant -- nothing branches between the ring reads and their use. */
  const bool whole = (WHOLE < 0) ? (vec_ok && (n + VS_SS <= N)) : (WHOLE != 0);
  int xin[VS_SS];
  if (KIND == VS_KIND_FILTER) {
    if (whole) {
      /* xnext[] holds this super-step's 48 bytes, loaded one super-step ago; the loads for the
       * next one are issued now and complete behind the ~1300 instructions below (with one
       * wave per SIMD nothing else hides an HBM round trip) */
#pragma unroll
      for (int k = 0; k < VS_SS / 8; ++k) vs_unpack8(xnext[k], &xin[8 * k]);
      if (n + 2 * VS_SS <= N) {
#pragma unroll
        for (int k = 0; k < VS_SS / 8; ++k) xnext[k] = *(const vs_u32x4 *)(irow + n + VS_SS + 8 * k);
      }
    } else {
#pragma unroll
      for (int t = 0; t < VS_SS; ++t) xin[t] = (n + t < N) ? (int)irow[n + t] : 0;
    }
  } else {
    /* the first 8 now, the rest in two more batches issued from inside the sample loop (each a
     * chunk ahead of its use): 24 ring samples held at once are 16 registers too many */
#pragma unroll
    for (int t = 0; t < 8; ++t) xin[t] = (int)rp[t * VS_GROUP_LANES];
  }

  /* results leave in chunks of 8 samples = one 16-byte store, as soon as a chunk is complete: 24
   * pending results would cost 24 registers (the three-role kernel runs at 168 per wavefront) */
  /* EAGER (the branch-free super-steps of the wave-specialised kernels): a pair of results is clamped and packed as
   * soon as its second sample is rounded, so a chunk in flight holds four packed words and at most one unpaired
   * result instead of eight integers -- the three registers between 168 and a spill of the three-role kernel's
   * pre-emphasis variants */
  constexpr bool EAGER = PACKED && (WHOLE == 1);
  /* LATE (the two-role kernels, which run at two wavefronts per SIMD and have the registers): the packed words of the
   * whole super-step are held -- 12 registers instead of 4 -- and leave as three 16-byte stores back to back at its
   * end.  Measured, same box (profiles/r05_store_width_probe.txt): config 2 -4 %, config 5 -1.5 % exact, fma alike;
   * HBM write traffic 1.11 x instead of 1.08 x the algorithmic bytes (the L2 merges early stores better, as round 3
   * found on the three-role kernel) -- taken for the time, not for the traffic. */
  constexpr int PKN = (EAGER && LATE) ? 12 : 4;
  uint32_t pk[PKN];
  auto put8 = [&](int k) {
    if (EAGER && LATE) {
      if (k == VS_SS / 8 - 1) {
#pragma unroll
        for (int c = 0; c < VS_SS / 8; ++c) {
          vs_u32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = pk[(4 * c + e) % PKN];
          if (store_ok) *(vs_u32x4 *)(orow + n + 8 * c) = v;
        }
      }
    } else if (EAGER) {
      vs_u32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = pk[e];
      if (store_ok) *(vs_u32x4 *)(orow + n + 8 * k) = v;
    } else if (whole) {
      vs_u32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        v[e] = PACKED ? vs_clamp_pack16(outv[8 * k + 2 * e], outv[8 * k + 2 * e + 1])
                      : vs_pack16(outv[8 * k + 2 * e], outv[8 * k + 2 * e + 1]);
      if (store_ok) *(vs_u32x4 *)(orow + n + 8 * k) = v;
    } else {
#pragma unroll
      for (int t = 8 * k; t < 8 * k + 8; ++t) {
        int v = outv[t];
        if (PACKED) v = (v > 32767) ? 32767 : ((v < -32767) ? -32767 : v);
        if (store_ok && (n + t < N)) orow[n + t] = (int16_t)v;
      }
    }
  };

  if (KIND == VS_KIND_SOURCE) {
#pragma unroll
    for (int t = 8; t < VS_SS; ++t) xin[t] = (int)rp[t * VS_GROUP_LANES];
#pragma unroll
    for (int t = 0; t < VS_SS; ++t) outv[t] = xin[t]; /* the flow itself */
#pragma unroll
    for (int k = 0; k < VS_SS / 8; ++k) put8(k);
  } else {
    const double ym1 = y[VS_SS - 1]; /* y[n-1]: only the quirk path below needs it once y[23] is replaced */
    int qhi = 0x7FFFFFFF;            /* signed minimum of the high words of the rounded values */
    uint32_t qlo = 0u;               /* unsigned maximum of their low words */
#pragma unroll
    for (int t = 0; t < VS_SS; ++t) {
      if (KIND != VS_KIND_FILTER && (t & 7) == 0) {
        if (t + 8 < VS_SS) {
#pragma unroll
          for (int u = t + 8; u < t + 16; ++u) xin[u] = (int)rp[u * VS_GROUP_LANES];
        }
        /* this chunk's eight samples are all "used" here: ONE s_waitcnt for the batch (the LDS answers in
         * order) instead of one in front of every sample's first use -- to a wavefront that issues an
         * instruction every ~5.3 ticks whatever it is, a wait that has nothing to wait for costs as much
         * as a multiplication (tools/ubench/ubench5.hip) */
        /* ... and the code behind it starts on an 8-byte boundary.  The super-step is a straight line of ~1300
         * instructions, most of them 8 bytes long (fp64 arithmetic only has the 64-bit VOP3 encoding); the 4-byte ones
         * come in pairs per sample (conversion + fused first difference, ceil + conversion), so a chunk of 8 samples
         * is either in step with the 8-byte grid or out of it as a whole -- and an 8-byte instruction that straddles
         * an 8-byte boundary costs a wavefront that is ALONE on its SIMD about one cycle of instruction fetch
         * (half-filled chips: BASELINE config 4's shard, config 2; profiles/r05_loop_alignment.txt: the same loop
         * took 4.58 or 4.82 ms, config 2 1.58 or 1.80 ms, depending on one 4-byte instruction more in the kernel's
         * prologue).  What puts a chunk out of step is the lone 4-byte s_waitcnt in front of this statement (and the
         * loop control in front of the first chunk): the assembler pads with one s_nop when it has. */
        asm volatile(".p2align 3" ::"v"(xin[t]), "v"(xin[t + 1]), "v"(xin[t + 2]), "v"(xin[t + 3]), "v"(xin[t + 4]),
                     "v"(xin[t + 5]), "v"(xin[t + 6]), "v"(xin[t + 7]));
      }
      /* y_double[0] = 0.0 + B[0]*x[i]*gain, B = {1, 0, ...} (vowel_new.c:266-269, 435-448) */
      double acc;
      const double y1 = y[(t + VS_SS - 1) % VS_SS];
      if (ARITH == VS_ARITH_EXACT) {
        /* y_double[0] = y_double[0] - A[j]*y_double[j], j = 1..22, each product and each
         * difference rounded on its own.  x*gain itself is EXACT in double -- an int16 times a float
         * gain has at most 16 + 24 significant bits -- so the first difference, x*gain - RN(A[1]*y[1]),
         * is one fused multiply-add with the same single rounding: one instruction less per sample. */
        acc = __builtin_fma((double)xin[t], gain, -(a[1] * y1));
#pragma unroll
        for (int j = 2; j <= VS_ORDER; ++j) acc = acc - a[j] * y[(t + VS_SS - j) % VS_SS];
      } else {
        acc = (double)xin[t] * gain;
        /* two partial sums over the older taps (a lone wavefront issues an independent fp64
         * instruction every ~5.3 ticks and a dependent one every ~8.4, so two alternating chains
         * never wait), the newest tap (j = 1) last: it is the only one on the sample-to-sample
         * critical path */
        /* VS_ARITH_FMA: a[] holds the NEGATED coefficients (vs_load_taps): p - A*y is fma(-A, y, p), and with the
         * sign in the register the multiply-add accumulates in place -- V_FMAC_F64, a 4-byte instruction, where the
         * negating form needs the 8-byte VOP3 encoding: half the loop's code bytes, and nothing that can straddle an
         * 8-byte boundary (see the chunk alignment above) */
        double p0 = acc, p1 = a[2] * y[(t + VS_SS - 2) % VS_SS];
#pragma unroll
        for (int j = 3; j <= VS_ORDER; ++j) {
          const double yj = y[(t + VS_SS - j) % VS_SS];
          if (j & 1) p0 = __builtin_fma(a[j], yj, p0);
          else p1 = __builtin_fma(a[j], yj, p1);
        }
        acc = __builtin_fma(a[1], y1, p0 + p1);
      }
      /* y[i] = round2int(y_double[0] - pre_emphasis*y_double[1]), vowel_new.c:284.  PRE1: every
       * lane has pre_emphasis == 1.0 (the reference's default), and 1.0*y is y exactly */
      const double o = PRE1 ? (acc - y1)
                            : ((ARITH == VS_ARITH_EXACT) ? (acc - pre * y1) : __builtin_fma(-pre, y1, acc));
      /* PACKED: the caller does not look at outv[]; the clamp rides on the packing (put8) */
      if (ARITH == VS_ARITH_FMA && PACKED) {
        /* The tolerance mode owes the reference <= 1 LSB, not its rounding quirks: round to nearest (ties to even:
         * V_RNDNE_F64 + the saturating convert, two instructions instead of three) and no quirk watch -- round2int()
         * differs from this only on exact ties (half-down there) and on its quirk set (one LSB, one chance in 2^32
         * per sample or a signal below 2^-54), both inside the mode's contract (tests: <= 1 LSB, RMS <= 1e-5). */
        outv[t] = (int)__builtin_rint(o);
        asm volatile("" : "+v"(outv[t]));
      } else {
        outv[t] = PACKED ? vs_round2int_half_down_unclamped(o) : vs_round2int_half_down(o);
        /* rounded HERE: left to itself the compiler keeps all 24 arguments (48 registers) and rounds
         * them behind the quirk test below, where the other branch does not need the results */
        asm volatile("" : "+v"(outv[t]));
        const int ohi = __double2hiint(o);
        const uint32_t olo = (uint32_t)__double2loint(o);
        qhi = (ohi < qhi) ? ohi : qhi;
        qlo = (olo > qlo) ? olo : qlo;
      }
      y[t] = acc; /* replaces y[n-24]; the window rotates by renaming, vowel_new.c:287-289 */
      if (EAGER && (t & 1)) pk[(t >> 1) % PKN] = vs_clamp_pack16(outv[t - 1], outv[t]);
      if (POW && (t & 1)) {
        vs_power_pair(pk[(t >> 1) % PKN], fp->sum);
        if ((t & 3) == 3) vs_frame_power_mark(*fp, t);
      }
      if ((t & 7) == 7) put8(t >> 3);
      /* keep each sample's products next to its chain: hoisted across samples they only park
       * in the accumulator registers and come back, two moves each way */
      __builtin_amdgcn_sched_barrier(0);
    }
    /* (the tolerance mode of the packed super-steps has no quirk path: it rounds to nearest-even, once) */
    if (__any((qhi <= VS_R2I_Q1_HI) || (qlo == 0xFFFFFFFFu)) || (POW && ARITH == VS_ARITH_EXACT && fp->force)) {
      /* some argument of this super-step may sit in round2int()'s quirk set (a signal that has
       * decayed to below 2^-54, or one chance in 2^32 per sample): round all of it again,
       * literally, and store it again */
      /* the sums have seen the first rounding: the frame this super-step ends in, and the one that ended inside it, are
       * left to the streaming pass (the caller stores "unknown") */
      if (POW) fp->requirk = true;
#pragma unroll
      for (int t = 0; t < VS_SS; ++t) {
        const double y1 = (t == 0) ? ym1 : y[t - 1];
        const double o = PRE1 ? (y[t] - y1)
                              : ((ARITH == VS_ARITH_EXACT) ? (y[t] - pre * y1) : __builtin_fma(-pre, y1, y[t]));
        outv[t] = vs_round2int(o);
      }
#pragma unroll
      for (int k = 0; k < VS_SS / 8; ++k) {
        if (EAGER) {
#pragma unroll
          for (int e = 0; e < 4; ++e) pk[(4 * k + e) % PKN] = vs_clamp_pack16(outv[8 * k + 2 * e], outv[8 * k + 2 * e + 1]);
        }
        put8(k);
      }
    }
  }
}

/*
 * VS_ARITH_F32 (opt-in): the recurrence of vowel_new.c:279-281 in SINGLE precision with PACKED fused
 * multiply-adds -- two taps per V_PK_FMA_F32 -- for callers who accept a measured distance from the reference instead of
 * its rounding sequence (include/voice_synth.h has the table: RMS up to 1.9e-5 of full scale, 25 LSB at most, worst for
 * /i/ at the default gain; tests/test_gpu_f32.py holds the kernels to it).  The window is 12 register pairs
 * {y[2q], y[2q+1]} (24 registers where the double window takes 48); for an even sample index the 22 taps are 11 aligned
 * pairs, for an odd one 10 pairs plus the newest and the oldest tap on their own: nce[k] = -{A[2k+2], A[2k+1]},
 * nco[k] = -{A[2k+3], A[2k+2]}.  17.5 vector instructions per sample where VS_ARITH_FMA takes 29 and VS_ARITH_EXACT 52.
 * Only the fused wave-specialised kernels have it; every other kernel (source-only and filter-only kinds, the one-wave
 * and the wide kernel) runs VS_ARITH_FMA when the context asks for VS_ARITH_F32.
 */
typedef float vs_f32x2 __attribute__((ext_vector_type(2)));
struct VsF32Filter {
  vs_f32x2 nce[11], nco[10], yp[12];
  float na1, na22, gain, pre;
};
__device__ __forceinline__ void vs_f32_load(const double *__restrict__ taps, const VsDevLane *__restrict__ L, VsF32Filter &f)
{
  const double *__restrict__ row = taps + (size_t)L->tap_row * VS_ORDER; /* row[j - 1] = A[j] */
#pragma unroll
  for (int k = 0; k < 11; ++k) f.nce[k] = (vs_f32x2){-(float)row[2 * k + 1], -(float)row[2 * k]};
#pragma unroll
  for (int k = 0; k < 10; ++k) f.nco[k] = (vs_f32x2){-(float)row[2 * k + 2], -(float)row[2 * k + 1]};
#pragma unroll
  for (int k = 0; k < 12; ++k) f.yp[k] = (vs_f32x2){0.0f, 0.0f}; /* vowel_new.c:222-224 */
  f.na1 = -(float)row[0];
  f.na22 = -(float)row[21];
  f.gain = (float)L->gain;
  f.pre = (float)L->pre;
}
/* round to nearest, ties upwards, saturating: V_CVT_RPI_I32_F32 = floor(x + 0.5) in one instruction (the tolerance modes owe
 * the reference a distance, not its half-down ties) */
__device__ __forceinline__ int vs_round_f32(float x)
{
  int r;
  asm("v_cvt_rpi_i32_f32_e32 %0, %1" : "=v"(r) : "v"(x));
  return r;
}
/* one super-step of 24 samples; WHOLE: 1 = inside the row, 16-byte stores (the branch-free loops), 0 = sample by sample
 * against N (a row's last super-step, unaligned rows) */
template <bool PRE1, int WHOLE, bool POW>
__device__ __forceinline__ void vs_superstep_f32(VsF32Filter &f, const int16_t *rp, int16_t *__restrict__ orow, int n, int N,
                                                 bool store_ok, VsFramePower *fp = nullptr)
{
  static_assert(!POW || WHOLE == 1, "frame powers ride on the packed stores of the branch-free super-step");
  int xin[VS_SS];
#pragma unroll
  for (int t = 0; t < 8; ++t) xin[t] = (int)rp[t * VS_GROUP_LANES];
  uint32_t pk[4];
#pragma unroll
  for (int m = 0; m < VS_SS / 2; ++m) {
    if ((m & 3) == 0) {
      if (2 * m + 8 < VS_SS) { /* the next chunk's ring samples, a chunk ahead of their use */
#pragma unroll
        for (int u = 2 * m + 8; u < 2 * m + 16; ++u) xin[u] = (int)rp[u * VS_GROUP_LANES];
      }
      /* this chunk's eight samples are all "used" here: ONE s_waitcnt for the batch instead of one per sample (vs_superstep) */
      asm volatile("" ::"v"(xin[2 * m]), "v"(xin[2 * m + 1]), "v"(xin[2 * m + 2]), "v"(xin[2 * m + 3]), "v"(xin[2 * m + 4]),
                   "v"(xin[2 * m + 5]), "v"(xin[2 * m + 6]), "v"(xin[2 * m + 7]));
    }
    int o0, o1;
    {
      /* even sample t = 2m: taps 1..22 are the pairs yp[m-1], yp[m-2], .. yp[m-11];
       * odd sample t = 2m + 1: tap 1 is the sample just made, taps 2..21 the pairs yp[m-1] .. yp[m-10], tap 22 the high
       * half of yp[m+1].  The two chains are written side by side: a packed multiply-add that reads the result of the one
       * in front of it costs a wait state (an s_nop: an issue slot), and all of the odd sample but its newest tap is
       * independent of the even one */
      /* (the input term x*gain joins at the END of each chain, as one fused multiply-add into the sum of the two halves:
       * the chains start from a packed product instead of {x*gain, 0} -- a multiplication and a zero less per sample) */
      vs_f32x2 pe = f.nce[0] * f.yp[(m + 11) % 12];
      vs_f32x2 po = f.nco[0] * f.yp[(m + 11) % 12];
#pragma unroll
      for (int k = 1; k < 10; ++k) {
        pe = __builtin_elementwise_fma(f.nce[k], f.yp[(m + 11 - k) % 12], pe);
        po = __builtin_elementwise_fma(f.nco[k], f.yp[(m + 11 - k) % 12], po);
      }
      pe = __builtin_elementwise_fma(f.nce[10], f.yp[(m + 1) % 12], pe);
      float sc = __builtin_fmaf(f.na22, f.yp[(m + 1) % 12].y, po.y);
      const float acc0 = __builtin_fmaf((float)xin[2 * m], f.gain, pe.x + pe.y);
      sc = __builtin_fmaf((float)xin[2 * m + 1], f.gain, sc + po.x);
      const float y1 = f.yp[(m + 11) % 12].y;
      o0 = vs_round_f32(PRE1 ? (acc0 - y1) : __builtin_fmaf(-f.pre, y1, acc0)); /* vowel_new.c:284 */
      const float acc1 = __builtin_fmaf(f.na1, acc0, sc);
      o1 = vs_round_f32(PRE1 ? (acc1 - acc0) : __builtin_fmaf(-f.pre, acc0, acc1));
      f.yp[m] = (vs_f32x2){acc0, acc1}; /* replaces y[n-24], y[n-23]: the window rotates by renaming, vowel_new.c:287-289 */
      __builtin_amdgcn_sched_barrier(0);
    }
    if (WHOLE == 1) {
      pk[m & 3] = vs_clamp_pack16(o0, o1);
      if (POW) {
        vs_power_pair(pk[m & 3], fp->sum);
        if (m & 1) vs_frame_power_mark(*fp, 2 * m + 1);
      }
      if ((m & 3) == 3) {
        vs_u32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = pk[e];
        if (store_ok) *(vs_u32x4 *)(orow + n + 2 * m - 6) = v;
      }
    } else {
      const int c0 = (o0 > 32767) ? 32767 : ((o0 < -32767) ? -32767 : o0), c1 = (o1 > 32767) ? 32767 : ((o1 < -32767) ? -32767 : o1);
      if (store_ok && (n + 2 * m < N)) orow[n + 2 * m] = (int16_t)c0;
      if (store_ok && (n + 2 * m + 1 < N)) orow[n + 2 * m + 1] = (int16_t)c1;
    }
  }
}

#endif
