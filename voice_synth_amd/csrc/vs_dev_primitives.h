/*
 * vs_dev_primitives.h -- what the device code is built from: Philox4x32-10 and the draw contract, the
 * reference's conversions and roundings (round2int, its half-down form, the packed clamp), the integer
 * square root, the LDS ring (slot-major columns, 8-slot runs that may wrap, the compare/select block)
 * Included by vs_kernels.hip only (device code, one translation unit per build: the 64-column build and
 * the narrow one, -DVS_GROUP_LANES=16).
 */
#ifndef VS_DEV_PRIMITIVES_H
#define VS_DEV_PRIMITIVES_H

#define VS_PHILOX_M0 0xD2511F53u
#define VS_PHILOX_M1 0xCD9E8D57u
#define VS_PHILOX_W0 0x9E3779B9u
#define VS_PHILOX_W1 0xBB67AE85u

typedef uint32_t vs_u32x4 __attribute__((ext_vector_type(4), aligned(4)));   /* 16 bytes of a PCM row: rows are only 4-byte aligned */

/* Diagnostic build only (-DVS_DIAG, tools/diag_bench.py): s_memtime stamps at phase boundaries,
 * summed per wavefront into args.diag.  The shipped library is built without it. */
struct VsDiag {
  unsigned long long acc[8];
  unsigned long long t;
  unsigned long long rounds, attend; /* generator rounds and the lanes that took part in them */
};
#ifdef VS_DIAG
__device__ __forceinline__ unsigned long long vs_stamp()
{
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
#define VS_DIAG_ADD(dg, k)                     \
  {                                            \
    const unsigned long long tn_ = vs_stamp(); \
    (dg).acc[k] += tn_ - (dg).t;               \
    (dg).t = tn_;                              \
  }
#else
#define VS_DIAG_ADD(dg, k)
#endif

/* {lo & 0xFFFF, hi << 16} in one instruction (V_PERM_B32: bytes 0,1 of lo, then bytes 0,1 of hi) */
__device__ __forceinline__ uint32_t vs_pack16(int lo, int hi)
{
  return __builtin_amdgcn_perm((uint32_t)hi, (uint32_t)lo, 0x05040100u);
}

/* a ^ b ^ c in one instruction (gfx950 V_BITOP3_B32, truth table 0x96) */
__device__ __forceinline__ uint32_t vs_xor3(uint32_t a, uint32_t b, uint32_t c)
{
  return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
}

/* Philox4x32-10 (Salmon et al., SC'11), counter = (blk, 0, 0, 0). */
__device__ __forceinline__ void vs_philox(uint32_t blk, uint32_t k0, uint32_t k1, uint32_t &o0,
                                          uint32_t &o1, uint32_t &o2, uint32_t &o3)
{
  uint32_t c0 = blk, c1 = 0u, c2 = 0u, c3 = 0u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)VS_PHILOX_M0 * c0;
    const uint64_t p1 = (uint64_t)VS_PHILOX_M1 * c2;
    const uint32_t n0 = vs_xor3((uint32_t)(p1 >> 32), c1, k0);
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = vs_xor3((uint32_t)(p0 >> 32), c3, k1);
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += VS_PHILOX_W0;
    k1 += VS_PHILOX_W1;
  }
  o0 = c0; o1 = c1; o2 = c2; o3 = c3;
}

/* The ten round keys of a lane, made once per glottal cycle for the noise loop (the key
 * schedule k + r*W does not depend on the counter). */
struct VsRoundKeys {
  uint32_t a[10], b[10];
};
__device__ __forceinline__ void vs_round_keys(uint32_t k0, uint32_t k1, VsRoundKeys &rk)
{
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    rk.a[r] = k0 + (uint32_t)r * VS_PHILOX_W0;
    rk.b[r] = k1 + (uint32_t)r * VS_PHILOX_W1;
    /* keep them as values: rematerialising the additions inside the loop is what this avoids */
    asm volatile("" : "+v"(rk.a[r]), "+v"(rk.b[r]));
  }
}
/* two consecutive blocks (blk, blk + 1) with the prepared keys: 8 draws, chains interleaved */
__device__ __forceinline__ void vs_philox2(uint32_t blk, const VsRoundKeys &rk, uint32_t (&o)[8])
{
  uint32_t c0 = blk, c1 = 0u, c2 = 0u, c3 = 0u;
  uint32_t e0 = blk + 1u, e1 = 0u, e2 = 0u, e3 = 0u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)VS_PHILOX_M0 * c0;
    const uint64_t p1 = (uint64_t)VS_PHILOX_M1 * c2;
    const uint64_t s0 = (uint64_t)VS_PHILOX_M0 * e0;
    const uint64_t s1 = (uint64_t)VS_PHILOX_M1 * e2;
    const uint32_t n0 = vs_xor3((uint32_t)(p1 >> 32), c1, rk.a[r]);
    const uint32_t n2 = vs_xor3((uint32_t)(p0 >> 32), c3, rk.b[r]);
    const uint32_t m0 = vs_xor3((uint32_t)(s1 >> 32), e1, rk.a[r]);
    const uint32_t m2 = vs_xor3((uint32_t)(s0 >> 32), e3, rk.b[r]);
    c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
    e1 = (uint32_t)s1; e3 = (uint32_t)s0; e0 = m0; e2 = m2;
  }
  o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
  o[4] = e0; o[5] = e1; o[6] = e2; o[7] = e3;
}

/* per-lane constants of the generator */
struct VsCfg {
  float jitter, shimmer, K, Kvar, DC, noise, t_hi, t_lo, a_hi, a_lo;
  int amp, P, T2, tab_off, dcs, thr;
  uint32_t flags, key0, key1;
};

/* per-lane generator state: the complete carried state of flowgen_shimmer.c's loop
 * (DeltaPer[0], DeltaShimmer[0], T4, T, CountSamples) plus the draw counter */
struct VsGen {
  uint32_t d;
  float dp0, ds0;
  int T4, T, g, wpos, cyc;
  /* the next cycle's period / amplitude / closing speed once its jitter, shimmer and Knew draws
   * are made (vs_cycle_scalars) and before its samples are written (vs_cycle_emit) */
  float amp_next, S_next, K_next;
  bool pend;
  int posted; /* three-role kernel: orders this lane has handed to the noise wavefront */
};

/* the Philox block the scalar draws of one cycle come from (local to vs_cycle_scalars) */
struct VsBlk {
  uint32_t idx, b0, b1, b2, b3;
};

/* next draw of the lane's sequential stream = what random() returns in the shimmed reference.
 * Called under the EXEC mask of the lanes that draw. */
__device__ __forceinline__ uint32_t vs_draw(const VsCfg &c, VsGen &s, VsBlk &k)
{
  const uint32_t b = s.d >> 2;
  if (b != k.idx) {
    vs_philox(b, c.key0, c.key1, k.b0, k.b1, k.b2, k.b3);
    k.idx = b;
  }
  /* word (d & 3) of the cached block; written as 64-bit select + shift so that the compiler
   * does not turn a four-way select into an indexed scratch array */
  const uint64_t q0 = (uint64_t)k.b0 | ((uint64_t)k.b1 << 32);
  const uint64_t q1 = (uint64_t)k.b2 | ((uint64_t)k.b3 << 32);
  const uint64_t q = (s.d & 2u) ? q1 : q0;
  const uint32_t v = (uint32_t)(q >> ((s.d & 1u) * 32u));
  s.d += 1u;
  return v >> 1;
}

/* (1.0*random())/RAND_MAX of flowgen_shimmer.c:325,387,398 for a draw r in [0, 2^31): the
 * correctly rounded quotient r / 2147483647 from one multiply and two fused multiply-adds
 * (Markstein's final-step form: q0 = r*inv is within one ulp, the residual r - q0*d is exact,
 * inv = RN(1/d)).  Equality with IEEE division is verified EXHAUSTIVELY over all 2^31 draws,
 * on the CPU by tests/test_div_shortcut.py and on the device by vs_ctx_selftest(). */
__device__ __forceinline__ double vs_unit_of_draw(uint32_t r)
{
  const double d = 2147483647.0;
  const double inv = 0x1.00000002p-31;
  const double x = (double)r;
  const double q0 = x * inv;
  const double e = __builtin_fma(-q0, d, x);
  return __builtin_fma(e, inv, q0);
}

/* (signed short) of a double, as gcc/x86-64 converts it: through int32, low 16 bits */
__device__ __forceinline__ int vs_short_of(double v) { return (int)(int16_t)(int)v; }

/* round2int() of vowel_new.c:413-427:
 *     dec = x - floor(x); if (dec > 0.5) x = x + 1; clamp x to [-32767, 32767]; return floor(x)
 * dec comes from V_FRACT_F64: x - floor(x) is exact for every double except -1 < x < 0, where both
 * forms round x + 1 to nearest; the instruction only differs in returning the largest double
 * below 1 where the subtraction rounds up to 1.0 (tiny negative x), and both are > 0.5 there.
 * The "+1" stays a double addition (it is part of the reference's rounding sequence: the
 * reference returns 1 for x = -1e-20); the clamp moves behind the floor into integers, which
 * gives the same result for every finite x: floor is monotone, floor(+-32767) = +-32767, and
 * v_cvt_i32_f64 saturates beyond int32.  x is never NaN (stable filter, int16 input). */
__device__ __forceinline__ int vs_round2int(double x)
{
  const double dec = __builtin_amdgcn_fract(x);
  x = x + ((dec > 0.5) ? 1.0 : 0.0); /* x + 0.0 only turns -0.0 into +0.0; both floor to 0 */
  const int v = (int)floor(x);
  return (v > 32767) ? 32767 : ((v < -32767) ? -32767 : v);
}

/*
 * round2int() without its double rounding: ceil(x - 0.5), clamped.  This is round2int(x) for every
 * double x EXCEPT the ones for which the reference's "x = x + 1" rounds up to an integer although
 * x lies just below it -- the quirk set
 *     Q1 = [-2^-54, -0)                    (x + 1 rounds to 1.0: the reference returns 1, not 0)
 *     Q2 = { 2^m - 2^(m-53), m = 0..51 }   (mantissa all ones: x + 1 is a tie that rounds up)
 * (for |x| >= 1 both x - 0.5 and x + 1 are exact or round without reaching an integer; the interval
 * (-1, 1) is gone through case by case in tests/test_round2int.py and on the device by
 * vs_ctx_selftest [3]).  Every member of Q1 has a high word in [0x80000000, 0xBC900000] and every
 * member of Q2 a low word of 0xFFFFFFFF, so a super-step keeps the signed minimum of the high words
 * and the unsigned maximum of the low words of its 24 arguments (one V_MIN3 / V_MAX3 per two
 * samples) and, when either hits, rounds that super-step again with vs_round2int() -- outputs are
 * not fed back, so nothing else has to be redone.  Three fp64 instructions per sample instead of five.
 */
__device__ __forceinline__ int vs_round2int_half_down_unclamped(double x) { return (int)ceil(x - 0.5); }
__device__ __forceinline__ int vs_round2int_half_down(double x)
{
  const int v = vs_round2int_half_down_unclamped(x);
  return (v > 32767) ? 32767 : ((v < -32767) ? -32767 : v);
}
/* two rounded values, clamped to [-32767, 32767] and packed: V_CVT_PK_I16_I32 saturates to int16,
 * V_PK_MAX_I16 lifts -32768 to the reference's -32767 (vowel_new.c:423-424) -- two instructions for two
 * samples instead of two V_MED3_I32 and a V_PERM_B32 */
typedef short vs_i16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t vs_clamp_pack16(int lo, int hi)
{
  const vs_i16x2 p = __builtin_amdgcn_cvt_pk_i16(lo, hi);
  const vs_i16x2 floor_ = {(short)-32767, (short)-32767};
  const vs_i16x2 q = __builtin_elementwise_max(p, floor_);
  return __builtin_bit_cast(uint32_t, q);
}
#define VS_R2I_Q1_HI ((int)0xBC900000) /* high word of -2^-54, as a signed integer */

/* (int)sqrt(v) of the reference (flowgen_shimmer.c:382) for a float-valued v >= 0: the
 * device sqrt only seeds an exact integer search, so its last-bit rounding cannot matter */
__device__ __forceinline__ int vs_isqrt_floor(double v)
{
  int s = (int)sqrt(v);
  if (s < 0) s = 0;
  while ((double)(s + 1) * (double)(s + 1) <= v) ++s;
  while (s > 0 && (double)s * (double)s > v) --s;
  return s;
}

/*
 * LDS ring layout: int16 ring[C + 8][64] -- slot-major, lane l owns column l, one slot of all 64
 * lanes is 128 contiguous bytes, so a ds_write_b16 / ds_read_i16 of a wavefront touches every bank
 * once (two lanes per 4-byte bank, same dword).  Slots [C, C + 8) are the trash rows: where lanes that
 * must not emit send their 8-sample trips.  (A lane-major layout -- 16 contiguous bytes per lane and
 * 8 slots -- lets the filter side read 8 samples per LDS instruction, but single-sample writes then
 * hit every bank eight times over and the generator alone runs 15 % longer; measured in round 3,
 * profiles/r03_kernel_experiments.txt.)
 */
/* utterances per group = lanes that own a ring column.  64, a whole wavefront -- except in the second
 * build of this file (vs_kernels_narrow.o, -DVS_GROUP_LANES=16), which exists for periods too long for a
 * 64-column ring (e.g. 48 kHz at F0 = 50 Hz with jitter: 1152 samples): a quarter of the columns, four
 * times the slots in the same LDS, three quarters of the wavefront idle.  Slow, and only ever used for
 * plans the wide ring cannot take (vs_plan_create); the reference accepts such rates
 * (flowgen_shimmer.c:535-540) and sizes its buffer by the period (fg:569). */
#ifndef VS_GROUP_LANES
#define VS_GROUP_LANES VS_WAVE
#else
#define vs_synth_kernel vs_synth_kernel_narrow /* the two builds end up in one library: no shared kernel names */
#endif
#define VS_RING_STEP (VS_GROUP_LANES * 2) /* bytes from a lane's slot s to its slot s + 1 */

/* int16 index of ring slot `slot` (0 <= slot < C + 8; slots [C, C + 8) are the trash rows) */
__device__ __forceinline__ int vs_ring_idx(int slot, int lane) { return slot * VS_GROUP_LANES + lane; }

/* int16 index of sample i of the cycle being written: the cycle starts at slot wpos and wraps
 * at most once (wpos < C, i < C + VS_TRASH_ROWS). */
__device__ __forceinline__ int vs_ring_at(int wpos, int C, int i, int lane)
{
  /* slot = (wpos + i) mod C for wpos + i < 2C, as min(s, s - C) on unsigned (two instructions) */
  const unsigned sl = (unsigned)(wpos + i);
  const unsigned wr = sl - (unsigned)C;
  return vs_ring_idx((int)((sl < wr) ? sl : wr), lane);
}

/* Eight consecutive ring slots of a lane that start ANYWHERE (the noise trips follow the Philox
 * blocks, not the ring): the run wraps at most once, after kw slots.  A sample costs one compare,
 * one select and the store (the slot offset W*128 sits in the store's immediate). */
typedef __attribute__((address_space(3))) char vs_lds_char;
typedef __attribute__((address_space(3))) int16_t vs_lds_i16;
struct VsRun8 {
  char *A, *B; /* LDS address of slot 0 of the run before / after the wrap */
  int kw;      /* slots before the wrap (>= 8: none in this run) */
};
__device__ __forceinline__ VsRun8 vs_run8(int16_t *ring, int wpos, int C, int i0, int lane)
{
  const unsigned sl = (unsigned)(wpos + i0);
  const unsigned wr = sl - (unsigned)C;
  const unsigned a0 = (sl < wr) ? sl : wr;
  VsRun8 r;
  r.kw = C - (int)a0;
  r.A = (char *)ring + (a0 * (unsigned)VS_RING_STEP + (unsigned)(2 * lane));
  r.B = r.A - (unsigned)C * (unsigned)VS_RING_STEP;
  return r;
}
/* the run of the next 8 slots */
__device__ __forceinline__ void vs_run8_advance(VsRun8 &r, int C)
{
  r.kw -= 8;
  r.A += 8 * VS_RING_STEP;
  const bool wrapped = r.kw <= 0; /* the whole of the next run lies behind the wrap */
  r.A = wrapped ? r.B + 8 * VS_RING_STEP : r.A;
  r.kw = wrapped ? r.kw + C : r.kw;
  r.B = r.A - (unsigned)C * (unsigned)VS_RING_STEP;
}
/* all eight stores of a lane go to the trash rows [C, C + 8) */
__device__ __forceinline__ VsRun8 vs_run8_trash(int16_t *ring, int C, int lane)
{
  VsRun8 r;
  r.kw = 8;
  r.A = (char *)ring + ((unsigned)C * (unsigned)VS_RING_STEP + (unsigned)(2 * lane));
  r.B = r.A;
  return r;
}
/* a lane's run if it still emits, the trash rows otherwise (field by field: a select of whole
 * structs makes the compiler index them in scratch memory) */
__device__ __forceinline__ VsRun8 vs_run8_or_trash(bool emit, int16_t *ring, int wpos, int C, int i0, int lane)
{
  const VsRun8 a = vs_run8(ring, wpos, C, i0, lane);
  const VsRun8 t = vs_run8_trash(ring, C, lane);
  VsRun8 r;
  r.kw = emit ? a.kw : t.kw;
  r.A = emit ? a.A : t.A;
  r.B = emit ? a.B : t.B;
  return r;
}

/* The eight store addresses of a run: pw[w] = (w < kw) ? A : B as LDS byte addresses.  Written out as
 * eight compares into eight SGPR pairs and then eight selects: on gfx950 a VALU instruction must not read
 * a mask within two wait states of the VALU instruction that wrote it, and left to itself the compiler
 * pairs every compare with its select and puts an s_nop between them -- eight instructions per trip that
 * do nothing, each at the price of one that does (ubench5). */
__device__ __forceinline__ uint32_t vs_lds_addr(const char *p) { return (uint32_t)(uintptr_t)(const vs_lds_char *)p; }
__device__ __forceinline__ void vs_wrap_select8(uint32_t A, uint32_t B, int kw, uint32_t (&pw)[8])
{
  unsigned long long m0, m1, m2, m3, m4, m5, m6, m7;
  asm volatile("v_cmp_lt_i32_e64 %8, 0, %18\n\t"
               "v_cmp_lt_i32_e64 %9, 1, %18\n\t"
               "v_cmp_lt_i32_e64 %10, 2, %18\n\t"
               "v_cmp_lt_i32_e64 %11, 3, %18\n\t"
               "v_cmp_lt_i32_e64 %12, 4, %18\n\t"
               "v_cmp_lt_i32_e64 %13, 5, %18\n\t"
               "v_cmp_lt_i32_e64 %14, 6, %18\n\t"
               "v_cmp_lt_i32_e64 %15, 7, %18\n\t"
               "v_cndmask_b32_e64 %0, %17, %16, %8\n\t"
               "v_cndmask_b32_e64 %1, %17, %16, %9\n\t"
               "v_cndmask_b32_e64 %2, %17, %16, %10\n\t"
               "v_cndmask_b32_e64 %3, %17, %16, %11\n\t"
               "v_cndmask_b32_e64 %4, %17, %16, %12\n\t"
               "v_cndmask_b32_e64 %5, %17, %16, %13\n\t"
               "v_cndmask_b32_e64 %6, %17, %16, %14\n\t"
               "v_cndmask_b32_e64 %7, %17, %16, %15"
               : "=&v"(pw[0]), "=&v"(pw[1]), "=&v"(pw[2]), "=&v"(pw[3]), "=&v"(pw[4]), "=&v"(pw[5]), "=&v"(pw[6]), "=&v"(pw[7]),
                 "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3), "=&s"(m4), "=&s"(m5), "=&s"(m6), "=&s"(m7)
               : "v"(A), "v"(B), "v"(kw));
}
template <int W>
__device__ __forceinline__ void vs_lds_store16(uint32_t addr, int v)
{
  *(vs_lds_i16 *)(uintptr_t)(addr + (uint32_t)(W * VS_RING_STEP)) = (int16_t)v;
}

template <int W>
__device__ __forceinline__ void vs_run8_store(const VsRun8 &r, int v)
{
  char *p = (W < r.kw) ? r.A : r.B;
  *(int16_t *)(p + W * VS_RING_STEP) = (int16_t)v;
}
/* the same, but to the trash rows (address trashA of row C) unless ok */
template <int W>
__device__ __forceinline__ void vs_run8_store_if(const VsRun8 &r, char *trashA, bool ok, int v)
{
  char *p = (W < r.kw) ? r.A : r.B;
  p = ok ? p : trashA;
  *(int16_t *)(p + W * VS_RING_STEP) = (int16_t)v;
}
__device__ __forceinline__ void vs_run8_store_all(const VsRun8 &r, const int (&x)[8])
{
  uint32_t pw[8];
  vs_wrap_select8(vs_lds_addr(r.A), vs_lds_addr(r.B), r.kw, pw);
  vs_lds_store16<0>(pw[0], x[0]); vs_lds_store16<1>(pw[1], x[1]); vs_lds_store16<2>(pw[2], x[2]); vs_lds_store16<3>(pw[3], x[3]);
  vs_lds_store16<4>(pw[4], x[4]); vs_lds_store16<5>(pw[5], x[5]); vs_lds_store16<6>(pw[6], x[6]); vs_lds_store16<7>(pw[7], x[7]);
}

#endif
