// Micro-benchmark 5 (round 3): what ONE instruction of each type used by the fused kernel costs when the
// SIMD is saturated.  W wavefronts per SIMD (W = 1..4) all run the same stream of one instruction type on
// four independent registers; reported: cycles of the shader clock per instruction per SIMD (clock taken
// from the launch time in ns and s_memtime, printed), i.e. the pipe cost a kernel at ~90 % VALU busy pays.
// Build: hipcc -O3 --offload-arch=gfx950 -o ubench5 ubench5.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

enum {
  O_MUL_F64 = 0, O_ADD_F64, O_FMA_F64, O_CHAIN, O_MAD_U64, O_MUL_LO, O_BITOP3, O_XOR, O_LSHR, O_CVT_F64_U32, O_CVT_I32_F64,
  O_CEIL_F64, O_CNDMASK, O_MUL24, O_CVT_F32_I32, O_ADD_F32, O_CVT_PK, O_PK_MAX, O_MIN_I32, O_MOV64, O_DSW16, O_DSW128,
  O_DSR64, O_DSR128, O_SALU, O_MIX_F64_XOR, O_MIX_F64_MAD, O_MIX_F64_SALU,
  O_CND_SGPR, O_CMP_CND, O_CMP_VCC, O_CMP_SGPR, O_CMP_CND_SGPR, O_MAX_I32, O_MED3_I32, O_CND_AFTER_SMOV, O_ADD_U32, O_ADD_CO, O_MIX_F64_WAITCNT, O_MIX_F64_NOP, O_MIX_F64_3WAIT, O_COUNT
};
static const char *op_name[O_COUNT] = {
  "v_mul_f64", "v_add_f64", "v_fma_f64", "mul+add f64 dependent chain", "v_mad_u64_u32", "v_mul_lo_u32", "v_bitop3_b32", "v_xor_b32",
  "v_lshrrev_b32", "v_cvt_f64_u32", "v_cvt_i32_f64", "v_ceil_f64", "v_cndmask_b32", "v_mul_i32_i24", "v_cvt_f32_i32",
  "v_add_f32", "v_cvt_pk_i16_i32", "v_pk_max_i16", "v_min_i32", "v_mov_b64 (pair)", "ds_write_b16", "ds_write_b128",
  "ds_read_b64", "ds_read_b128", "s_add_u32", "1 v_mul_f64 : 1 v_xor_b32", "1 v_mul_f64 : 1 v_mad_u64_u32", "1 v_mul_f64 : 1 s_add_u32",
  "v_cndmask_b32 (sgpr pair mask)", "v_cmp vcc + v_cndmask vcc", "v_cmp_lt_u32 -> vcc", "v_cmp_lt_u32 -> sgpr pair", "v_cmp sgpr + v_cndmask sgpr",
  "v_max_i32", "v_med3_i32", "v_cndmask_b32 vcc (vcc from s_mov)", "v_add_u32", "v_add_co_u32 (writes vcc)", "1 v_mul_f64 : 1 s_waitcnt (nothing pending)", "1 v_mul_f64 : 1 s_nop 0",
  "3 v_mul_f64 : 1 s_waitcnt"};

struct Regs {
  double a0, a1, a2, a3, b0, b1;
  unsigned u0, u1, u2, u3, w0;
  unsigned long long q0, q1, q2, q3;
  unsigned la, lb;
  unsigned long long m64;
  __attribute__((ext_vector_type(4))) unsigned v4;
};


template <int O>
__device__ __forceinline__ void body(Regs &r, unsigned &sacc)
{
  if (O == O_MUL_F64) {
    REP16(asm volatile("v_mul_f64 %0, %0, %4\n\tv_mul_f64 %1, %1, %4\n\tv_mul_f64 %2, %2, %4\n\tv_mul_f64 %3, %3, %4" : "+v"(r.a0), "+v"(r.a1), "+v"(r.a2), "+v"(r.a3) : "v"(r.b0));)
  } else if (O == O_ADD_F64) {
    REP16(asm volatile("v_add_f64 %0, %0, %4\n\tv_add_f64 %1, %1, %4\n\tv_add_f64 %2, %2, %4\n\tv_add_f64 %3, %3, %4" : "+v"(r.a0), "+v"(r.a1), "+v"(r.a2), "+v"(r.a3) : "v"(r.b0));)
  } else if (O == O_FMA_F64) {
    REP16(asm volatile("v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %4, %5\n\tv_fma_f64 %3, %3, %4, %5" : "+v"(r.a0), "+v"(r.a1), "+v"(r.a2), "+v"(r.a3) : "v"(r.b0), "v"(r.b1));)
  } else if (O == O_CHAIN) {
    REP16(asm volatile("v_mul_f64 %1, %2, %3\n\tv_add_f64 %0, %0, -%1\n\tv_mul_f64 %1, %2, %4\n\tv_add_f64 %0, %0, -%1" : "+v"(r.a0), "+v"(r.a1) : "v"(r.a2), "v"(r.b0), "v"(r.b1));)
  } else if (O == O_MAD_U64) {
    REP16(asm volatile("v_mad_u64_u32 %0, vcc, %4, %8, 0\n\tv_mad_u64_u32 %1, vcc, %5, %8, 0\n\tv_mad_u64_u32 %2, vcc, %6, %8, 0\n\tv_mad_u64_u32 %3, vcc, %7, %8, 0"
                       : "=&v"(r.q0), "=&v"(r.q1), "=&v"(r.q2), "=&v"(r.q3) : "v"(r.u0), "v"(r.u1), "v"(r.u2), "v"(r.u3), "v"(r.w0) : "vcc");)
  } else if (O == O_MUL_LO) {
    REP16(asm volatile("v_mul_lo_u32 %0, %0, %4\n\tv_mul_lo_u32 %1, %1, %4\n\tv_mul_lo_u32 %2, %2, %4\n\tv_mul_lo_u32 %3, %3, %4" : "+v"(r.u0), "+v"(r.u1), "+v"(r.u2), "+v"(r.u3) : "v"(r.w0));)
  } else if (O == O_BITOP3) {
    REP16(asm volatile("v_bitop3_b32 %0, %0, %4, %1 bitop3:0x96\n\tv_bitop3_b32 %1, %1, %4, %2 bitop3:0x96\n\tv_bitop3_b32 %2, %2, %4, %3 bitop3:0x96\n\tv_bitop3_b32 %3, %3, %4, %0 bitop3:0x96" : "+v"(r.u0), "+v"(r.u1), "+v"(r.u2), "+v"(r.u3) : "v"(r.w0));)
  } else if (O == O_XOR) {
    REP16(asm volatile("v_xor_b32 %0, %0, %4\n\tv_xor_b32 %1, %1, %4\n\tv_xor_b32 %2, %2, %4\n\tv_xor_b32 %3, %3, %4" : "+v"(r.u0), "+v"(r.u1), "+v"(r.u2), "+v"(r.u3) : "v"(r.w0));)
  } else if (O == O_LSHR) {
    REP16(asm volatile("v_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %1, 1, %1\n\tv_lshrrev_b32 %2, 1, %2\n\tv_lshrrev_b32 %3, 1, %3" : "+v"(r.u0), "+v"(r.u1), "+v"(r.u2), "+v"(r.u3));)
  } else if (O == O_CVT_F64_U32) {
    REP16(asm volatile("v_cvt_f64_u32 %0, %4\n\tv_cvt_f64_u32 %1, %5\n\tv_cvt_f64_u32 %2, %6\n\tv_cvt_f64_u32 %3, %7" : "=&v"(r.a0), "=&v"(r.a1), "=&v"(r.a2), "=&v"(r.a3) : "v"(r.u0), "v"(r.u1), "v"(r.u2), "v"(r.u3));)
  } else if (O == O_CVT_I32_F64) {
    REP16(asm volatile("v_cvt_i32_f64 %0, %4\n\tv_cvt_i32_f64 %1, %5\n\tv_cvt_i32_f64 %2, %6\n\tv_cvt_i32_f64 %3, %7" : "=&v"(r.u0), "=&v"(r.u1), "=&v"(r.u2), "=&v"(r.u3) : "v"(r.a0), "v"(r.a1), "v"(r.a2), "v"(r.a3));)
  } else if (O == O_CEIL_F64) {
    REP16(asm volatile("v_ceil_f64 %0, %0\n\tv_ceil_f64 %1, %1\n\tv_ceil_f64 %2, %2\n\tv_ceil_f64 %3, %3" : "+v"(r.a0), "+v"(r.a1), "+v"(r.a2), "+v"(r.a3));)
  } else if (O == O_CNDMASK) {
    REP16(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n\tv_cndmask_b32 %1, %1, %4, vcc\n\tv_cndmask_b32 %2, %2, %4, vcc\n\tv_cndmask_b32 %3, %3, %4, vcc" : "+v"(r.u0), "+v"(r.u1), "+v"(r.u2), "+v"(r.u3) : "v"(r.w0) : "vcc");)
  } else if (O == O_MUL24) {
    REP16(asm volatile("v_mul_i32_i24 %0, %0, %4\n\tv_mul_i32_i24 %1, %1, %4\n\tv_mul_i32_i24 %2, %2, %4\n\tv_mul_i32_i24 %3, %3, %4" : "+v"(r.u0), "+v"(r.u1), "+v"(r.u2), "+v"(r.u3) : "v"(r.w0));)
  } else if (O == O_CVT_F32_I32) {
    REP16(asm volatile("v_cvt_f32_i32 %0, %0\n\tv_cvt_f32_i32 %1, %1\n\tv_cvt_f32_i32 %2, %2\n\tv_cvt_f32_i32 %3, %3" : "+v"(r.u0), "+v"(r.u1), "+v"(r.u2), "+v"(r.u3));)
  } else if (O == O_ADD_F32) {
    REP16(asm volatile("v_add_f32 %0, %0, %4\n\tv_add_f32 %1, %1, %4\n\tv_add_f32 %2, %2, %4\n\tv_add_f32 %3, %3, %4" : "+v"(r.u0), "+v"(r.u1), "+v"(r.u2), "+v"(r.u3) : "v"(r.w0));)
  } else if (O == O_CVT_PK) {
    REP16(asm volatile("v_cvt_pk_i16_i32 %0, %0, %4\n\tv_cvt_pk_i16_i32 %1, %1, %4\n\tv_cvt_pk_i16_i32 %2, %2, %4\n\tv_cvt_pk_i16_i32 %3, %3, %4" : "+v"(r.u0), "+v"(r.u1), "+v"(r.u2), "+v"(r.u3) : "v"(r.w0));)
  } else if (O == O_PK_MAX) {
    REP16(asm volatile("v_pk_max_i16 %0, %0, %4\n\tv_pk_max_i16 %1, %1, %4\n\tv_pk_max_i16 %2, %2, %4\n\tv_pk_max_i16 %3, %3, %4" : "+v"(r.u0), "+v"(r.u1), "+v"(r.u2), "+v"(r.u3) : "v"(r.w0));)
  } else if (O == O_MIN_I32) {
    REP16(asm volatile("v_min_i32 %0, %0, %4\n\tv_min_i32 %1, %1, %4\n\tv_min_i32 %2, %2, %4\n\tv_min_i32 %3, %3, %4" : "+v"(r.u0), "+v"(r.u1), "+v"(r.u2), "+v"(r.u3) : "v"(r.w0));)
  } else if (O == O_MOV64) {
    REP16(asm volatile("v_mov_b64 %0, %4\n\tv_mov_b64 %1, %4\n\tv_mov_b64 %2, %4\n\tv_mov_b64 %3, %4" : "=&v"(r.a0), "=&v"(r.a1), "=&v"(r.a2), "=&v"(r.a3) : "v"(r.b0));)
  } else if (O == O_DSW16) {
    REP16(asm volatile("ds_write_b16 %0, %1\n\tds_write_b16 %0, %1 offset:128\n\tds_write_b16 %0, %1 offset:256\n\tds_write_b16 %0, %1 offset:384" : : "v"(r.la), "v"(r.u0) : "memory");)
  } else if (O == O_DSW128) {
    REP16(asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %1\n\tds_write_b128 %0, %1\n\tds_write_b128 %0, %1" : : "v"(r.lb), "v"(r.v4) : "memory");)
  } else if (O == O_DSR64) {
    REP16(asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:512\n\tds_read_b64 %2, %4 offset:1024\n\tds_read_b64 %3, %4 offset:1536\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r.a0), "=&v"(r.a1), "=&v"(r.a2), "=&v"(r.a3) : "v"(r.lb) : "memory");)
  } else if (O == O_DSR128) {
    REP16(asm volatile("ds_read_b128 %0, %1\n\tds_read_b128 %0, %1\n\tds_read_b128 %0, %1\n\tds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r.v4) : "v"(r.lb) : "memory");)
  } else if (O == O_SALU) {
    REP16(asm volatile("s_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 3\n\ts_add_u32 %0, %0, 5\n\ts_add_u32 %0, %0, 7" : "+s"(sacc) : : "scc");)
  } else if (O == O_MIX_F64_XOR) {
    REP16(asm volatile("v_mul_f64 %0, %0, %4\n\tv_xor_b32 %2, %2, %5\n\tv_mul_f64 %1, %1, %4\n\tv_xor_b32 %3, %3, %5" : "+v"(r.a0), "+v"(r.a1), "+v"(r.u0), "+v"(r.u1) : "v"(r.b0), "v"(r.w0));)
  } else if (O == O_MIX_F64_MAD) {
    REP16(asm volatile("v_mul_f64 %0, %0, %4\n\tv_mad_u64_u32 %2, vcc, %5, %6, 0\n\tv_mul_f64 %1, %1, %4\n\tv_mad_u64_u32 %3, vcc, %7, %6, 0" : "+v"(r.a0), "+v"(r.a1), "=&v"(r.q0), "=&v"(r.q1) : "v"(r.b0), "v"(r.u0), "v"(r.w0), "v"(r.u1) : "vcc");)
  } else if (O == O_MIX_F64_SALU) {
    REP16(asm volatile("v_mul_f64 %0, %0, %3\n\ts_add_u32 %2, %2, 1\n\tv_mul_f64 %1, %1, %3\n\ts_add_u32 %2, %2, 3" : "+v"(r.a0), "+v"(r.a1), "+s"(sacc) : "v"(r.b0) : "scc");)
  } else if (O == O_CND_SGPR) {
    REP16(asm volatile("v_cndmask_b32_e64 %0, %0, %4, %5\n\tv_cndmask_b32_e64 %1, %1, %4, %5\n\tv_cndmask_b32_e64 %2, %2, %4, %5\n\tv_cndmask_b32_e64 %3, %3, %4, %5" : "+v"(r.u0), "+v"(r.u1), "+v"(r.u2), "+v"(r.u3) : "v"(r.w0), "s"(r.m64));)
  } else if (O == O_CMP_CND) {
    REP16(asm volatile("v_cmp_lt_u32 vcc, %0, %2\n\tv_cndmask_b32 %0, %0, %2, vcc\n\tv_cmp_lt_u32 vcc, %1, %2\n\tv_cndmask_b32 %1, %1, %2, vcc" : "+v"(r.u0), "+v"(r.u1) : "v"(r.w0) : "vcc");)
  } else if (O == O_CMP_VCC) {
    REP16(asm volatile("v_cmp_lt_u32 vcc, %0, %2\n\tv_cmp_lt_u32 vcc, %1, %2\n\tv_cmp_lt_u32 vcc, %0, %2\n\tv_cmp_lt_u32 vcc, %1, %2" : : "v"(r.u0), "v"(r.u1), "v"(r.w0) : "vcc");)
  } else if (O == O_CMP_SGPR) {
    REP16(asm volatile("v_cmp_lt_u32_e64 %0, %1, %3\n\tv_cmp_lt_u32_e64 %0, %2, %3\n\tv_cmp_lt_u32_e64 %0, %1, %3\n\tv_cmp_lt_u32_e64 %0, %2, %3" : "=&s"(r.m64) : "v"(r.u0), "v"(r.u1), "v"(r.w0));)
  } else if (O == O_CMP_CND_SGPR) {
    REP16(asm volatile("v_cmp_lt_u32_e64 %2, %0, %3\n\tv_cndmask_b32_e64 %0, %0, %3, %2\n\tv_cmp_lt_u32_e64 %2, %1, %3\n\tv_cndmask_b32_e64 %1, %1, %3, %2" : "+v"(r.u0), "+v"(r.u1), "=&s"(r.m64) : "v"(r.w0));)
  } else if (O == O_MAX_I32) {
    REP16(asm volatile("v_max_i32 %0, %0, %4\n\tv_max_i32 %1, %1, %4\n\tv_max_i32 %2, %2, %4\n\tv_max_i32 %3, %3, %4" : "+v"(r.u0), "+v"(r.u1), "+v"(r.u2), "+v"(r.u3) : "v"(r.w0));)
  } else if (O == O_MED3_I32) {
    REP16(asm volatile("v_med3_i32 %0, %0, %4, %5\n\tv_med3_i32 %1, %1, %4, %5\n\tv_med3_i32 %2, %2, %4, %5\n\tv_med3_i32 %3, %3, %4, %5" : "+v"(r.u0), "+v"(r.u1), "+v"(r.u2), "+v"(r.u3) : "v"(r.w0), "v"(r.la));)
  } else if (O == O_CND_AFTER_SMOV) {
    asm volatile("s_mov_b64 vcc, %0" : : "s"(r.m64) : "vcc");
    REP16(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n\tv_cndmask_b32 %1, %1, %4, vcc\n\tv_cndmask_b32 %2, %2, %4, vcc\n\tv_cndmask_b32 %3, %3, %4, vcc" : "+v"(r.u0), "+v"(r.u1), "+v"(r.u2), "+v"(r.u3) : "v"(r.w0));)
  } else if (O == O_ADD_U32) {
    REP16(asm volatile("v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4" : "+v"(r.u0), "+v"(r.u1), "+v"(r.u2), "+v"(r.u3) : "v"(r.w0));)
  } else if (O == O_ADD_CO) {
    REP16(asm volatile("v_add_co_u32 %0, vcc, %0, %4\n\tv_add_co_u32 %1, vcc, %1, %4\n\tv_add_co_u32 %2, vcc, %2, %4\n\tv_add_co_u32 %3, vcc, %3, %4" : "+v"(r.u0), "+v"(r.u1), "+v"(r.u2), "+v"(r.u3) : "v"(r.w0) : "vcc");)
  } else if (O == O_MIX_F64_WAITCNT) {
    REP16(asm volatile("v_mul_f64 %0, %0, %2\n\ts_waitcnt lgkmcnt(0)\n\tv_mul_f64 %1, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "+v"(r.a0), "+v"(r.a1) : "v"(r.b0));)
  } else if (O == O_MIX_F64_NOP) {
    REP16(asm volatile("v_mul_f64 %0, %0, %2\n\ts_nop 0\n\tv_mul_f64 %1, %1, %2\n\ts_nop 0" : "+v"(r.a0), "+v"(r.a1) : "v"(r.b0));)
  } else if (O == O_MIX_F64_3WAIT) {
    REP16(asm volatile("v_mul_f64 %0, %0, %3\n\tv_mul_f64 %1, %1, %3\n\tv_mul_f64 %2, %2, %3\n\ts_waitcnt lgkmcnt(0)" : "+v"(r.a0), "+v"(r.a1), "+v"(r.a2) : "v"(r.b0));)
  }
}

// out[wave] = ticks for iters x 64 instructions
template <int O>
__global__ void __launch_bounds__(1024) k_one(unsigned long long *out, double *sink, int iters)
{
  __shared__ __attribute__((aligned(16))) unsigned lds[64 * 16 * 16 / 4 * 4];
  const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  Regs r;
  r.a0 = threadIdx.x * 1.0001 + 1.0; r.a1 = r.a0 + 1.5; r.a2 = r.a0 + 2.5; r.a3 = r.a0 + 3.5;
  r.b0 = 0.999999; r.b1 = 1.000001;
  r.u0 = threadIdx.x * 2654435761u + 7u; r.u1 = r.u0 ^ 0x9E3779B9u; r.u2 = r.u1 * 3u; r.u3 = r.u2 + 11u; r.w0 = r.u0 + 1u;
  r.q0 = r.u0; r.q1 = r.u1; r.q2 = r.u2; r.q3 = r.u3;
  r.la = wave * 1024u + lane * 2u;            // ds_write_b16: 64 lanes x 2 bytes, +offsets up to 384 (+2): inside the wave's 1024 bytes
  r.lb = wave * 4096u + lane * 16u;           // 16-byte accesses: 64 x 16 = 1024 bytes (+1536 offset + 8): inside the wave's 4096 bytes
  r.v4 = {r.u0, r.u1, r.u2, r.u3};
  r.m64 = __builtin_amdgcn_readfirstlane(wave) * 0x0123456789ABCDEFull + 0x5555AAAA5555AAAAull;
  lds[threadIdx.x] = r.u0;
  __syncthreads();
  unsigned sacc = __builtin_amdgcn_readfirstlane(wave);
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) body<O>(r, sacc);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (lane == 0) out[blockIdx.x * 16 + wave] = t1 - t0;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = r.a0 + r.a1 + r.a2 + r.a3 + r.u0 + r.u1 + r.u2 + r.u3 + (double)r.q0 + (double)r.q1 +
                                                (double)r.q2 + (double)r.q3 + r.v4.x + sacc + (double)r.m64 + lds[(threadIdx.x * 7u) & 1023u];
}

typedef void (*kern_t)(unsigned long long *, double *, int);

template <int O>
static void fill(kern_t *t) { t[O] = k_one<O>; fill<O + 1>(t); }
template <>
void fill<O_COUNT>(kern_t *) {}

int main()
{
  kern_t fn[O_COUNT];
  fill<0>(fn);
  const int iters = 1000;
  unsigned long long *d_out;
  double *d_sink;
  (void)hipMalloc(&d_out, 256 * 16 * sizeof(unsigned long long));
  (void)hipMalloc(&d_sink, 256 * 1024 * sizeof(double));
  std::vector<unsigned long long> h(256 * 16);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  printf("one workgroup per CU, W wavefronts per SIMD, every wavefront runs %d x 64 instructions of one type (four independent registers)\n", iters);
  printf("columns: s_memtime ticks per instruction per SIMD at W = 1, 2, 3, 4; launch ns per instruction per SIMD at W = 4; ticks per ns\n");
  printf("%-34s %8s %8s %8s %8s %10s %8s\n", "instruction", "W=1", "W=2", "W=3", "W=4", "ns (W=4)", "tick/ns");
  for (int o = 0; o < O_COUNT; ++o) {
    double col[5] = {0, 0, 0, 0, 0}, ns4 = 0;
    for (int W = 1; W <= 4; ++W) {
      hipLaunchKernelGGL(fn[o], dim3(256), dim3(256 * W), 0, 0, d_out, d_sink, iters);
      (void)hipMemset(d_out, 0, 256 * 16 * sizeof(unsigned long long));
      (void)hipEventRecord(e0, 0);
      hipLaunchKernelGGL(fn[o], dim3(256), dim3(256 * W), 0, 0, d_out, d_sink, iters);
      (void)hipEventRecord(e1, 0);
      (void)hipDeviceSynchronize();
      float ms = 0;
      (void)hipEventElapsedTime(&ms, e0, e1);
      (void)hipMemcpy(h.data(), d_out, 256 * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
      double t = 0;
      for (int wg = 0; wg < 256; ++wg)
        for (int w = 0; w < 4 * W; ++w) t += (double)h[wg * 16 + w];
      t /= 256.0 * 4 * W;                                   // ticks of one wavefront for iters x 64 instructions
      col[W] = t / (iters * 64.0) / W;                      // per instruction per SIMD
      if (W == 4) ns4 = ms * 1e6 / (iters * 64.0) / W;
    }
    printf("%-34s %8.2f %8.2f %8.2f %8.2f %10.3f %8.2f\n", op_name[o], col[1], col[2], col[3], col[4], ns4, ns4 > 0 ? col[4] / ns4 : 0.0);
    fflush(stdout);
  }
  (void)hipFree(d_out);
  (void)hipFree(d_sink);
  return 0;
}
