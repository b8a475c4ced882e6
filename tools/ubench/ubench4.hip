// Micro-benchmark 4 (round 3): THREE wavefronts per SIMD (ubench3 with a third role) -- the situation of the wave-specialised
// kernel, where every SIMD hosts a filter wavefront (a dependent fp64 mul/add chain) and a generator
// wavefront (Philox multiplies, three-input xors, conversions, LDS writes).
// Questions: what does s_setprio do to the two instruction rates; what does each stream get while the
// other runs; what do LDS writes of 2 / 16 bytes per lane cost next to a busy fp64 wavefront; does an
// instruction with half of its lanes masked off occupy the pipe for less time.
// Layout as in vs_synth_ws_kernel: ONE workgroup of 8 wavefronts per CU, wavefronts 0-3 role A, 4-7 role B;
// a workgroup's wavefronts are dealt to the four SIMDs cyclically, so wavefront w and w+4 share a SIMD.
// Role A runs a fixed number of 64-instruction iterations and then raises a flag in LDS; role B runs until
// it sees the flag (one LDS read per iteration) and reports how far it got: both rates are those of the
// time in which BOTH were running.
// Build: hipcc -O3 --offload-arch=gfx950 -o ubench3 ubench3.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

enum { S_NONE = 0, S_F64, S_CHAIN, S_XOR, S_GEN, S_LDSW16, S_LDSW128, S_F64_HALF, S_F64_QUARTER, S_CHAIN_TOGGLE, S_FMA, S_OPEN, S_NOISE };

struct Regs {
  double a0, a1, a2, a3, b0, b1;
  unsigned u0, u1, u2, u3, w0;
  unsigned long long q0, q1;
  unsigned la, lb;
};

template <int S>
__device__ __forceinline__ void body(Regs &r)
{
  if (S == S_F64) {
    REP16(asm volatile("v_mul_f64 %0, %0, %4\n\tv_mul_f64 %1, %1, %4\n\tv_mul_f64 %2, %2, %4\n\tv_mul_f64 %3, %3, %4" : "+v"(r.a0), "+v"(r.a1), "+v"(r.a2), "+v"(r.a3) : "v"(r.b0));)
  } else if (S == S_FMA) {
    REP16(asm volatile("v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %4, %5\n\tv_fma_f64 %3, %3, %4, %5" : "+v"(r.a0), "+v"(r.a1), "+v"(r.a2), "+v"(r.a3) : "v"(r.b0), "v"(r.b1));)
  } else if (S == S_CHAIN) {
    REP16(asm volatile("v_mul_f64 %1, %2, %3\n\tv_add_f64 %0, %0, -%1\n\tv_mul_f64 %1, %2, %4\n\tv_add_f64 %0, %0, -%1" : "+v"(r.a0), "+v"(r.a1) : "v"(r.a2), "v"(r.b0), "v"(r.b1));)
  } else if (S == S_CHAIN_TOGGLE) {
    // the same chain, but the wavefront lowers its own priority for every other group of 8 instructions
    REP4(asm volatile("s_setprio 3\n\t"
                      "v_mul_f64 %1, %2, %3\n\tv_add_f64 %0, %0, -%1\n\tv_mul_f64 %1, %2, %4\n\tv_add_f64 %0, %0, -%1\n\t"
                      "v_mul_f64 %1, %2, %3\n\tv_add_f64 %0, %0, -%1\n\tv_mul_f64 %1, %2, %4\n\tv_add_f64 %0, %0, -%1\n\t"
                      "s_setprio 0\n\t"
                      "v_mul_f64 %1, %2, %3\n\tv_add_f64 %0, %0, -%1\n\tv_mul_f64 %1, %2, %4\n\tv_add_f64 %0, %0, -%1\n\t"
                      "v_mul_f64 %1, %2, %3\n\tv_add_f64 %0, %0, -%1\n\tv_mul_f64 %1, %2, %4\n\tv_add_f64 %0, %0, -%1"
                      : "+v"(r.a0), "+v"(r.a1) : "v"(r.a2), "v"(r.b0), "v"(r.b1));)
  } else if (S == S_XOR) {
    REP16(asm volatile("v_xor_b32 %0, %0, %4\n\tv_xor_b32 %1, %1, %4\n\tv_xor_b32 %2, %2, %4\n\tv_xor_b32 %3, %3, %4" : "+v"(r.u0), "+v"(r.u1), "+v"(r.u2), "+v"(r.u3) : "v"(r.w0));)
  } else if (S == S_GEN) {
    // Philox-like: two 32x32->64 multiplies, two three-input xors fed by them
    REP16(asm volatile("v_mad_u64_u32 %0, vcc, %2, %4, 0\n\tv_mad_u64_u32 %1, vcc, %3, %4, 0\n\t"
                       "v_bitop3_b32 %2, %2, %3, %4 bitop3:0x96\n\tv_bitop3_b32 %3, %3, %2, %4 bitop3:0x96"
                       : "=&v"(r.q0), "=&v"(r.q1), "+v"(r.u0), "+v"(r.u1) : "v"(r.w0) : "vcc");)
  } else if (S == S_LDSW16) {
    // 15 vector instructions per 2-byte LDS write: the generator's noise trips have about that ratio
    REP4(asm volatile("v_xor_b32 %0, %0, %4\n\tv_xor_b32 %1, %1, %4\n\tv_xor_b32 %2, %2, %4\n\tv_xor_b32 %3, %3, %4\n\t"
                      "v_xor_b32 %0, %0, %4\n\tv_xor_b32 %1, %1, %4\n\tv_xor_b32 %2, %2, %4\n\tv_xor_b32 %3, %3, %4\n\t"
                      "v_xor_b32 %0, %0, %4\n\tv_xor_b32 %1, %1, %4\n\tv_xor_b32 %2, %2, %4\n\tv_xor_b32 %3, %3, %4\n\t"
                      "v_xor_b32 %0, %0, %4\n\tv_xor_b32 %1, %1, %4\n\tv_xor_b32 %2, %2, %4\n\tds_write_b16 %5, %3"
                      : "+v"(r.u0), "+v"(r.u1), "+v"(r.u2), "+v"(r.u3) : "v"(r.w0), "v"(r.la) : "memory");)
  } else if (S == S_LDSW128) {
    // the same work, but one 16-byte write per 8 "samples" (63 vector instructions + one LDS write per iteration
    // would be 1:128; here 2 per iteration = 1:32, i.e. eight times fewer LDS instructions than S_LDSW16 x 2)
    asm volatile("s_nop 0" ::: "memory");
    REP4(asm volatile("v_xor_b32 %0, %0, %4\n\tv_xor_b32 %1, %1, %4\n\tv_xor_b32 %2, %2, %4\n\tv_xor_b32 %3, %3, %4\n\t"
                      "v_xor_b32 %0, %0, %4\n\tv_xor_b32 %1, %1, %4\n\tv_xor_b32 %2, %2, %4\n\tv_xor_b32 %3, %3, %4\n\t"
                      "v_xor_b32 %0, %0, %4\n\tv_xor_b32 %1, %1, %4\n\tv_xor_b32 %2, %2, %4\n\tv_xor_b32 %3, %3, %4\n\t"
                      "v_xor_b32 %0, %0, %4\n\tv_xor_b32 %1, %1, %4\n\tv_xor_b32 %2, %2, %4\n\tv_xor_b32 %3, %3, %4"
                      : "+v"(r.u0), "+v"(r.u1), "+v"(r.u2), "+v"(r.u3) : "v"(r.w0));)
    asm volatile("ds_write_b128 %0, %1" : : "v"(r.lb), "v"(*(__attribute__((ext_vector_type(4))) unsigned *)&r.u0) : "memory");
  } else if (S == S_OPEN) {
    // open-phase-like: per sample 4 fp64 (sub, mul, ceil, cvt), 3 for the power sum (mul24, cvt, add f32), compare+select, one LDS write
    REP4(asm volatile("v_add_f64 %0, -%0, 1.0\n\tv_mul_f64 %1, %1, %6\n\tv_ceil_f64 %0, %0\n\tv_cvt_i32_f64 %2, %1\n\t"
                      "v_mul_i32_i24 %3, %2, %2\n\tv_cvt_f32_i32 %4, %3\n\tv_add_f32 %5, %5, %4\n\tv_cmp_lt_u32 vcc, %2, %3\n\t"
                      "v_cndmask_b32 %3, %3, %2, vcc\n\tds_write_b16 %7, %3\n\t"
                      "v_add_f64 %0, -%0, 1.0\n\tv_mul_f64 %1, %1, %6\n\tv_ceil_f64 %0, %0\n\tv_cvt_i32_f64 %2, %1\n\t"
                      "v_mul_i32_i24 %3, %2, %2\n\tv_cvt_f32_i32 %4, %3"
                      : "+v"(r.a0), "+v"(r.a1), "+v"(r.u0), "+v"(r.u1), "+v"(r.u2), "+v"(r.u3) : "v"(r.b0), "v"(r.la) : "vcc", "memory");)
  } else if (S == S_NOISE) {
    // noise-trip-like: per 4 draws 10 x (2 mad_u64 + 2 bitop3) / 4 ... here 16 instructions = 6 mad, 6 bitop3, then shift/cvt/fma/cvt
    REP4(asm volatile("v_mad_u64_u32 %0, vcc, %2, %4, 0\n\tv_mad_u64_u32 %1, vcc, %3, %4, 0\n\t"
                      "v_bitop3_b32 %2, %2, %3, %4 bitop3:0x96\n\tv_bitop3_b32 %3, %3, %2, %4 bitop3:0x96\n\t"
                      "v_mad_u64_u32 %0, vcc, %2, %4, 0\n\tv_mad_u64_u32 %1, vcc, %3, %4, 0\n\t"
                      "v_bitop3_b32 %2, %2, %3, %4 bitop3:0x96\n\tv_bitop3_b32 %3, %3, %2, %4 bitop3:0x96\n\t"
                      "v_mad_u64_u32 %0, vcc, %2, %4, 0\n\tv_mad_u64_u32 %1, vcc, %3, %4, 0\n\t"
                      "v_bitop3_b32 %2, %2, %3, %4 bitop3:0x96\n\tv_bitop3_b32 %3, %3, %2, %4 bitop3:0x96\n\t"
                      "v_lshrrev_b32 %2, 1, %2\n\tv_cvt_f64_u32 %5, %2\n\tv_fma_f64 %5, %5, %6, %6\n\tv_cvt_i32_f64 %3, %5"
                      : "=&v"(r.q0), "=&v"(r.q1), "+v"(r.u0), "+v"(r.u1) : "v"(r.w0), "v"(r.a0), "v"(r.b0) : "vcc");)
  } else if (S == S_F64_HALF || S == S_F64_QUARTER) {
    REP16(asm volatile("v_mul_f64 %0, %0, %4\n\tv_mul_f64 %1, %1, %4\n\tv_mul_f64 %2, %2, %4\n\tv_mul_f64 %3, %3, %4" : "+v"(r.a0), "+v"(r.a1), "+v"(r.a2), "+v"(r.a3) : "v"(r.b0));)
  }
}

// out[wave] = {ticks, iterations done}; roles A (wavefronts 0-3), B (4-7), C (8-11): wavefront w, w+4, w+8 share a SIMD
template <int SA, int PA, int SB, int PB, int SC, int PC>
__global__ void __launch_bounds__(768) k_trio(unsigned long long *out, double *sink, int itersA)
{
  __shared__ __attribute__((aligned(16))) unsigned lds[64 * 16 * 12 + 16];
  volatile unsigned *flag = &lds[64 * 16 * 12];
  const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  const unsigned role = wave >> 2;
  Regs r;
  r.a0 = threadIdx.x * 1.0001 + 1.0; r.a1 = r.a0 + 1.5; r.a2 = r.a0 + 2.5; r.a3 = r.a0 + 3.5;
  r.b0 = 0.999999; r.b1 = 1.000001;
  r.u0 = threadIdx.x * 2654435761u + 7u; r.u1 = r.u0 ^ 0x9E3779B9u; r.u2 = r.u1 * 3u; r.u3 = r.u2 + 11u; r.w0 = r.u0 + 1u;
  r.q0 = r.u0; r.q1 = r.u1;
  r.la = wave * 1024u + lane * 2u;
  r.lb = (wave * 64u + lane) * 16u;
  if (threadIdx.x < 4) flag[threadIdx.x] = 0u;
  __syncthreads();
  unsigned long long t0 = 0, t1 = 0, done = 0;
  // the first role that is present is the one that runs a fixed count and raises the flag
  const int lead = (SA != S_NONE) ? 0 : ((SB != S_NONE) ? 1 : 2);
  if (role == 0) {
    if (SA == S_NONE) return;
    if (PA == 3) __builtin_amdgcn_s_setprio(3);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < itersA; ++it) body<SA>(r);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    done = (unsigned long long)itersA;
    if (lane == 0) flag[wave & 3] = 1u;
  } else if (role == 1) {
    if (SB == S_NONE) return;
    if (PB == 3) __builtin_amdgcn_s_setprio(3);
    if (PB == 1) __builtin_amdgcn_s_setprio(1);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    const int cap = (lead == 1) ? itersA : (1 << 24);
    for (int it = 0; it < cap; ++it) {
      body<SB>(r);
      done += 1;
      if (lead != 1 && flag[wave & 3] != 0u) break;
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (lead == 1 && lane == 0) flag[wave & 3] = 1u;
  } else {
    if (SC == S_NONE) return;
    if (PC == 3) __builtin_amdgcn_s_setprio(3);
    if (PC == 1) __builtin_amdgcn_s_setprio(1);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    const int cap = (lead == 2) ? itersA : (1 << 24);
    for (int it = 0; it < cap; ++it) {
      body<SC>(r);
      done += 1;
      if (lead != 2 && flag[wave & 3] != 0u) break;
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  }
  if (lane == 0) {
    const unsigned wv = blockIdx.x * 12 + wave;
    out[2 * wv] = t1 - t0;
    out[2 * wv + 1] = done;
  }
  sink[blockIdx.x * blockDim.x + threadIdx.x] = r.a0 + r.a1 + r.a2 + r.a3 + r.u0 + r.u1 + r.u2 + r.u3 + (double)r.q0 + (double)r.q1;
}

typedef void (*kern_t)(unsigned long long *, double *, int);
struct Case { const char *name; kern_t fn; };

int main()
{
  Case cases[] = {
      {"open alone                              ", k_trio<S_NONE, 0, S_OPEN, 0, S_NONE, 0>},
      {"noise alone                             ", k_trio<S_NONE, 0, S_NOISE, 0, S_NONE, 0>},
      {"open | noise                            ", k_trio<S_NONE, 0, S_OPEN, 0, S_NOISE, 0>},
      {"noise | open                            ", k_trio<S_NONE, 0, S_NOISE, 0, S_OPEN, 0>},
      {"noise | noise                           ", k_trio<S_NONE, 0, S_NOISE, 0, S_NOISE, 0>},
      {"open | open                             ", k_trio<S_NONE, 0, S_OPEN, 0, S_OPEN, 0>},
      {"xor | xor                               ", k_trio<S_NONE, 0, S_XOR, 0, S_XOR, 0>},
      {"chain prio3 | noise                     ", k_trio<S_CHAIN, 3, S_NOISE, 0, S_NONE, 0>},
      {"chain prio3 | open                      ", k_trio<S_CHAIN, 3, S_OPEN, 0, S_NONE, 0>},
      {"chain prio3 | open | noise              ", k_trio<S_CHAIN, 3, S_OPEN, 0, S_NOISE, 0>},
      {"chain prio3 | noise | open              ", k_trio<S_CHAIN, 3, S_NOISE, 0, S_OPEN, 0>},
      {"chain prio3 | open prio1 | noise prio1  ", k_trio<S_CHAIN, 3, S_OPEN, 1, S_NOISE, 1>},
      {"fma prio3 | open | noise                ", k_trio<S_FMA, 3, S_OPEN, 0, S_NOISE, 0>},
      {"chain prio0 | open | noise              ", k_trio<S_CHAIN, 0, S_OPEN, 0, S_NOISE, 0>},
  };
  const int iters = 2000;
  unsigned long long *d_out;
  double *d_sink;
  (void)hipMalloc(&d_out, 2 * 3072 * sizeof(unsigned long long));
  (void)hipMalloc(&d_sink, 256 * 768 * sizeof(double));
  std::vector<unsigned long long> h(2 * 3072);
  printf("one workgroup of 12 wavefronts per CU: wavefronts 0-3 = A, 4-7 = B, 8-11 = C; w, w+4, w+8 share a SIMD\n");
  printf("ticks per instruction of each wavefront while all run (64 instructions per iteration; the first role present runs a fixed count)\n");
  printf("%-40s %10s %10s %10s %12s\n", "case", "A", "B", "C", "SIMD total");
  for (auto &c : cases) {
    (void)hipMemset(d_out, 0, 2 * 3072 * sizeof(unsigned long long));
    hipLaunchKernelGGL(c.fn, dim3(256), dim3(768), 0, 0, d_out, d_sink, iters);
    (void)hipMemset(d_out, 0, 2 * 3072 * sizeof(unsigned long long));
    hipLaunchKernelGGL(c.fn, dim3(256), dim3(768), 0, 0, d_out, d_sink, iters);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h.data(), d_out, 2 * 3072 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double t[3] = {0, 0, 0}, n[3] = {0, 0, 0};
    for (int wg = 0; wg < 256; ++wg)
      for (int w = 0; w < 12; ++w) {
        t[w >> 2] += (double)h[2 * (wg * 12 + w)];
        n[w >> 2] += (double)h[2 * (wg * 12 + w) + 1] * 64.0;
      }
    double rate = 0, c3[3];
    for (int k = 0; k < 3; ++k) {
      c3[k] = n[k] > 0 ? t[k] / n[k] : 0;
      if (c3[k] > 0) rate += 1.0 / c3[k];
    }
    printf("%-40s %10.2f %10.2f %10.2f %12.2f\n", c.name, c3[0], c3[1], c3[2], rate > 0 ? 1.0 / rate : 0.0);
    fflush(stdout);
  }
  (void)hipFree(d_out);
  (void)hipFree(d_sink);
  return 0;
}
