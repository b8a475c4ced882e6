// Micro-benchmark 3 (round 3): TWO DIFFERENT wavefronts per SIMD -- the situation of the wave-specialised
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

enum { S_NONE = 0, S_F64, S_CHAIN, S_XOR, S_GEN, S_LDSW16, S_LDSW128, S_F64_HALF, S_F64_QUARTER, S_CHAIN_TOGGLE, S_FMA };

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
  } else if (S == S_F64_HALF || S == S_F64_QUARTER) {
    REP16(asm volatile("v_mul_f64 %0, %0, %4\n\tv_mul_f64 %1, %1, %4\n\tv_mul_f64 %2, %2, %4\n\tv_mul_f64 %3, %3, %4" : "+v"(r.a0), "+v"(r.a1), "+v"(r.a2), "+v"(r.a3) : "v"(r.b0));)
  }
}

// out[wave] = {ticks, iterations done}
template <int SA, int PA, int SB, int PB>
__global__ void __launch_bounds__(512) k_pair(unsigned long long *out, double *sink, int itersA)
{
  __shared__ __attribute__((aligned(16))) unsigned lds[64 * 16 * 8 + 16];
  volatile unsigned *flag = &lds[64 * 16 * 8];
  const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  const bool roleA = wave < 4;
  Regs r;
  r.a0 = threadIdx.x * 1.0001 + 1.0; r.a1 = r.a0 + 1.5; r.a2 = r.a0 + 2.5; r.a3 = r.a0 + 3.5;
  r.b0 = 0.999999; r.b1 = 1.000001;
  r.u0 = threadIdx.x * 2654435761u + 7u; r.u1 = r.u0 ^ 0x9E3779B9u; r.u2 = r.u1 * 3u; r.u3 = r.u2 + 11u; r.w0 = r.u0 + 1u;
  r.q0 = r.u0; r.q1 = r.u1;
  r.la = wave * 1024u + lane * 2u;   // bytes: 2 per lane, a 1 KiB strip per wavefront
  r.lb = (wave * 64u + lane) * 16u;  // bytes: 16 per lane
  if (threadIdx.x < 4) flag[threadIdx.x] = 0u;
  __syncthreads();
  unsigned long long t0, t1, done = 0;
  if (roleA) {
    if (SA == S_NONE) return;
    if (PA == 3) __builtin_amdgcn_s_setprio(3);
    if (PA == 1) __builtin_amdgcn_s_setprio(1);
    /* masked cases: an ordinary divergent branch, so that the compiler itself manages EXEC */
    const unsigned nact = (SA == S_F64_HALF) ? 32u : ((SA == S_F64_QUARTER) ? 16u : 64u);
    t0 = t1 = 0;
    if (lane < nact) {
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
      for (int it = 0; it < itersA; ++it) body<SA>(r);
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    }
    done = (unsigned long long)itersA;
    if (lane == 0) flag[wave] = 1u;  // wavefront w tells its SIMD partner w + 4
  } else {
    if (SB == S_NONE) return;
    if (PB == 3) __builtin_amdgcn_s_setprio(3);
    if (PB == 1) __builtin_amdgcn_s_setprio(1);
    const unsigned nact = (SB == S_F64_HALF) ? 32u : ((SB == S_F64_QUARTER) ? 16u : 64u);
    t0 = t1 = 0;
    if (lane < nact) {
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
      const int cap = (SA == S_NONE) ? itersA : (1 << 24); /* bounded whatever happens to the flag */
      for (int it = 0; it < cap; ++it) {
        body<SB>(r);
        done += 1;
        if (SA != S_NONE && flag[wave - 4] != 0u) break;
      }
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    }
  }
  if (lane == 0) {
    const unsigned wv = blockIdx.x * 8 + wave;
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
      {"A fp64 mul alone                       ", k_pair<S_F64, 0, S_NONE, 0>},
      {"A exact chain alone                    ", k_pair<S_CHAIN, 0, S_NONE, 0>},
      {"B Philox-like alone                    ", k_pair<S_NONE, 0, S_GEN, 0>},
      {"B xor alone                            ", k_pair<S_NONE, 0, S_XOR, 0>},
      {"B 15 xor : 1 ds_write_b16 alone        ", k_pair<S_NONE, 0, S_LDSW16, 0>},
      {"B 32 xor : 1 ds_write_b128 alone       ", k_pair<S_NONE, 0, S_LDSW128, 0>},
      {"A fp64 half exec alone                 ", k_pair<S_F64_HALF, 0, S_NONE, 0>},
      {"A fp64 quarter exec alone              ", k_pair<S_F64_QUARTER, 0, S_NONE, 0>},
      {"A fp64 prio0 | B fp64 prio0            ", k_pair<S_F64, 0, S_F64, 0>},
      {"A fp64 prio3 | B fp64 prio0            ", k_pair<S_F64, 3, S_F64, 0>},
      {"A fp64 prio0 | B fp64 prio3            ", k_pair<S_F64, 0, S_F64, 3>},
      {"A chain prio0 | B Philox prio0         ", k_pair<S_CHAIN, 0, S_GEN, 0>},
      {"A chain prio3 | B Philox prio0         ", k_pair<S_CHAIN, 3, S_GEN, 0>},
      {"A chain prio0 | B Philox prio3         ", k_pair<S_CHAIN, 0, S_GEN, 3>},
      {"A chain toggling 3/0 | B Philox prio1  ", k_pair<S_CHAIN_TOGGLE, 0, S_GEN, 1>},
      {"A chain prio3 | B xor prio0            ", k_pair<S_CHAIN, 3, S_XOR, 0>},
      {"A chain prio0 | B xor prio0            ", k_pair<S_CHAIN, 0, S_XOR, 0>},
      {"A chain prio3 | B xor+ds_write_b16     ", k_pair<S_CHAIN, 3, S_LDSW16, 0>},
      {"A chain prio3 | B xor+ds_write_b128    ", k_pair<S_CHAIN, 3, S_LDSW128, 0>},
      {"A chain prio0 | B xor+ds_write_b16     ", k_pair<S_CHAIN, 0, S_LDSW16, 0>},
      {"A fp64 half exec | B fp64 half exec    ", k_pair<S_F64_HALF, 0, S_F64_HALF, 0>},
      {"A fma prio3 | B Philox prio0           ", k_pair<S_FMA, 3, S_GEN, 0>},
  };
  const int iters = 2000;
  unsigned long long *d_out;
  double *d_sink;
  (void)hipMalloc(&d_out, 2 * 2048 * sizeof(unsigned long long));
  (void)hipMalloc(&d_sink, 256 * 512 * sizeof(double));
  std::vector<unsigned long long> h(2 * 2048);
  printf("one workgroup of 8 wavefronts per CU: wavefronts 0-3 = A, 4-7 = B, wavefront w and w+4 share a SIMD\n");
  printf("ticks per instruction of each wavefront while both run (64 instructions per iteration; B runs until A is done)\n");
  printf("%-40s %10s %10s %12s\n", "case", "A", "B", "SIMD total");
  for (auto &c : cases) {
    (void)hipMemset(d_out, 0, 2 * 2048 * sizeof(unsigned long long));
    hipLaunchKernelGGL(c.fn, dim3(256), dim3(512), 0, 0, d_out, d_sink, iters);
    (void)hipMemset(d_out, 0, 2 * 2048 * sizeof(unsigned long long));
    hipLaunchKernelGGL(c.fn, dim3(256), dim3(512), 0, 0, d_out, d_sink, iters);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h.data(), d_out, 2 * 2048 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double ta = 0, ia = 0, tb = 0, ib = 0;
    for (int wg = 0; wg < 256; ++wg)
      for (int w = 0; w < 8; ++w) {
        const double t = (double)h[2 * (wg * 8 + w)], n = (double)h[2 * (wg * 8 + w) + 1] * 64.0;
        if (w < 4) { ta += t; ia += n; } else { tb += t; ib += n; }
      }
    const double ca = ia > 0 ? ta / ia : 0, cb = ib > 0 ? tb / ib : 0;
    const double rate = (ca > 0 ? 1.0 / ca : 0) + (cb > 0 ? 1.0 / cb : 0);
    printf("%-40s %10.2f %10.2f %12.2f\n", c.name, ca, cb, rate > 0 ? 1.0 / rate : 0.0);
    fflush(stdout);
  }
  (void)hipFree(d_out);
  (void)hipFree(d_sink);
  return 0;
}
