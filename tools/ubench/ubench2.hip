// Micro-benchmark 2: VALU issue cost of instruction MIXES at 1..4 waves per SIMD on gfx950.
// Questions behind it (round 2): what does a second / third / fourth wave per SIMD buy for the
// filter's fp64 stream and for the generator's integer stream; is a 32-bit integer instruction
// free in the shadow of an fp64 one; what do LDS byte accesses and exec-mask branches cost.
// Build: hipcc -O3 --offload-arch=gfx950 -o ubench2 ubench2.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP32(x) REP16(x) REP16(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

#define BENCH(NAME, BODY)                                                                  \
  __global__ void NAME(unsigned long long *out, double *sink, int iters)                   \
  {                                                                                        \
    __shared__ short lds[64 * 64];                                                         \
    double a0 = threadIdx.x * 1.0001 + 1.0, a1 = a0 + 1.5, a2 = a0 + 2.5, a3 = a0 + 3.5;   \
    double c0 = a0 * 0.5, c1 = a1 * 0.5, c2 = a2 * 0.5, c3 = a3 * 0.5;                     \
    double b0 = 0.999999, b1 = 1.000001;                                                   \
    unsigned u0 = threadIdx.x * 2654435761u + 7u, u1 = u0 ^ 0x9E3779B9u, u2 = u1 * 3u, u3 = u2 + 11u; \
    unsigned w0 = u0 + 1u, w1 = u1 + 2u, w2 = u2 + 3u, w3 = u3 + 4u;                       \
    unsigned long long q0 = u0, q1 = u1;                                                   \
    float g0 = threadIdx.x * 0.5f + 1.0f, g1 = g0 + 1.5f, g2 = g0 + 2.5f, g3 = g0 + 3.5f;  \
    float h0 = 0.999999f, h1 = 1.000001f;                                                  \
    unsigned la = (threadIdx.x & 63u) * 2u;                                                \
    lds[threadIdx.x] = (short)u0;                                                          \
    __syncthreads();                                                                       \
    unsigned long long t0, t1;                                                             \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");             \
    for (int it = 0; it < iters; ++it) {                                                   \
      BODY                                                                                 \
    }                                                                                      \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory"); \
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0; \
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + c0 + c1 + c2 + c3 + b0 + b1 + u0 + u1 + u2 + u3 + w0 + w1 + w2 + w3 + (double)q0 + (double)q1 + lds[threadIdx.x] + la + g0 + g1 + g2 + g3 + h0 + h1; \
  }

// ---- 64 instructions per iteration unless noted ----
BENCH(k_f64_indep, REP16(asm volatile("v_mul_f64 %0, %0, %4\n\tv_mul_f64 %1, %1, %4\n\tv_mul_f64 %2, %2, %4\n\tv_mul_f64 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));))
BENCH(k_fma64_indep, REP16(asm volatile("v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %4, %5\n\tv_fma_f64 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));))
// exact filter shape: product, then the chain's subtraction
BENCH(k_muladd_chain, REP16(asm volatile("v_mul_f64 %1, %2, %3\n\tv_add_f64 %0, %0, -%1\n\tv_mul_f64 %1, %2, %4\n\tv_add_f64 %0, %0, -%1" : "+v"(a0), "+v"(a1) : "v"(a2), "v"(b0), "v"(b1));))
// two interleaved exact chains (two independent utterances per lane)
BENCH(k_muladd_chain2, REP16(asm volatile("v_mul_f64 %1, %4, %6\n\tv_mul_f64 %3, %5, %6\n\tv_add_f64 %0, %0, -%1\n\tv_add_f64 %2, %2, -%3" : "+v"(a0), "+v"(a1), "+v"(c0), "+v"(c1) : "v"(a2), "v"(c2), "v"(b0));))
// fma chain (FMA-mode filter shape, one partial sum)
BENCH(k_fma_chain, REP64(asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a0) : "v"(b0), "v"(b1));))
BENCH(k_fma_chain4, REP16(asm volatile("v_fma_f64 %0, %4, %5, %0\n\tv_fma_f64 %1, %4, %5, %1\n\tv_fma_f64 %2, %4, %5, %2\n\tv_fma_f64 %3, %4, %5, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));))
// single precision: what an fp32 filter mode would issue
BENCH(k_fma32_indep, REP16(asm volatile("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5" : "+v"(g0), "+v"(g1), "+v"(g2), "+v"(g3) : "v"(h0), "v"(h1));))
BENCH(k_fma32_chain2, REP32(asm volatile("v_fma_f32 %0, %2, %3, %0\n\tv_fma_f32 %1, %2, %3, %1" : "+v"(g0), "+v"(g1) : "v"(h0), "v"(h1));))
BENCH(k_pkfma32_indep, REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n\tv_pk_fma_f32 %1, %1, %4, %5\n\tv_pk_fma_f32 %2, %2, %4, %5\n\tv_pk_fma_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));))
BENCH(k_pkfma32_chain2, REP32(asm volatile("v_pk_fma_f32 %0, %2, %3, %0\n\tv_pk_fma_f32 %1, %2, %3, %1" : "+v"(a0), "+v"(a1) : "v"(b0), "v"(b1));))
// 32-bit integer streams
BENCH(k_xor_indep, REP16(asm volatile("v_xor_b32 %0, %0, %4\n\tv_xor_b32 %1, %1, %4\n\tv_xor_b32 %2, %2, %4\n\tv_xor_b32 %3, %3, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(w0));))
BENCH(k_mad64_indep, REP16(asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, 0\n\tv_mad_u64_u32 %1, vcc, %4, %3, 0\n\tv_mad_u64_u32 %0, vcc, %5, %3, 0\n\tv_mad_u64_u32 %1, vcc, %2, %4, 0" : "=&v"(q0), "=&v"(q1) : "v"(u0), "v"(u1), "v"(u2), "v"(u3) : "vcc");))
BENCH(k_mulhi_indep, REP16(asm volatile("v_mul_hi_u32 %0, %4, %5\n\tv_mul_hi_u32 %1, %4, %6\n\tv_mul_hi_u32 %2, %5, %6\n\tv_mul_hi_u32 %3, %4, %4" : "=&v"(u0), "=&v"(u1), "=&v"(u2), "=&v"(u3) : "v"(w0), "v"(w1), "v"(w2));))
BENCH(k_mullo_indep, REP16(asm volatile("v_mul_lo_u32 %0, %4, %5\n\tv_mul_lo_u32 %1, %4, %6\n\tv_mul_lo_u32 %2, %5, %6\n\tv_mul_lo_u32 %3, %4, %4" : "=&v"(u0), "=&v"(u1), "=&v"(u2), "=&v"(u3) : "v"(w0), "v"(w1), "v"(w2));))
// mixes: fp64 next to 32-bit integer work of the same wave
BENCH(k_f64_xor_1to1, REP16(asm volatile("v_mul_f64 %0, %0, %4\n\tv_xor_b32 %2, %2, %5\n\tv_mul_f64 %1, %1, %4\n\tv_xor_b32 %3, %3, %5" : "+v"(a0), "+v"(a1), "+v"(u0), "+v"(u1) : "v"(b0), "v"(w0));))
BENCH(k_f64_xor_1to2, REP16(asm volatile("v_mul_f64 %0, %0, %4\n\tv_xor_b32 %1, %1, %5\n\tv_xor_b32 %2, %2, %5\n\tv_xor_b32 %3, %3, %5" : "+v"(a0), "+v"(u0), "+v"(u1), "+v"(u2) : "v"(b0), "v"(w0));))
BENCH(k_f64_mad64_1to1, REP16(asm volatile("v_mul_f64 %0, %0, %4\n\tv_mad_u64_u32 %2, vcc, %5, %6, 0\n\tv_mul_f64 %1, %1, %4\n\tv_mad_u64_u32 %3, vcc, %6, %7, 0" : "+v"(a0), "+v"(a1), "=&v"(q0), "=&v"(q1) : "v"(b0), "v"(u0), "v"(u1), "v"(u2) : "vcc");))
// chain with one independent filler between dependent instructions
BENCH(k_chain_filler, REP32(asm volatile("v_add_f64 %0, %0, %2\n\tv_mul_f64 %1, %1, %2" : "+v"(a0), "+v"(a1) : "v"(b0));))
BENCH(k_chain_filler_xor, REP32(asm volatile("v_add_f64 %0, %0, %2\n\tv_xor_b32 %1, %1, %3" : "+v"(a0), "+v"(u0) : "v"(b0), "v"(w0));))
BENCH(k_chain_2filler_xor, REP16(asm volatile("v_add_f64 %0, %0, %3\n\tv_xor_b32 %1, %1, %4\n\tv_xor_b32 %2, %2, %4\n\tv_add_f64 %0, %0, %3" : "+v"(a0), "+v"(u0), "+v"(u1) : "v"(b0), "v"(w0));))
// conversions, rounding, compare/select as the kernels use them
BENCH(k_cvt_f64_u32_indep, REP16(asm volatile("v_cvt_f64_u32 %0, %4\n\tv_cvt_f64_u32 %1, %5\n\tv_cvt_f64_u32 %2, %6\n\tv_cvt_f64_u32 %3, %7" : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(u0), "v"(u1), "v"(u2), "v"(u3));))
BENCH(k_cvt_i32_f64_indep, REP16(asm volatile("v_cvt_i32_f64 %0, %4\n\tv_cvt_i32_f64 %1, %5\n\tv_cvt_i32_f64 %2, %6\n\tv_cvt_i32_f64 %3, %7" : "=v"(u0), "=v"(u1), "=v"(u2), "=v"(u3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));))
BENCH(k_floor_indep, REP16(asm volatile("v_floor_f64 %0, %4\n\tv_floor_f64 %1, %5\n\tv_floor_f64 %2, %6\n\tv_floor_f64 %3, %7" : "=v"(c0), "=v"(c1), "=v"(c2), "=v"(c3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));))
BENCH(k_cmp_cnd_vcc, REP16(asm volatile("v_cmp_lt_u32 vcc, %0, %2\n\tv_cndmask_b32 %0, %0, %3, vcc\n\tv_cmp_lt_u32 vcc, %1, %3\n\tv_cndmask_b32 %1, %1, %2, vcc" : "+v"(u0), "+v"(u1) : "v"(w0), "v"(w1) : "vcc");))
BENCH(k_cmp_cnd_sgpr, REP16(asm volatile("v_cmp_lt_u32 s[20:21], %0, %2\n\tv_cmp_lt_u32 s[22:23], %1, %3\n\tv_cndmask_b32 %0, %0, %3, s[20:21]\n\tv_cndmask_b32 %1, %1, %2, s[22:23]" : "+v"(u0), "+v"(u1) : "v"(w0), "v"(w1) : "s20", "s21", "s22", "s23");))
BENCH(k_med3_indep, REP16(asm volatile("v_med3_i32 %0, %0, %4, %5\n\tv_med3_i32 %1, %1, %4, %5\n\tv_med3_i32 %2, %2, %4, %5\n\tv_med3_i32 %3, %3, %4, %5" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(w0), "v"(w1));))
// LDS: one int16 per lane, conflict-free ([slot][lane] layout of the ring)
BENCH(k_ds_write_b16, REP16(asm volatile("ds_write_b16 %0, %1\n\tds_write_b16 %0, %2 offset:128\n\tds_write_b16 %0, %3 offset:256\n\tds_write_b16 %0, %4 offset:384" : : "v"(la), "v"(u0), "v"(u1), "v"(u2), "v"(u3) : "memory");))
BENCH(k_ds_read_i16, REP16(asm volatile("ds_read_i16 %0, %4\n\tds_read_i16 %1, %4 offset:128\n\tds_read_i16 %2, %4 offset:256\n\tds_read_i16 %3, %4 offset:384\n\ts_waitcnt lgkmcnt(0)" : "=v"(u0), "=v"(u1), "=v"(u2), "=v"(u3) : "v"(la) : "memory");))
BENCH(k_ds_write_f64mix, REP16(asm volatile("ds_write_b16 %2, %3\n\tv_mul_f64 %0, %0, %4\n\tds_write_b16 %2, %3 offset:128\n\tv_mul_f64 %1, %1, %4" : "+v"(a0), "+v"(a1) : "v"(la), "v"(u0), "v"(b0) : "memory");))
// scalar / branch overhead in a lone wave: 2 valu + 2 salu
BENCH(k_salu_mix, REP16(asm volatile("v_mul_f64 %0, %0, %2\n\ts_add_u32 s20, s20, 1\n\tv_mul_f64 %1, %1, %2\n\ts_and_b32 s21, s20, 7" : "+v"(a0), "+v"(a1) : "v"(b0) : "s20", "s21");))
BENCH(k_saveexec_mix, REP16(asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\ts_and_saveexec_b64 s[20:21], vcc\n\tv_add_u32 %0, %0, %2\n\ts_or_b64 exec, exec, s[20:21]" : "+v"(u0) : "v"(w0), "v"(w1) : "vcc", "s20", "s21");))

// in-kernel clock (MI355X_MICROARCH.md, DVFS item 6): delta s_memtime / delta s_memrealtime x 100 MHz
__global__ void k_clock(unsigned long long *out, double *sink, int iters)
{
  double a0 = threadIdx.x * 1.0001 + 1.0, a1 = a0 + 1.5, a2 = a0 + 2.5, a3 = a0 + 3.5, b0 = 0.999999;
  unsigned long long t0, t1, r0, r1;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
  for (int it = 0; it < iters; ++it) {
    REP16(asm volatile("v_mul_f64 %0, %0, %4\n\tv_mul_f64 %1, %1, %4\n\tv_mul_f64 %2, %2, %4\n\tv_mul_f64 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));)
  }
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
  if ((threadIdx.x & 63) == 0) {
    const unsigned wv = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    out[2 * wv] = t1 - t0;
    out[2 * wv + 1] = r1 - r0;
  }
  sink[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}

typedef void (*kern_t)(unsigned long long *, double *, int);
struct Case { const char *name; kern_t fn; int instr_per_iter; };

int main()
{
  Case cases[] = {
      {"v_mul_f64 indep x4", k_f64_indep, 64},
      {"v_fma_f64 indep x4", k_fma64_indep, 64},
      {"mul+add chain (exact filter shape)", k_muladd_chain, 64},
      {"two interleaved mul+add chains", k_muladd_chain2, 64},
      {"fma chain (1 accumulator)", k_fma_chain, 64},
      {"fma chains (4 accumulators)", k_fma_chain4, 64},
      {"v_fma_f32 indep x4", k_fma32_indep, 64},
      {"v_fma_f32 two chains", k_fma32_chain2, 64},
      {"v_pk_fma_f32 indep x4 (2 fp32 fma each)", k_pkfma32_indep, 64},
      {"v_pk_fma_f32 two chains", k_pkfma32_chain2, 64},
      {"v_xor_b32 indep x4", k_xor_indep, 64},
      {"v_mad_u64_u32 indep", k_mad64_indep, 64},
      {"v_mul_hi_u32 indep", k_mulhi_indep, 64},
      {"v_mul_lo_u32 indep", k_mullo_indep, 64},
      {"mix f64 : xor = 1:1", k_f64_xor_1to1, 64},
      {"mix f64 : xor = 1:3", k_f64_xor_1to2, 64},
      {"mix f64 : mad_u64 = 1:1", k_f64_mad64_1to1, 64},
      {"f64 chain + f64 filler", k_chain_filler, 64},
      {"f64 chain + xor filler", k_chain_filler_xor, 64},
      {"f64 chain x2 + 2 xor fillers (4 instr)", k_chain_2filler_xor, 64},
      {"v_cvt_f64_u32 indep", k_cvt_f64_u32_indep, 64},
      {"v_cvt_i32_f64 indep", k_cvt_i32_f64_indep, 64},
      {"v_floor_f64 indep", k_floor_indep, 64},
      {"v_cmp + v_cndmask via vcc", k_cmp_cnd_vcc, 64},
      {"v_cmp x2 + v_cndmask x2 via sgpr pairs", k_cmp_cnd_sgpr, 64},
      {"v_med3_i32 indep", k_med3_indep, 64},
      {"ds_write_b16", k_ds_write_b16, 64},
      {"ds_read_i16 x4 + wait (5 instr)", k_ds_read_i16, 80},
      {"ds_write_b16 : f64 = 1:1", k_ds_write_f64mix, 64},
      {"f64 : salu = 1:1", k_salu_mix, 64},
      {"cmp+saveexec+add+or", k_saveexec_mix, 64},
  };
  const int iters = 1000;
  unsigned long long *d_out;
  double *d_sink;
  const int wps[] = {1, 2, 3, 4};
  hipMalloc(&d_out, 4096 * sizeof(unsigned long long));
  hipMalloc(&d_sink, 4096 * 64 * sizeof(double));
  std::vector<unsigned long long> h(4096);
  printf("cycles per instruction PER WAVE (divide by waves/SIMD for the SIMD's combined issue cost)\n");
  printf("%-42s %8s %8s %8s %8s\n", "waves per SIMD", "1", "2", "3", "4");
  for (auto &c : cases) {
    printf("%-42s", c.name);
    for (int w : wps) {
      // ONE workgroup of 4*w wavefronts per CU: a workgroup's wavefronts are dealt to the CU's four
      // SIMDs cyclically, so every SIMD gets exactly w of them (single-wave workgroups land
      // unevenly -- 3+3+1+1 instead of 2+2+2+2 -- which is what the first version of this
      // benchmark measured without knowing it)
      const int grid = 1024 * w;  // wavefronts
      hipLaunchKernelGGL(c.fn, dim3(256), dim3(256 * w), 0, 0, d_out, d_sink, iters);
      hipLaunchKernelGGL(c.fn, dim3(256), dim3(256 * w), 0, 0, d_out, d_sink, iters);
      hipDeviceSynchronize();
      hipMemcpy(h.data(), d_out, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost);
      double s = 0;
      for (int i = 0; i < grid; ++i) s += (double)h[i];
      s /= grid;
      printf(" %8.2f", s / ((double)iters * c.instr_per_iter));
    }
    printf("\n");
    fflush(stdout);
  }
  // wall-clock check of the tick: the same fp64 kernel timed with HIP events
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int w : wps) {
    const int grid = 1024 * w;
    hipLaunchKernelGGL(k_f64_indep, dim3(256), dim3(256 * w), 0, 0, d_out, d_sink, 20000);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_f64_indep, dim3(256), dim3(256 * w), 0, 0, d_out, d_sink, 20000);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h.data(), d_out, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < grid; ++i) s += (double)h[i];
    s /= grid;
    printf("wall check, %d wave(s)/SIMD: %.3f ms for %.0f ticks per wave -> %.3f GHz tick rate; %.2f TFLOP/s fp64 (mul only)\n",
           w, ms, s, s / (ms * 1e6), (double)grid * 64 * 20000 * 64 / (ms * 1e-3) / 1e12);
  }
  // the same question asked inside the kernel: shader cycles per 100 MHz reference tick
  {
    unsigned long long *d2;
    hipMalloc(&d2, 2 * 4096 * sizeof(unsigned long long));
    std::vector<unsigned long long> h2(2 * 4096);
    for (int w : wps) {
      const int grid = 1024 * w;
      for (int rep = 0; rep < 40; ++rep)  // ~0.3-1 s of back-to-back launches: let the clock settle
        hipLaunchKernelGGL(k_clock, dim3(256), dim3(256 * w), 0, 0, d2, d_sink, 20000);
      hipDeviceSynchronize();
      hipMemcpy(h2.data(), d2, 2 * grid * sizeof(unsigned long long), hipMemcpyDeviceToHost);
      std::vector<double> ghz(grid);
      for (int i = 0; i < grid; ++i) ghz[i] = (double)h2[2 * i] / (double)h2[2 * i + 1] * 0.1;
      std::sort(ghz.begin(), ghz.end());
      printf("in-kernel clock, %d wave(s)/SIMD of v_mul_f64: median %.3f GHz (min %.3f, max %.3f) = delta s_memtime / delta s_memrealtime x 100 MHz\n",
             w, ghz[grid / 2], ghz[0], ghz[grid - 1]);
    }
    hipFree(d2);
  }
  hipFree(d_out);
  hipFree(d_sink);
  return 0;
}
