// Micro-benchmark: cycles per VALU instruction for one wave per SIMD on gfx950 (s_memtime).
// Build: hipcc -O3 --offload-arch=gfx950 -o ubench ubench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

#define BENCH(NAME, BODY)                                                                  \
  __global__ void NAME(unsigned long long *out, double *sink, int iters)                   \
  {                                                                                        \
    if ((int)threadIdx.x >= iters >> 16) return; /* iters>>16 = active lanes */                \
    iters &= 0xFFFF;                                                                       \
    double a0 = threadIdx.x * 1.0001 + 1.0, a1 = a0 + 1.5, a2 = a0 + 2.5, a3 = a0 + 3.5;   \
    double b0 = 0.999999, b1 = 1.000001;                                                   \
    unsigned u0 = threadIdx.x * 2654435761u + 7u, u1 = u0 ^ 0x9E3779B9u, u2 = u1 * 3u, u3 = u2 + 11u; \
    unsigned long long q0 = u0, q1 = u1;                                                   \
    float f0 = threadIdx.x + 0.5f, f1 = f0 + 1.0f;                                         \
    unsigned long long t0, t1;                                                             \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");             \
    for (int it = 0; it < iters; ++it) {                                                   \
      BODY                                                                                 \
    }                                                                                      \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");             \
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                       \
    sink[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + b0 + b1 + u0 + u1 + u2 + u3 + (double)q0 + (double)q1 + f0 + f1; \
  }

// 64 instructions per iteration
BENCH(k_mul_f64_indep, REP16(asm volatile("v_mul_f64 %0, %0, %4\n\tv_mul_f64 %1, %1, %4\n\tv_mul_f64 %2, %2, %4\n\tv_mul_f64 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));))
BENCH(k_mul_f64_dep, REP64(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a0) : "v"(b0));))
BENCH(k_add_f64_dep, REP64(asm volatile("v_add_f64 %0, %0, %1" : "+v"(a0) : "v"(b0));))
BENCH(k_fma_f64_dep, REP64(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a0) : "v"(b0), "v"(b1));))
BENCH(k_fma_f64_indep, REP16(asm volatile("v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %4, %5\n\tv_fma_f64 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));))
BENCH(k_muladd_chain, REP16(asm volatile("v_mul_f64 %1, %2, %3\n\tv_add_f64 %0, %0, -%1\n\tv_mul_f64 %1, %2, %4\n\tv_add_f64 %0, %0, -%1" : "+v"(a0), "+v"(a1) : "v"(a2), "v"(b0), "v"(b1));))
BENCH(k_ceil_f64, REP64(asm volatile("v_ceil_f64 %0, %0" : "+v"(a0));))
BENCH(k_cvt_i32_f64, REP64(asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(u0) : "v"(a0));))
BENCH(k_cvt_f64_u32, REP64(asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(a0) : "v"(u0));))
BENCH(k_xor_dep, REP64(asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u0) : "v"(u1));))
BENCH(k_xor_indep, REP16(asm volatile("v_xor_b32 %0, %0, %4\n\tv_xor_b32 %1, %1, %4\n\tv_xor_b32 %2, %2, %4\n\tv_xor_b32 %3, %3, %0" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(f0));))
BENCH(k_mad_u64_u32_dep, REP64(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(q0) : "v"(u0), "v"(u1) : "vcc"); u0 = (unsigned)(q0 >> 32);))
BENCH(k_mad_u64_u32_indep, REP16(asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, 0\n\tv_mad_u64_u32 %1, vcc, %4, %3, 0\n\tv_mad_u64_u32 %0, vcc, %5, %3, 0\n\tv_mad_u64_u32 %1, vcc, %2, %4, 0" : "=&v"(q0), "=&v"(q1) : "v"(u0), "v"(u1), "v"(u2), "v"(u3) : "vcc");))
BENCH(k_mul_hi_u32, REP64(asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u0) : "v"(u1));))
BENCH(k_mul_lo_u32, REP64(asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u0) : "v"(u1));))
BENCH(k_mul_f32_dep, REP64(asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f0) : "v"(f1));))
BENCH(k_cndmask, REP64(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u0) : "v"(u1) : "vcc");))
BENCH(k_cmp_cnd, REP16(asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc\n\tv_cmp_lt_u32 vcc, %0, %2\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(u0) : "v"(u1), "v"(u2) : "vcc");))
BENCH(k_cmp_saveexec, REP16(asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\ts_and_saveexec_b64 s[20:21], vcc\n\tv_add_u32 %0, %0, %2\n\ts_or_b64 exec, exec, s[20:21]" : "+v"(u0) : "v"(u1), "v"(u2) : "vcc", "s20", "s21");))
BENCH(k_floor_f64, REP64(asm volatile("v_floor_f64 %0, %0" : "+v"(a0));))
BENCH(k_cmp_f64, REP64(asm volatile("v_cmp_gt_f64 vcc, %0, %1" : : "v"(a0), "v"(a1) : "vcc");))

typedef void (*kern_t)(unsigned long long *, double *, int);
struct Case { const char *name; kern_t fn; int instr_per_iter; };

int main()
{
  Case cases[] = {
      {"v_mul_f64 indep x4", k_mul_f64_indep, 64}, {"v_mul_f64 dependent", k_mul_f64_dep, 64},
      {"v_add_f64 dependent", k_add_f64_dep, 64}, {"v_fma_f64 dependent", k_fma_f64_dep, 64},
      {"v_fma_f64 indep x4", k_fma_f64_indep, 64}, {"mul+add chain (exact filter shape)", k_muladd_chain, 64},
      {"v_ceil_f64", k_ceil_f64, 64}, {"v_floor_f64", k_floor_f64, 64}, {"v_cvt_i32_f64", k_cvt_i32_f64, 64},
      {"v_cvt_f64_u32", k_cvt_f64_u32, 64}, {"v_cmp_gt_f64", k_cmp_f64, 64}, {"v_xor_b32 dependent", k_xor_dep, 64},
      {"v_xor_b32 indep", k_xor_indep, 64}, {"v_mad_u64_u32 dependent", k_mad_u64_u32_dep, 64},
      {"v_mad_u64_u32 indep", k_mad_u64_u32_indep, 64}, {"v_mul_hi_u32 dep", k_mul_hi_u32, 64},
      {"v_mul_lo_u32 dep", k_mul_lo_u32, 64}, {"v_mul_f32 dep", k_mul_f32_dep, 64},
      {"v_cndmask dep", k_cndmask, 64}, {"v_cmp+v_cndmask", k_cmp_cnd, 64},
      {"cmp+saveexec+add+or (4 instr)", k_cmp_saveexec, 64},
  };
  const int iters = 2000;
  unsigned long long *d_out;
  double *d_sink;
  for (int cfg = 0; cfg < 4; ++cfg) {
    const int waves_per_simd = (cfg == 0) ? 1 : 2;
    const int active = (cfg == 2) ? 32 : (cfg == 3 ? 16 : 64);
    const int grid = 1024 * waves_per_simd;  // 64-thread blocks: 1 or 2 waves per SIMD on 256 CUs
    hipMalloc(&d_out, grid * sizeof(unsigned long long));
    hipMalloc(&d_sink, grid * 64 * sizeof(double));
    std::vector<unsigned long long> h(grid);
    printf("---- %d wave(s) per SIMD (grid %d x 64 threads), %d active lanes per wave ----\n", waves_per_simd, grid, active);
    for (auto &c : cases) {
      hipLaunchKernelGGL(c.fn, dim3(grid), dim3(64), 0, 0, d_out, d_sink, iters | (active << 16));
      hipLaunchKernelGGL(c.fn, dim3(grid), dim3(64), 0, 0, d_out, d_sink, iters | (active << 16));
      hipDeviceSynchronize();
      hipMemcpy(h.data(), d_out, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost);
      double s = 0;
      for (auto v : h) s += (double)v;
      s /= grid;
      printf("%-38s %6.2f cycles/instr (per wave)\n", c.name, s / ((double)iters * c.instr_per_iter));
    }
    hipFree(d_out);
    hipFree(d_sink);
  }
  return 0;
}
