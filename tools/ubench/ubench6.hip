// Micro-benchmark 6 (round 4): the fp64 matrix pipe of gfx950 -- what a block-form filter could be built on.
//   part 1: issue cost of v_mfma_f64_16x16x4_f64 / v_mfma_f64_4x4x4_4b_f64, one wavefront per SIMD: independent
//           accumulators (1, 2, 4), a dependent accumulation chain, a result fed back as the B operand, and k
//           independent vector instructions (fp64 / 32-bit) in each gap between two MFMAs;
//   part 2: an MFMA wavefront next to one or two vector-only wavefronts on the SAME SIMD (the three-role shape);
//   part 3: operand layouts, checked numerically: D = A x B against the host, and "D of one product is the B of the
//           next, register for register" (row = (lane >> 4) + 4 * reg, k = lane >> 4).
// Build: hipcc -O3 --offload-arch=gfx950 -o ubench6 ubench6.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef double v4d __attribute__((ext_vector_type(4)));

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

enum {
  M16_1 = 0,   // 16x16x4, one accumulator (dependent chain through srcC)
  M16_2,       // two accumulators alternating
  M16_4,       // four accumulators
  M16_FEED,    // four accumulators; B operand of each = a register pair of ANOTHER accumulator written 3 MFMAs earlier
  M16_FEED1,   // two accumulators; B of each = pair of the accumulator written by the PREVIOUS MFMA
  M4_1,        // 4x4x4 (4 blocks), one accumulator
  M4_4,        // four accumulators
  M16_F2, M16_F4, M16_F8, M16_F12, M16_F16,  // 4 accumulators, k v_fma_f64 (independent) per MFMA
  M16_X4, M16_X8, M16_X16,                    // k v_xor_b32 per MFMA
  M16_L2,                                     // 2 ds_read_u16 + wait per MFMA
  M4_F2, M4_F4,
  F_ONLY,                                     // v_fma_f64 only (reference)
  P1_COUNT
};
static const char *p1_name[P1_COUNT] = {
  "mfma16 x1 acc (dependent)", "mfma16 x2 acc", "mfma16 x4 acc", "mfma16 x4, B = other acc (3 back)", "mfma16 x2, B = previous D",
  "mfma4x4 x1 acc (dependent)", "mfma4x4 x4 acc",
  "mfma16 + 2 v_fma_f64", "mfma16 + 4 v_fma_f64", "mfma16 + 8 v_fma_f64", "mfma16 + 12 v_fma_f64", "mfma16 + 16 v_fma_f64",
  "mfma16 + 4 v_xor_b32", "mfma16 + 8 v_xor_b32", "mfma16 + 16 v_xor_b32", "mfma16 + 2 ds_read_u16", "mfma4x4 + 2 v_fma_f64", "mfma4x4 + 4 v_fma_f64",
  "v_fma_f64 only (per 1)"};
// instructions per "unit" (one MFMA gap) -- for reporting
static const int p1_mfma_per_body[P1_COUNT] = {64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64};

struct R6 {
  v4d c0, c1, c2, c3;
  double a, b;
  double s0, s1, s2, s3; // 4x4 accumulators
  double f0, f1, f2, f3, g0, g1;
  unsigned u0, u1, u2, u3, w0;
  unsigned la;
};

template <int O>
__device__ __forceinline__ void body6(R6 &r)
{
  if (O == M16_1) {
    REP16(asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0\n\tv_mfma_f64_16x16x4_f64 %0, %1, %2, %0\n\tv_mfma_f64_16x16x4_f64 %0, %1, %2, %0\n\tv_mfma_f64_16x16x4_f64 %0, %1, %2, %0"
                       : "+v"(r.c0) : "v"(r.a), "v"(r.b));)
  } else if (O == M16_2) {
    REP16(asm volatile("v_mfma_f64_16x16x4_f64 %0, %2, %3, %0\n\tv_mfma_f64_16x16x4_f64 %1, %2, %3, %1\n\tv_mfma_f64_16x16x4_f64 %0, %2, %3, %0\n\tv_mfma_f64_16x16x4_f64 %1, %2, %3, %1"
                       : "+v"(r.c0), "+v"(r.c1) : "v"(r.a), "v"(r.b));)
  } else if (O == M16_4) {
    REP16(asm volatile("v_mfma_f64_16x16x4_f64 %0, %4, %5, %0\n\tv_mfma_f64_16x16x4_f64 %1, %4, %5, %1\n\tv_mfma_f64_16x16x4_f64 %2, %4, %5, %2\n\tv_mfma_f64_16x16x4_f64 %3, %4, %5, %3"
                       : "+v"(r.c0), "+v"(r.c1), "+v"(r.c2), "+v"(r.c3) : "v"(r.a), "v"(r.b));)
  } else if (O == M16_FEED) {
    // B of accumulator i = first register pair of accumulator (i+1)%4, written three MFMAs earlier
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      r.c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(r.a, r.c1[0], r.c0, 0, 0, 0);
      r.c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(r.a, r.c2[0], r.c1, 0, 0, 0);
      r.c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(r.a, r.c3[0], r.c2, 0, 0, 0);
      r.c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(r.a, r.c0[0], r.c3, 0, 0, 0);
    }
  } else if (O == M16_FEED1) {
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      r.c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(r.a, r.c1[1], r.c0, 0, 0, 0);
      r.c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(r.a, r.c0[1], r.c1, 0, 0, 0);
    }
  } else if (O == M4_1) {
    REP16(asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0\n\tv_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0\n\tv_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0\n\tv_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0"
                       : "+v"(r.s0) : "v"(r.a), "v"(r.b));)
  } else if (O == M4_4) {
    REP16(asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %4, %5, %0\n\tv_mfma_f64_4x4x4_4b_f64 %1, %4, %5, %1\n\tv_mfma_f64_4x4x4_4b_f64 %2, %4, %5, %2\n\tv_mfma_f64_4x4x4_4b_f64 %3, %4, %5, %3"
                       : "+v"(r.s0), "+v"(r.s1), "+v"(r.s2), "+v"(r.s3) : "v"(r.a), "v"(r.b));)
  } else if (O >= M16_F2 && O <= M16_F16) {
    constexpr int K = (O == M16_F2) ? 2 : (O == M16_F4) ? 4 : (O == M16_F8) ? 8 : (O == M16_F12) ? 12 : 16;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      asm volatile("v_mfma_f64_16x16x4_f64 %0, %4, %5, %0\n\t" : "+v"(r.c0), "+v"(r.c1), "+v"(r.c2), "+v"(r.c3) : "v"(r.a), "v"(r.b));
#pragma unroll
      for (int k = 0; k < K; k += 2) asm volatile("v_fma_f64 %0, %0, %2, %3\n\tv_fma_f64 %1, %1, %2, %3" : "+v"(r.f0), "+v"(r.f1) : "v"(r.g0), "v"(r.g1));
      asm volatile("v_mfma_f64_16x16x4_f64 %1, %4, %5, %1\n\t" : "+v"(r.c0), "+v"(r.c1), "+v"(r.c2), "+v"(r.c3) : "v"(r.a), "v"(r.b));
#pragma unroll
      for (int k = 0; k < K; k += 2) asm volatile("v_fma_f64 %0, %0, %2, %3\n\tv_fma_f64 %1, %1, %2, %3" : "+v"(r.f2), "+v"(r.f3) : "v"(r.g0), "v"(r.g1));
      asm volatile("v_mfma_f64_16x16x4_f64 %2, %4, %5, %2\n\t" : "+v"(r.c0), "+v"(r.c1), "+v"(r.c2), "+v"(r.c3) : "v"(r.a), "v"(r.b));
#pragma unroll
      for (int k = 0; k < K; k += 2) asm volatile("v_fma_f64 %0, %0, %2, %3\n\tv_fma_f64 %1, %1, %2, %3" : "+v"(r.f0), "+v"(r.f1) : "v"(r.g0), "v"(r.g1));
      asm volatile("v_mfma_f64_16x16x4_f64 %3, %4, %5, %3\n\t" : "+v"(r.c0), "+v"(r.c1), "+v"(r.c2), "+v"(r.c3) : "v"(r.a), "v"(r.b));
#pragma unroll
      for (int k = 0; k < K; k += 2) asm volatile("v_fma_f64 %0, %0, %2, %3\n\tv_fma_f64 %1, %1, %2, %3" : "+v"(r.f2), "+v"(r.f3) : "v"(r.g0), "v"(r.g1));
    }
  } else if (O >= M16_X4 && O <= M16_X16) {
    constexpr int K = (O == M16_X4) ? 4 : (O == M16_X8) ? 8 : 16;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        if (m == 0) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(r.c0) : "v"(r.a), "v"(r.b));
        if (m == 1) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(r.c1) : "v"(r.a), "v"(r.b));
        if (m == 2) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(r.c2) : "v"(r.a), "v"(r.b));
        if (m == 3) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(r.c3) : "v"(r.a), "v"(r.b));
#pragma unroll
        for (int k = 0; k < K; k += 4)
          asm volatile("v_xor_b32 %0, %0, %4\n\tv_xor_b32 %1, %1, %4\n\tv_xor_b32 %2, %2, %4\n\tv_xor_b32 %3, %3, %4" : "+v"(r.u0), "+v"(r.u1), "+v"(r.u2), "+v"(r.u3) : "v"(r.w0));
      }
    }
  } else if (O == M16_L2) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        if (m == 0) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(r.c0) : "v"(r.a), "v"(r.b));
        if (m == 1) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(r.c1) : "v"(r.a), "v"(r.b));
        if (m == 2) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(r.c2) : "v"(r.a), "v"(r.b));
        if (m == 3) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(r.c3) : "v"(r.a), "v"(r.b));
        asm volatile("ds_read_u16 %0, %2\n\tds_read_u16 %1, %2 offset:128\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r.u0), "=&v"(r.u1) : "v"(r.la) : "memory");
      }
    }
  } else if (O == M4_F2 || O == M4_F4) {
    constexpr int K = (O == M4_F2) ? 2 : 4;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        if (m == 0) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(r.s0) : "v"(r.a), "v"(r.b));
        if (m == 1) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(r.s1) : "v"(r.a), "v"(r.b));
        if (m == 2) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(r.s2) : "v"(r.a), "v"(r.b));
        if (m == 3) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(r.s3) : "v"(r.a), "v"(r.b));
#pragma unroll
        for (int k = 0; k < K; k += 2) asm volatile("v_fma_f64 %0, %0, %2, %3\n\tv_fma_f64 %1, %1, %2, %3" : "+v"(r.f0), "+v"(r.f1) : "v"(r.g0), "v"(r.g1));
      }
    }
  } else if (O == F_ONLY) {
    REP16(asm volatile("v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %4, %5\n\tv_fma_f64 %3, %3, %4, %5" : "+v"(r.f0), "+v"(r.f1), "+v"(r.f2), "+v"(r.f3) : "v"(r.g0), "v"(r.g1));)
  }
}

__device__ __forceinline__ void init6(R6 &r, unsigned wave, unsigned lane)
{
  const double z = 1e-3 * (double)(threadIdx.x & 15u);
  r.c0 = {z, z, z, z}; r.c1 = r.c0; r.c2 = r.c0; r.c3 = r.c0;
  r.a = 1e-3; r.b = 0.5 + z;
  r.s0 = r.s1 = r.s2 = r.s3 = z;
  r.f0 = 1.0 + z; r.f1 = 1.5; r.f2 = 2.5; r.f3 = 3.5; r.g0 = 0.999999; r.g1 = 1e-9;
  r.u0 = threadIdx.x * 2654435761u + 7u; r.u1 = r.u0 ^ 0x9E3779B9u; r.u2 = r.u1 * 3u; r.u3 = r.u2 + 11u; r.w0 = r.u0 + 1u;
  r.la = wave * 1024u + lane * 2u;
}
__device__ __forceinline__ double fold6(const R6 &r)
{
  return r.c0.x + r.c0.w + r.c1.x + r.c2.y + r.c3.z + r.s0 + r.s1 + r.s2 + r.s3 + r.f0 + r.f1 + r.f2 + r.f3 + r.u0 + r.u1 + r.u2 + r.u3;
}

template <int O>
__global__ void __launch_bounds__(256) k_p1(unsigned long long *out, double *sink, int iters)
{
  __shared__ unsigned lds[4096];
  const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  R6 r;
  init6(r, wave, lane);
  lds[threadIdx.x] = r.u0;
  __syncthreads();
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) body6<O>(r);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (lane == 0) out[blockIdx.x * 16 + wave] = t1 - t0;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = fold6(r) + lds[(threadIdx.x * 7u) & 1023u];
}

// part 2: 4 * NW wavefronts per workgroup; wavefronts w, w+4, w+8 share a SIMD.  Role of wavefront w / 4:
//   'M' = mfma16 x4 accumulators (prio given), 'F' = v_fma_f64 stream, 'X' = v_xor stream, 'm' = mfma16+4 fma interleaved
// The FIRST role runs a fixed number of iterations and raises a flag; the others count how many they complete until then.
struct P2Case { const char *name; char role[3]; int prio[3]; };

__global__ void __launch_bounds__(768) k_p2(unsigned long long *out, double *sink, int iters, int r0, int r1, int r2, int p0, int p1, int p2)
{
  __shared__ int done[4];
  const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  const unsigned simd = wave & 3u, slot = wave >> 2;
  const int role = (slot == 0) ? r0 : (slot == 1) ? r1 : r2;
  const int prio = (slot == 0) ? p0 : (slot == 1) ? p1 : p2;
  R6 r;
  init6(r, wave, lane);
  if (threadIdx.x < 4) done[threadIdx.x] = 0;
  __syncthreads();
  if (prio == 3) __builtin_amdgcn_s_setprio(3);
  else if (prio == 2) __builtin_amdgcn_s_setprio(2);
  else if (prio == 1) __builtin_amdgcn_s_setprio(1);
  unsigned long long t0, t1;
  long n = 0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  if (slot == 0) {
    for (int it = 0; it < iters; ++it) {
      if (role == 'M') body6<M16_4>(r);
      else if (role == 'm') body6<M16_F4>(r);
      else if (role == 'F') body6<F_ONLY>(r);
      else body6<M16_X16>(r);
      n++;
    }
    __hip_atomic_store(&done[simd], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  } else {
    while (__hip_atomic_load(&done[simd], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0 && n < 4 * (long)iters) {
      if (role == 'M') body6<M16_4>(r);
      else if (role == 'm') body6<M16_F4>(r);
      else if (role == 'F') body6<F_ONLY>(r);
      else if (role == 'X') { REP16(asm volatile("v_xor_b32 %0, %0, %4\n\tv_xor_b32 %1, %1, %4\n\tv_xor_b32 %2, %2, %4\n\tv_xor_b32 %3, %3, %4" : "+v"(r.u0), "+v"(r.u1), "+v"(r.u2), "+v"(r.u3) : "v"(r.w0));) }
      n++;
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (lane == 0) {
    out[(blockIdx.x * 16 + wave) * 2] = t1 - t0;
    out[(blockIdx.x * 16 + wave) * 2 + 1] = (unsigned long long)n;
  }
  sink[blockIdx.x * blockDim.x + threadIdx.x] = fold6(r);
}

// part 3: layouts.  tile: D[16][16] = sum_k A[16][4] * B[4][16];  lane l: A[l & 15][l >> 4], B[l >> 4][l & 15];
// D register v of lane l = D[(l >> 4) + 4 * v][l & 15].  Then E = A2 x D[rows 4v..4v+3] using D's register v as B.
__global__ void __launch_bounds__(64) k_layout(const double *A, const double *B, const double *A2, double *D, double *E)
{
  const int l = threadIdx.x;
  const double a = A[(l & 15) * 4 + (l >> 4)];
  const double b = B[(l >> 4) * 16 + (l & 15)];
  v4d c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int v = 0; v < 4; ++v) D[((l >> 4) + 4 * v) * 16 + (l & 15)] = c[v];
  // E[16][16] = sum over v of A2v[16][4] x D[4v .. 4v+3][16]
  v4d e = {0, 0, 0, 0};
  for (int v = 0; v < 4; ++v) {
    const double a2 = A2[(l & 15) * 16 + 4 * v + (l >> 4)];
    e = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, c[v], e, 0, 0, 0);
  }
  for (int v = 0; v < 4; ++v) E[((l >> 4) + 4 * v) * 16 + (l & 15)] = e[v];
}

typedef void (*kern_t)(unsigned long long *, double *, int);
template <int O>
static void fill(kern_t *t) { t[O] = k_p1<O>; fill<O + 1>(t); }
template <>
void fill<P1_COUNT>(kern_t *) {}

int main()
{
  kern_t fn[P1_COUNT];
  fill<0>(fn);
  const int iters = 400;
  unsigned long long *d_out;
  double *d_sink;
  (void)hipMalloc(&d_out, 256 * 16 * 2 * sizeof(unsigned long long));
  (void)hipMalloc(&d_sink, 256 * 1024 * sizeof(double));
  std::vector<unsigned long long> h(256 * 16 * 2);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);

  // ---- part 3 first: layouts ----
  {
    std::vector<double> A(64), B(64), A2(256), D(256), E(256), Dh(256), Eh(256);
    srand(7);
    for (auto &x : A) x = rand() / (double)RAND_MAX - 0.5;
    for (auto &x : B) x = rand() / (double)RAND_MAX - 0.5;
    for (auto &x : A2) x = rand() / (double)RAND_MAX - 0.5;
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        double s = 0;
        for (int k = 0; k < 4; ++k) s = fma(A[i * 4 + k], B[k * 16 + j], s);
        Dh[i * 16 + j] = s;
      }
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        double s = 0;
        for (int k = 0; k < 16; ++k) s = fma(A2[i * 16 + k], Dh[k * 16 + j], s);
        Eh[i * 16 + j] = s;
      }
    double *dA, *dB, *dA2, *dD, *dE;
    (void)hipMalloc(&dA, 64 * 8); (void)hipMalloc(&dB, 64 * 8); (void)hipMalloc(&dA2, 256 * 8); (void)hipMalloc(&dD, 256 * 8); (void)hipMalloc(&dE, 256 * 8);
    (void)hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(dA2, A2.data(), 256 * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dA, dB, dA2, dD, dE);
    (void)hipMemcpy(D.data(), dD, 256 * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(E.data(), dE, 256 * 8, hipMemcpyDeviceToHost);
    double md = 0, me = 0;
    int exact_d = 0;
    for (int i = 0; i < 256; ++i) {
      md = fmax(md, fabs(D[i] - Dh[i]));
      me = fmax(me, fabs(E[i] - Eh[i]));
      exact_d += (D[i] == Dh[i]);
    }
    printf("part 3 (layouts): max |D - host| = %.3e (%d of 256 equal to a k-ordered fma chain), max |E - host| = %.3e  [E uses D's registers as B operands]\n", md, exact_d, me);
  }

  printf("\npart 1: one wavefront per SIMD (256-thread workgroup per CU), ticks per MFMA gap (one MFMA + its fillers)\n");
  printf("%-40s %10s %10s %8s\n", "stream", "ticks/gap", "ns/gap", "tick/ns");
  for (int o = 0; o < P1_COUNT; ++o) {
    hipLaunchKernelGGL(fn[o], dim3(256), dim3(256), 0, 0, d_out, d_sink, iters);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(fn[o], dim3(256), dim3(256), 0, 0, d_out, d_sink, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(h.data(), d_out, 256 * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double t = 0;
    for (int wg = 0; wg < 256; ++wg)
      for (int w = 0; w < 4; ++w) t += (double)h[wg * 16 + w];
    t /= 256.0 * 4;
    const double per = t / (iters * (double)p1_mfma_per_body[o]);
    const double ns = ms * 1e6 / (iters * (double)p1_mfma_per_body[o]);
    printf("%-40s %10.2f %10.3f %8.2f\n", p1_name[o], per, ns, per / ns);
    fflush(stdout);
  }

  printf("\npart 2: wavefronts w, w+4, w+8 of a 768- (or 512-) thread workgroup share a SIMD; role A runs %d x 64 units, the others until A is done\n", iters);
  printf("units: M = 1 mfma16, m = 1 mfma16 + 4 v_fma_f64, F = 1 v_fma_f64, X = 1 v_xor_b32 (M16_X16 as A: 1 mfma + 16 xor)\n");
  printf("%-46s %12s %12s %12s\n", "case", "A ticks/unit", "B ticks/unit", "C ticks/unit");
  const P2Case cases[] = {
    {"M prio3 alone", {'M', 0, 0}, {3, 0, 0}},
    {"M prio3 | F", {'M', 'F', 0}, {3, 0, 0}},
    {"M prio0 | F", {'M', 'F', 0}, {0, 0, 0}},
    {"M prio3 | X", {'M', 'X', 0}, {3, 0, 0}},
    {"M prio3 | F | F", {'M', 'F', 'F'}, {3, 0, 0}},
    {"M prio3 | F | X", {'M', 'F', 'X'}, {3, 0, 0}},
    {"M prio0 | F prio1 | X prio1", {'M', 'F', 'X'}, {0, 1, 1}},
    {"m prio3 | F | X", {'m', 'F', 'X'}, {3, 0, 0}},
    {"m prio3 | F | F", {'m', 'F', 'F'}, {3, 0, 0}},
    {"F prio3 | F | F (no mfma, reference)", {'F', 'F', 'F'}, {3, 0, 0}},
    {"F alone", {'F', 0, 0}, {0, 0, 0}},
  };
  for (const P2Case &c : cases) {
    const int nw = c.role[2] ? 3 : (c.role[1] ? 2 : 1);
    hipLaunchKernelGGL(k_p2, dim3(256), dim3(256 * nw), 0, 0, d_out, d_sink, iters, c.role[0], c.role[1], c.role[2], c.prio[0], c.prio[1], c.prio[2]);
    (void)hipDeviceSynchronize();
    hipLaunchKernelGGL(k_p2, dim3(256), dim3(256 * nw), 0, 0, d_out, d_sink, iters, c.role[0], c.role[1], c.role[2], c.prio[0], c.prio[1], c.prio[2]);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h.data(), d_out, 256 * 16 * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double per[3] = {0, 0, 0};
    for (int s = 0; s < nw; ++s) {
      double t = 0, n = 0;
      for (int wg = 0; wg < 256; ++wg)
        for (int w = 0; w < 4; ++w) {
          t += (double)h[(wg * 16 + s * 4 + w) * 2];
          n += (double)h[(wg * 16 + s * 4 + w) * 2 + 1];
        }
      per[s] = (n > 0) ? t / (n * 64.0) : 0.0;
    }
    printf("%-46s %12.2f %12.2f %12.2f\n", c.name, per[0], per[1], per[2]);
    fflush(stdout);
  }
  (void)hipFree(d_out);
  (void)hipFree(d_sink);
  return 0;
}
