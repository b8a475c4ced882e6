// What V_PERMLANE32_SWAP_B32 (gfx950) moves: r = __builtin_amdgcn_permlane32_swap(a, b, false, false) with a = 1000 + lane,
// b = 2000 + lane; prints r[0] and r[1] of lanes 0, 1, 31, 32, 33, 63.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/permlane_probe tools/ubench/permlane_probe.hip && /tmp/permlane_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned vs_u2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned *out)
{
  const unsigned lane = threadIdx.x;
  vs_u2 r = __builtin_amdgcn_permlane32_swap(1000u + lane, 2000u + lane, false, false);
  out[lane] = r[0];
  out[64 + lane] = r[1];
}
int main()
{
  unsigned *d, h[128];
  hipMalloc((void **)&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const int lanes[] = {0, 1, 31, 32, 33, 63};
  for (int l : lanes) printf("lane %2d: r[0] = %u  r[1] = %u\n", l, h[l], h[64 + l]);
  return 0;
}
