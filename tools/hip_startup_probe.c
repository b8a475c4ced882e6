/* The floor under any HIP program's start-up on this box, call by call (tools/cli_startup.py): what of the drop-in
 * programs' 0.2-0.3 s is the runtime's and what is this library's.
 *   gcc -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -o /tmp/hip_startup_probe tools/hip_startup_probe.c -L/opt/rocm/lib -lamdhip64 */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double now_ms(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return 1e3 * (double)ts.tv_sec + 1e-6 * (double)ts.tv_nsec;
}
#define STEP(name, call)                                              \
  do {                                                                \
    const double t0_ = now_ms();                                      \
    const int e_ = (int)(call);                                       \
    printf("%-28s %8.2f ms (%d)\n", name, now_ms() - t0_, e_);        \
  } while (0)

int main(void)
{
  int n = 0, cus = 0;
  hipDeviceProp_t prop;
  void *d = NULL;
  char *h = (char *)calloc(1, 1 << 20);
  char name[64];
  STEP("hipInit", hipInit(0));
  STEP("hipGetDeviceCount", hipGetDeviceCount(&n));
  STEP("hipDeviceGetAttribute(CUs)", hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  STEP("hipDeviceGetName", hipDeviceGetName(name, sizeof(name), 0));
  STEP("hipGetDeviceProperties", hipGetDeviceProperties(&prop, 0));
  STEP("hipSetDevice", hipSetDevice(0));
  void *pin = NULL, *pin_dev = NULL;
  STEP("hipHostMalloc 256 KiB mapped", hipHostMalloc(&pin, 256 << 10, hipHostMallocMapped));
  STEP("hipHostGetDevicePointer", hipHostGetDevicePointer(&pin_dev, pin, 0));
  STEP("hipMalloc 1 MiB", hipMalloc(&d, 1 << 20));
  if (getenv("PROBE_SMALL_FIRST")) {
    STEP("hipMemcpyAsync H2D 4 B (first)", hipMemcpyAsync(d, h, 4, hipMemcpyHostToDevice, NULL));
    STEP("hipStreamSynchronize", hipStreamSynchronize(NULL));
  }
  if (getenv("PROBE_PINNED_FIRST")) {
    STEP("hipMemcpyAsync H2D 64K pinned", hipMemcpyAsync(d, pin, 64 << 10, hipMemcpyHostToDevice, NULL));
    STEP("hipStreamSynchronize", hipStreamSynchronize(NULL));
    STEP("hipMemcpyAsync D2H 64K pinned", hipMemcpyAsync(pin, d, 64 << 10, hipMemcpyDeviceToHost, NULL));
    STEP("hipStreamSynchronize", hipStreamSynchronize(NULL));
  }
  STEP("hipMemcpyAsync H2D 1 MiB", hipMemcpyAsync(d, h, 1 << 20, hipMemcpyHostToDevice, NULL));
  STEP("hipStreamSynchronize", hipStreamSynchronize(NULL));
  STEP("hipMemcpyAsync H2D 4 B", hipMemcpyAsync(d, h, 4, hipMemcpyHostToDevice, NULL));
  STEP("hipStreamSynchronize", hipStreamSynchronize(NULL));
  STEP("hipHostFree", hipHostFree(pin));
  STEP("hipMemcpyAsync D2H 4 B", hipMemcpyAsync(h, d, 4, hipMemcpyDeviceToHost, NULL));
  STEP("hipStreamSynchronize", hipStreamSynchronize(NULL));
  STEP("hipFree", hipFree(d));
  printf("device: %s, %d CUs (%s)\n", name, cus, prop.gcnArchName);
  return 0;
}
