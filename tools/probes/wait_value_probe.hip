// Probe: does hipStreamWaitValue32 on signal memory hold a stream until a kernel on another stream bumps the value?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ void bump_after(unsigned *p, long long cycles, int add) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) {}
  if (threadIdx.x == 0) __hip_atomic_fetch_add(p, (unsigned)add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void mark(unsigned long long *out) { *out = wall_clock64(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
  int can = 0; CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
  printf("CanUseStreamWaitValue = %d\n", can);
  for (int kind = 0; kind < 2; kind++) {
    unsigned *p = nullptr; unsigned long long *t = nullptr;
    if (kind == 0) CK(hipExtMallocWithFlags((void **)&p, 8, hipMallocSignalMemory)); else CK(hipMalloc((void **)&p, 8));
    CK(hipMemset(p, 0, 8)); CK(hipMalloc((void **)&t, 16));
    hipStream_t a, b; CK(hipStreamCreate(&a)); CK(hipStreamCreate(&b));
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(mark, dim3(1), dim3(1), 0, a, t);
    hipLaunchKernelGGL(bump_after, dim3(1), dim3(64), 0, a, p, 5000000ll /* 50 ms at 100 MHz */, 7);
    hipError_t we = hipStreamWaitValue32(b, p, 7, hipStreamWaitValueGte, 0xffffffffu);
    printf("%s memory: hipStreamWaitValue32 -> %s\n", kind == 0 ? "signal" : "plain device", hipGetErrorString(we));
    hipLaunchKernelGGL(mark, dim3(1), dim3(1), 0, b, t + 1);
    CK(hipDeviceSynchronize());
    unsigned long long h[2]; CK(hipMemcpy(h, t, 16, hipMemcpyDeviceToHost));
    printf("  stream b ran %.2f ms after stream a started (50 ms = it waited for the value)\n", (double)(h[1] - h[0]) / 1e5);
    (void)hipGetLastError();
  }
  return 0;
}
