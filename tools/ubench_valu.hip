// ubench_valu.hip -- measures the issue rate of the VALU forms the GMM kernel
// can use on gfx950: plain v_mul/v_add (VGPR and SGPR operand) vs packed
// v_pk_mul_f32/v_pk_add_f32.  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o /tmp/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP 64
template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, int iters, float s) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, ps = {s, s};
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int r = 0; r < REP; r++) {
      if (MODE == 0) {  // 8 independent v_mul_f32 (VGPR operands)
        asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                     "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
      } else if (MODE == 1) {  // same with an SGPR operand
        asm volatile("v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n v_mul_f32 %2, %8, %2\n v_mul_f32 %3, %8, %3\n"
                     "v_mul_f32 %4, %8, %4\n v_mul_f32 %5, %8, %5\n v_mul_f32 %6, %8, %6\n v_mul_f32 %7, %8, %7\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(s));
      } else if (MODE == 2) {  // 4 v_pk_mul_f32 = 8 multiplies
        asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(ps));
      } else if (MODE == 3) {  // dependent chain of 8 v_add on ONE register (latency)
        asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n"
                     "v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n"
                     : "+v"(a0) : "v"(s));
      } else if (MODE == 4) {  // 4 v_pk_add_f32
        asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(ps));
      } else if (MODE == 5) {  // v_pk_mul_f32 with an SGPR pair operand
        asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "s"(ps));
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <int MODE>
void run(const char *name, float *d, int wpb_blocks) {
  const int iters = 2000;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(wpb_blocks), dim3(256), 0, 0, d, 10, 1.0001f);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(wpb_blocks), dim3(256), 0, 0, d, iters, 1.0001f);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double ops = (double)wpb_blocks * 256 * iters * REP * 8;  // scalar fp32 operations
  printf("%-34s blocks=%5d  %.3f ms  %.2f Tops/s (fp32 scalar-equivalent ops)\n", name, wpb_blocks, ms, ops / ms / 1e9);
}

int main() {
  float *d; hipMalloc(&d, 4 * 256 * 8192);
  for (int blocks : {256 * 2, 256 * 4, 256 * 8}) {
    run<0>("v_mul_f32 vgpr x8 indep", d, blocks);
    run<1>("v_mul_f32 sgpr-operand x8 indep", d, blocks);
    run<2>("v_pk_mul_f32 x4 (8 mul)", d, blocks);
    run<5>("v_pk_mul_f32 sgpr-pair x4 (8 mul)", d, blocks);
    run<4>("v_pk_add_f32 x4 (8 add)", d, blocks);
    run<3>("v_add_f32 dependent chain x8", d, blocks);
  }
  return 0;
}
