// launch-floor microbenchmark: empty / sleeping kernels with the rasterizer's launch shape
#include <hip/hip_runtime.h>
extern "C" __global__ void __launch_bounds__(1024) k_empty(int *p) { extern __shared__ int sm[]; if (p && threadIdx.x == 9999) p[0] = sm[0]; }
extern "C" __global__ void __launch_bounds__(1024) k_spin(int *p, int cycles) {
  extern __shared__ int sm[];
  const long long t0 = clock64();
  while (clock64() - t0 < cycles) {}
  if (p && threadIdx.x == 9999) p[0] = sm[0];
}
extern "C" int run(int which, int grid, int block, int lds, int cycles, int reps, float *ms, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  hipFuncSetAttribute((const void *)k_empty, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void *)k_spin, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; w++) {
    hipEventRecord(e0, s);
    for (int i = 0; i < reps; i++) {
      if (which == 0) hipLaunchKernelGGL(k_empty, dim3(grid), dim3(block), lds, s, (int *)nullptr);
      else hipLaunchKernelGGL(k_spin, dim3(grid), dim3(block), lds, s, (int *)nullptr, cycles);
    }
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
  }
  hipEventElapsedTime(ms, e0, e1);
  return (int)hipGetLastError();
}
