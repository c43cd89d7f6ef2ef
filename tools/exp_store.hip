// store-issue microbenchmark: how long does a wave take to ISSUE k wave-wide stores, and to retire them?
#include <hip/hip_runtime.h>
extern "C" __global__ void __launch_bounds__(1024) k_store(float4 *out, long long *tb, int nstore, int active_waves, int mode) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nw = blockDim.x >> 6;
  float4 *base = out + (size_t)blockIdx.x * nw * nstore * 64;
  __syncthreads();
  const long long t0 = clock64();
  long long t1 = t0, t2 = t0;
  if (wave < active_waves) {
    const float4 v = make_float4(1.f, 2.f, 3.f, (float)tid);
    if (mode == 0) {
      for (int i = 0; i < nstore; i++) base[((size_t)i * nw + wave) * 64 + lane] = v;   // 1 KB per instr
    } else {
      float *b1 = reinterpret_cast<float *>(base);
      for (int i = 0; i < nstore; i++) b1[((size_t)i * nw + wave) * 64 + lane] = v.w;    // 256 B per instr
    }
    t1 = clock64();
    __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0) expcnt(0) lgkmcnt(0)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    t2 = clock64();
  }
  if (lane == 0) {
    long long *t = tb + ((size_t)blockIdx.x * nw + wave) * 4;
    t[0] = t0; t[1] = t1; t[2] = t2;
  }
}
extern "C" int launch_store(float *out, long long *tb, int grid, int nstore, int active, int mode, void *stream) {
  hipLaunchKernelGGL(k_store, dim3(grid), dim3(1024), 0, (hipStream_t)stream, (float4 *)out, tb, nstore, active, mode);
  return (int)hipGetLastError();
}
