// How fast can a few STREAMING waves push write-through stores while the other waves of the workgroup keep the SIMDs
// busy?  (Forward rasterizer, DESIGN 4.1: the store path idles during the scan conversion.)
//   grid = 256 workgroups x 16 waves.  Waves [16 - S, 16) each issue `nst` wave-wide 16-byte stores (1 KB apiece)
//   into the workgroup's slice; the other waves run `iters` rounds of 32 independent FMAs (4 chains x 8).
//   mode bit 0: s_setprio 3 on the streaming waves; bit 1: plain stores instead of sc1; bit 2: streaming waves are
//   the OLDEST (0 .. S-1) instead of the youngest; bit 3: the compute waves interleave the stores themselves
//   (one store every `every` rounds; S = 0).
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
extern "C" __global__ void __launch_bounds__(1024)
storewave(float *out, long long *tbuf, int S, int nst, int iters, int mode, int every) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool oldest = mode & 4;
  const bool streamer = S > 0 && (oldest ? wave < S : wave >= 16 - S);
  float4 *base = reinterpret_cast<float4 *>(out) + (size_t)blockIdx.x * 64 * 64;   // 64 KB per workgroup
  const long long t0 = clock64();
  long long t1 = t0;
  if (streamer) {
    if (mode & 1) __builtin_amdgcn_s_setprio(3);
    const int k = oldest ? wave : wave - (16 - S);
    const v4u v = {0x42c80000u, 0x42c80000u, 0x42c80000u, 0x42c80000u};
    for (int i = 0; i < nst; i++) {
      float4 *p = base + ((k + i * S) & 63) * 64 + lane;
      if (mode & 2) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
      else asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    }
    t1 = clock64();
  } else {
    float a = lane, b = 1.0001f, c = 0.5f, d = 2.f, x = 1.00001f;
    const v4u v = {0x42c80000u, 0x42c80000u, 0x42c80000u, 0x42c80000u};
    int issued = 0;
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int r = 0; r < 8; r++) {
        a = __builtin_fmaf(a, x, 0.25f); b = __builtin_fmaf(b, x, 0.25f);
        c = __builtin_fmaf(c, x, 0.25f); d = __builtin_fmaf(d, x, 0.25f);
      }
      if ((mode & 8) && every > 0 && (i % every) == every - 1 && issued < nst) {
        float4 *p = base + ((wave + issued * 16) & 63) * 64 + lane;
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
        issued++;
      }
    }
    t1 = clock64();
    if (a + b + c + d == 12345.f) out[0] = a;
  }
  if (lane == 0) { tbuf[((size_t)blockIdx.x * 16 + wave) * 2] = t0; tbuf[((size_t)blockIdx.x * 16 + wave) * 2 + 1] = t1; }
}
extern "C" int storewave_launch(float *out, long long *tbuf, int S, int nst, int iters, int mode, int every, void *stream) {
  hipLaunchKernelGGL(storewave, dim3(256), dim3(1024), 0, (hipStream_t)stream, out, tbuf, S, nst, iters, mode, every);
  return (int)hipGetLastError();
}
