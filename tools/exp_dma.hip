// LDS-DMA sanity: global_load_lds_dwordx4 / _dword place lane l's bytes at m0 + l*16 / l*4; exec-masked lanes skip
#include <hip/hip_runtime.h>
#include <stdint.h>
__device__ __forceinline__ void glds16(const void *gsrc, uint32_t lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds4(const void *gsrc, uint32_t lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
extern "C" __global__ void __launch_bounds__(256) k_dma(const float4 *g16, const uint32_t *g4, float4 *o16, uint32_t *o4, int nvalid) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float4 *s16 = reinterpret_cast<float4 *>(smem);                 // 4 waves x 64 x 16 B
  uint32_t *s4 = reinterpret_cast<uint32_t *>(smem + 4096);       // 4 waves x 64 x 4 B
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 1024 + 256; i += 256) reinterpret_cast<uint32_t *>(smem)[i] = 0xdeadbeefu;
  __syncthreads();
  const uint32_t base16 = (uint32_t)(uintptr_t)(s16 + wave * 64), base4 = (uint32_t)(uintptr_t)(s4 + wave * 64);
  if (tid < nvalid) {
    glds16(g16 + tid, __builtin_amdgcn_readfirstlane(base16));
    glds4(g4 + tid, __builtin_amdgcn_readfirstlane(base4));
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  o16[tid] = s16[tid];
  o4[tid] = s4[tid];
}
extern "C" int run_dma(const void *g16, const void *g4, void *o16, void *o4, int nvalid, void *stream) {
  hipLaunchKernelGGL(k_dma, dim3(1), dim3(256), 4096 + 1024, (hipStream_t)stream, (const float4 *)g16, (const uint32_t *)g4, (float4 *)o16, (uint32_t *)o4, nvalid);
  return (int)hipGetLastError();
}
