// LDS atomic throughput under same-address conflicts (gfx950): a wave-wide ds_add_u64 / ds_add_u32 / ds_add_f32
// whose 64 lanes hit D distinct addresses in runs of 64/D lanes; 16 waves per CU (1024 threads), all CUs.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ void k(unsigned long long *out, int D, int iters) {
  __shared__ unsigned long long s[16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  s[wave][lane] = 0;
  __syncthreads();
  const int slot = lane / (64 / D);
  unsigned long long *p = &s[wave][slot];
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) __hip_atomic_fetch_add(p, (unsigned long long)(lane + it), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else if (MODE == 1) __hip_atomic_fetch_add((unsigned *)p, (unsigned)(lane + it), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_fetch_add((float *)p, (float)(lane + it), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __syncthreads();
  out[blockIdx.x * 1024 + threadIdx.x] = s[wave][lane];
}
template <int MODE> void run(const char *name) {
  unsigned long long *out; hipMalloc(&out, 256 * 1024 * 8);
  const int iters = 2000;
  for (int D : {1, 2, 4, 8, 16, 32, 64}) {
    k<MODE><<<256, 1024>>>(out, D, iters);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); k<MODE><<<256, 1024>>>(out, D, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-10s D=%2d distinct addresses (multiplicity %2d): %.1f ns per wave-instruction per CU (16 waves issuing)\n", name, D, 64 / D, ms * 1e6 / (iters * 16.0));
  }
}
int main() { run<0>("ds_add_u64"); run<1>("ds_add_u32"); run<2>("ds_add_f32"); return 0; }
