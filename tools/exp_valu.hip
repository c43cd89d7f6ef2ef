// VALU issue-rate microbenchmark (gfx950): cycles per wave-instruction for plain / packed fp32 FMA,
// v_sqrt_f32, v_cmp+v_cndmask, with 1, 2, 4 waves per SIMD (256 / 512 / 1024-thread workgroups, one per CU).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int MODE>
__global__ void k(float *out, long long *cycles, int iters) {
  float a[16];
  for (int i = 0; i < 16; i++) a[i] = threadIdx.x * 0.001f + i;
  float b = 1.0001f, c = 0.5f;
  typedef float float2_ __attribute__((ext_vector_type(2)));
  __syncthreads();
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 16; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
    } else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        float2_ v = {a[i], a[i + 1]}, bb = {b, b}, cc = {c, c};
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(bb), "v"(cc));
        a[i] = v.x; a[i + 1] = v.y;
      }
    } else if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < 16; i++) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
    } else if (MODE == 3) {
#pragma unroll
      for (int i = 0; i < 16; i++) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[i]) : "v"(b), "v"(c) : "vcc");
    } else if (MODE == 4) {
#pragma unroll
      for (int i = 0; i < 16; i++) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
    } else if (MODE == 5) {
#pragma unroll
      for (int i = 0; i < 16; i++) asm volatile("v_min_f32 %0, |%0|, %1" : "+v"(a[i]) : "v"(b));
    } else if (MODE == 6) {
#pragma unroll
      for (int i = 0; i < 16; i++) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 16; i++) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char *name, int n_per_iter) {
  float *out; long long *cyc;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 8);
  const int iters = 2000;
  for (int threads : {64, 256, 512, 1024}) {
    k<MODE><<<256, threads>>>(out, cyc, iters);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<MODE><<<256, threads>>>(out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const double waves_per_simd = threads / 256.0;
    const double instr = (double)iters * n_per_iter;
    printf("%-22s threads %4d: %.2f clk-counter ticks per wave-instr, wall %.1f us -> %.2f ns per instr per wave; per SIMD %.2f ns/instr\n",
           name, threads, h[0] / instr, ms * 1e3, ms * 1e6 / instr, ms * 1e6 / instr / (waves_per_simd < 1 ? 1 : waves_per_simd));
  }
}

int main() {
  run<0>("v_fma_f32", 16);
  run<1>("v_pk_fma_f32", 8);
  run<2>("v_sqrt_f32", 16);
  run<6>("v_rsq_f32", 16);
  run<3>("v_cmp+v_cndmask", 32);
  run<4>("v_sub_f32", 16);
  run<5>("v_min_f32 |x|", 16);
  return 0;
}
