// Which way does v_sqrt_f32 err on gfx950?  For every fp32 in [lo_bits, hi_bits): compare the hardware estimate with the
// correctly rounded root (sqrtf) and count equal / one ulp low / one ulp high / further off.
#include <hip/hip_runtime.h>
#include <stdint.h>
__global__ void classify(unsigned lo, unsigned hi, unsigned long long *out) {
  unsigned long long eq = 0, low = 0, high = 0, far = 0;
  for (unsigned long long b = (unsigned long long)lo + blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; b < hi;
       b += (unsigned long long)gridDim.x * blockDim.x) {
    const float x = __uint_as_float((unsigned)b);
    const float h = __builtin_amdgcn_sqrtf(x), r = sqrtf(x);
    const int d = (int)__float_as_uint(h) - (int)__float_as_uint(r);
    if (d == 0) eq++; else if (d == -1) low++; else if (d == 1) high++; else far++;
  }
  atomicAdd(out + 0, eq); atomicAdd(out + 1, low); atomicAdd(out + 2, high); atomicAdd(out + 3, far);
}
extern "C" int exp_sqrt_classify(unsigned lo, unsigned hi, unsigned long long *out, void *stream) {
  hipLaunchKernelGGL(classify, dim3(4096), dim3(256), 0, (hipStream_t)stream, lo, hi, out);
  return (int)hipGetLastError();
}
