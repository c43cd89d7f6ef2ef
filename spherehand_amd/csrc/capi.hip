// capi.hip -- library introspection entry points of libspherehand_hip.so.
#include <string.h>

#include "sphere_zbuf.h"

extern "C" int shr_abi_version(void) { return 13; }

extern "C" const char *shr_error_string(int code) {
  switch (code) {
    case SHR_OK: return "ok";
    case SHR_EINVAL: return "invalid argument (null pointer, non-positive size or misaligned buffer)";
    case SHR_ETOOLARGE: return "size beyond the kernel's indexing limits";
    case SHR_ENODEVICE: return "no gfx950 (MI355X) device is current";
    default: break;
  }
  if (code > 0) return hipGetErrorString((hipError_t)code);
  return "unknown spherehand error";
}

extern "C" int shr_device_info(char *name_host, int name_len, int *num_cu_host) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return SHR_ENODEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return SHR_ENODEVICE;
  if (name_host && name_len > 0) {
    strncpy(name_host, prop.gcnArchName, (size_t)name_len - 1);
    name_host[name_len - 1] = 0;
  }
  if (num_cu_host) *num_cu_host = prop.multiProcessorCount;
  return strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? SHR_OK : SHR_ENODEVICE;
}

// Self-test hook: counts fp32 bit patterns in [lo_bits, hi_bits) for which the
// rasterizer's sqrt_rn() differs from the correctly rounded sqrtf().
__global__ void sqrt_selftest_kernel(unsigned lo, unsigned hi, unsigned long long *mismatches) {
  unsigned long long bad = 0;
  for (unsigned long long b = (unsigned long long)lo + blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x;
       b < hi; b += (unsigned long long)gridDim.x * blockDim.x) {
    const float x = __uint_as_float((unsigned)b);
    bad += (__float_as_uint(shr::sqrt_rn(x)) != __float_as_uint(sqrtf(x)));
  }
  if (bad) atomicAdd(mismatches, bad);
}

extern "C" int shr_selftest_sqrt(unsigned lo_bits, unsigned hi_bits, unsigned long long *mismatches, void *stream) {
  if (!mismatches || hi_bits < lo_bits) return SHR_EINVAL;
  hipLaunchKernelGGL(sqrt_selftest_kernel, dim3(4096), dim3(256), 0, (hipStream_t)stream, lo_bits, hi_bits,
                     mismatches);
  return (int)hipGetLastError();
}
