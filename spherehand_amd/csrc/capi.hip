// capi.hip -- library introspection entry points of libspherehand_hip.so.
#include <string.h>

#include "sphere_zbuf.h"

extern "C" int shr_abi_version(void) { return 22; }

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

// Self-test hook: tri_pixel_depth's shared-reciprocal divisions against the plain ones on pseudo-random operands (weights
// in [0, 1] with exponents down to 2^-70 and exact zeros and ones among them, corner depths over 2^-45 .. 2^45 and negative
// ones: both sides of every guard) -- counts the cases whose depth differs in any bit.
__device__ __forceinline__ unsigned st_hash(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__global__ void division_selftest_kernel(unsigned seed, unsigned per_thread, unsigned long long *mismatches) {
  unsigned long long bad = 0;
  unsigned h = st_hash(seed ^ (blockIdx.x * blockDim.x + threadIdx.x) * 0x9e3779b9u);
  auto next = [&]() { h = st_hash(h + 0x6d2b79f5u); return h; };
  auto weight = [&]() {
    const unsigned r = next();
    if ((r & 15u) == 0u) return 0.f;
    if ((r & 15u) == 1u) return 1.f;
    const int e = (r & 16u) ? -(int)((r >> 5) % 71u) : -(int)((r >> 5) % 8u);          // down to 2^-70, mostly near 1
    const float m = 1.0f + (float)(next() >> 9) * 0x1p-23f;
    return fminf(ldexpf(m, e), 1.f);
  };
  auto depth_z = [&]() {
    const unsigned r = next();
    const int e = (r & 3u) ? (int)((r >> 2) % 12u) : (int)((r >> 2) % 91u) - 45;       // mostly 1 .. 2^11, sometimes anything
    const float m = 1.0f + (float)(next() >> 9) * 0x1p-23f;
    const float z = ldexpf(m, e);
    return ((r >> 20) & 3u) == 0u ? -z : z;
  };
  for (unsigned i = 0; i < per_thread; i++) {
    const float w0 = weight(), w1 = weight(), w2 = weight();
    const float w_sum = (w0 + w1) + w2;
    const float pz[3] = {depth_z(), depth_z(), depth_z()};
    const bool tame = shr::div_tame_z(pz[0]) && shr::div_tame_z(pz[1]) && shr::div_tame_z(pz[2]);
    float rz[3] = {0.f, 0.f, 0.f};
    if (tame) { rz[0] = shr::div_rcp_refined(pz[0]); rz[1] = shr::div_rcp_refined(pz[1]); rz[2] = shr::div_rcp_refined(pz[2]); }
    const float a = shr::tri_pixel_depth(w0, w1, w2, w_sum, pz, rz, tame);
    const float q0 = w0 / w_sum, q1 = w1 / w_sum, q2 = w2 / w_sum;
    const float b = 1.0f / ((q0 / pz[0] + q1 / pz[1]) + q2 / pz[2]);
    bad += (__float_as_uint(a) != __float_as_uint(b)) && !(a != a && b != b);
  }
  if (bad) atomicAdd(mismatches, bad);
}

extern "C" int shr_selftest_division(unsigned seed, unsigned per_thread, unsigned long long *mismatches, void *stream) {
  if (!mismatches) return SHR_EINVAL;
  hipLaunchKernelGGL(division_selftest_kernel, dim3(4096), dim3(256), 0, (hipStream_t)stream, seed, per_thread, mismatches);
  return (int)hipGetLastError();
}

// Launch-floor probe (bench.py `roofline.launch_floor`): the sphere forward's launch shape at one crop per CU -- N
// workgroups of 1024 threads, `lds_bytes` of dynamic LDS (the forward takes the whole CU's: one workgroup per CU) --
// doing NOTHING but the forward's memory traffic: the crop's 16 J record bytes in, the H x W fp32 depth image and the
// owner bytes of rows [row0, row1) out, all as full-line 16-byte stores with the forward's cache policy.  What this
// takes is the floor of ANY kernel of that shape over those bytes: dispatch of one workgroup per CU, one request
// round trip, the drain of the write queues, the end-of-kernel release.
__global__ void __launch_bounds__(1024)
launch_floor_kernel(const float4 *__restrict__ spheres, int J, int H, int W, int row0, int row1,
                    float *__restrict__ depth, uint8_t *__restrict__ argmin) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float4 *s_rec = reinterpret_cast<float4 *>(smem);
  const int n = blockIdx.x, tid = threadIdx.x;
  if (tid < J) {   // (kept alive: the request is issued and awaited like the forward's)
    const float4 t = spheres[(size_t)n * J + tid];
    s_rec[tid] = t;
    asm volatile("" : : "v"(t.x), "v"(t.y), "v"(t.z), "v"(t.w));
  }
  float4 *out4 = reinterpret_cast<float4 *>(depth + (size_t)n * H * W);
  const int nchunk = (H * W) >> 2;
  const float4 bg = make_float4(shr::kBackground, shr::kBackground, shr::kBackground, shr::kBackground);
  for (int c = tid; c < nchunk; c += 1024) shr::stream_store(out4 + c, bg);
  if (argmin) {
    uint4 *a16 = reinterpret_cast<uint4 *>(argmin + (size_t)n * H * W + (size_t)row0 * W);
    const int npiece = ((row1 - row0) * W) >> 4;
    const uint4 none = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
    for (int c = tid; c < npiece; c += 1024) shr::stream_store(a16 + c, none);
  }
}

extern "C" int shr_selftest_launch_floor(const float *spheres, int N, int J, int H, int W, int row0, int row1,
                                         float *depth, uint8_t *argmin, int lds_bytes, void *stream) {
  if (N == 0) return SHR_OK;
  if (!spheres || !depth || N < 0 || J <= 0 || J > SHR_MAX_SPHERES || H <= 0 || W <= 0 || (W & 15) || row0 < 0 ||
      row1 < row0 || row1 > H || lds_bytes < 1024 || lds_bytes > 160 * 1024)
    return SHR_EINVAL;
  if ((((uintptr_t)spheres | (uintptr_t)depth | (uintptr_t)argmin) & 15u) != 0) return SHR_EINVAL;
  const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(launch_floor_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(launch_floor_kernel, dim3((unsigned)N), dim3(1024), (size_t)lds_bytes, (hipStream_t)stream,
                     reinterpret_cast<const float4 *>(spheres), J, H, W, row0, row1, depth, argmin);
  return (int)hipGetLastError();
}
