// capi.hip -- library introspection entry points of libspherehand_hip.so.
#include <string.h>

#include "common.h"

extern "C" int shr_abi_version(void) { return 2; }

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
