/* tools/cloop.c -- the headline step (forward + backward through the C ABI) launched `steps` times from a C loop:
 * under rocprofv3 the Python + tracer cost per launch exceeds these 7-us kernels' duration and the dispatches stop
 * overlapping; from C the host keeps ahead of the device with the tracer attached as well (tools/prof_cloop.py).
 * gcc -O2 -shared -fPIC -o tools/libcloop.so tools/cloop.c -ldl */
#include <dlfcn.h>
#include <stdint.h>
typedef int (*fwd_t)(const float *, int, int, int, int, float *, uint8_t *, int, void *);
typedef int (*bwd_t)(const float *, const float *, const uint8_t *, int, int, int, int, float *, void *);
int cloop_run(const char *lib, const float *spheres, int N, int J, int H, int W, float *depth, uint8_t *owner,
              const float *grad, float *gs, int flags, int steps, int what, void *stream) {
  void *h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
  if (!h) return -1;
  fwd_t fwd = (fwd_t)dlsym(h, "shr_sphere_raster_fwd_ex");
  bwd_t bwd = (bwd_t)dlsym(h, "shr_sphere_raster_bwd");
  if (!fwd || !bwd) return -2;
  for (int i = 0; i < steps; i++) {
    if (what & 1) { int e = fwd(spheres, N, J, H, W, depth, owner, flags, stream); if (e) return e; }
    if (what & 2) { int e = bwd(spheres, grad, owner, N, J, H, W, gs, stream); if (e) return e; }
  }
  return 0;
}
