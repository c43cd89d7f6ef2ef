"""Ablation copy of data_to_model_kernel: mode bits skip phases."""
src = open('spherehand_amd/csrc/data_to_model.hip').read()
a = src.index('template <bool WANT_GRAD>\n__global__')
b = src.index('}  // namespace shr')
k = src[a:b]
k = k.replace('data_to_model_kernel(', 'exp_d2m(')
k = k.replace('float *__restrict__ grad_centres) {', 'float *__restrict__ grad_centres, int mode) {')
# bit0: skip search+grad entirely (load, scan, compact only); bit1: no pruning (all candidates); bit2: skip owner reduction
k = k.replace('      for (int i0 = wave * 64; i0 < total; i0 += kD2mThreads) {', '      if (!(mode & 1)) for (int i0 = wave * 64; i0 < total; i0 += kD2mThreads) {')
k = k.replace('          unsigned long long cand = __ballot(rem <= reach && rem != inf);', '          unsigned long long cand = (mode & 2) ? 0ull : __ballot(rem <= reach && rem != inf);')
k = k.replace('          for (int it = 0; it < kD2mSeeds; it++) {', '          for (int it = 0; it < ((mode & 16) ? 0 : kD2mSeeds); it++) {')
k = k.replace('          unsigned long long todo = __ballot(owner >= 0);', '          unsigned long long todo = (mode & 4) ? 0ull : __ballot(owner >= 0);\n          if (mode & 4) loss += gx + gy + gz;')
# bit3: skip compaction stores
k = k.replace('            s_q[slot] = e;', '            if (!(mode & 8)) s_q[slot] = e;')
exp = '#include "../spherehand_amd/csrc/common.h"\nnamespace shr {\n' + src[src.index('constexpr int kD2mThreads'):a] + k + '''}
extern "C" int exp_d2m_launch(const float *depth, const float *centres, const float *radii, int N, int J, int H, int W,
                              float *loss_sum, float *grad, int mode, void *stream) {
  hipLaunchKernelGGL(shr::exp_d2m<true>, dim3(N), dim3(shr::kD2mThreads), 0, (hipStream_t)stream, depth, (const int *)nullptr, centres, radii, J, H, W, loss_sum, grad, mode);
  return (int)hipGetLastError();
}
'''
open('tools/exp_d2m.hip', 'w').write(exp)
