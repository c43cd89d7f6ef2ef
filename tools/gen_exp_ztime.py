"""Generate tools/exp_ztime.hip: a copy of sphere_zbuf_fwd_kernel with clock64 stamps."""
src = open('spherehand_amd/csrc/sphere_zbuf.h').read()
# NOTE: -DEXP_FAKELAYOUT is applied to the shared header through a macro below (timing only: wrong results)
start = src.index('template <bool OWNER, bool VEC4, bool POW2, bool PERSIST, bool BOX, bool TABLE = false>\n__global__ void __launch_bounds__(1024)\nsphere_zbuf_fwd_kernel')
end = src.index('// ---------------------------------------------------------------------------\n// Backward with the forward')
k = src[start:end]
k = k.replace('sphere_zbuf_fwd_kernel(', 'exp_zfwd_t(')
assert 'int w4_shift_flags, int shares, int zcells_, AxisK axk) {' in k
k = k.replace('int w4_shift_flags, int shares, int zcells_, AxisK axk) {', 'int w4_shift_flags, int shares, int zcells_, AxisK axk, long long *tbuf) {\n  const long long T0 = clock64();\n  long long T3b = 0, T4 = 0, T5 = 0, NP = 0; long long wclk[4] = {0, 0, 0, 0}; long long T1c = 0;')
k = k.replace('  // (the box is not known yet:', '  const long long T1 = clock64();\n  // (the box is not known yet:')
k = k.replace('  __syncthreads();\n  if (!(list_wave || bg_wave || tab_wave)) sph = s_sph[lane];', '  const long long T2 = clock64();\n  __syncthreads();\n  const long long T3 = clock64();\n  if (!(list_wave || bg_wave || tab_wave)) sph = s_sph[lane];')
k = k.replace('if (pf_wave && has_next) s_next[lane] = sph_next;   // (arrived long ago: the wave\'s own scan slice lies in between)\n    __syncthreads();', 'if (pf_wave && has_next) s_next[lane] = sph_next;\n    T4 = clock64();\n    __syncthreads();\n    T5 = clock64();')
k = k.replace('    // ---- scan-convert the chunk list', '    T3b = clock64();\n    // ---- scan-convert the chunk list')
k = k.replace('''  // Waves 1..kBgWaves store the background rows while wave 0 builds the list: they are the''', '''  const long long T1b = clock64();\n  // Waves 1..kBgWaves store the background rows while wave 0 builds the list: they are the''')
assert k.count('T1b = clock64()') == 1
assert k.count('T3b = clock64()') == 1 and k.count('T4 = clock64()') == 1 and k.count('T2 = clock64()') == 1 and k.count('T1 = clock64()') == 1
# ablations of the scan conversion (compile with -DEXP_NOATOMIC / -DEXP_NOFIX / -DEXP_PLAIN_STORE)
_atom = '''              if (OWNER)
                atomicMin(reinterpret_cast<unsigned long long *>(cell),
                          ((unsigned long long)depth_key(d) << 32) | jv);'''
assert _atom in k
k = k.replace(_atom, '''#if defined(EXP_NOATOMIC)
            if (OWNER) asm volatile("" ::"v"(cell), "v"(depth_key(d)), "s"(j));
#elif defined(EXP_PLAIN_STORE)
            if (OWNER) *reinterpret_cast<unsigned long long *>(cell) = ((unsigned long long)depth_key(d) << 32) | (unsigned)j;
#elif defined(EXP_MIN32)
            if (OWNER) atomicMin(reinterpret_cast<unsigned int *>(cell) + 1, depth_key(d));
#else
            if (OWNER)
              atomicMin(reinterpret_cast<unsigned long long *>(cell),
                        ((unsigned long long)depth_key(d) << 32) | jv);
#endif''')
k = k.replace('          if (has_b) {  // wave-uniform; branch-free up to the atomics', '''#ifdef EXP_NOBODY
          asm volatile("" ::"v"(qa), "v"(qb), "v"(cell_a));
          return;
#endif
          if (has_b) {  // wave-uniform; branch-free up to the atomics''')
k = k.replace('sqrt_rn(', 'EXP_SQRT(')
k = k.replace('[](int) {}, RunTab{s_tab, s_run});', """[](int) {},
#ifdef SHR_EXP_WALKCLK
          RunTab{s_tab, s_run}
#else
          RunTab{s_tab, s_run}
#endif
          );""")
k = k.replace('  if (TABLE && wave_s <= 3) {\n    // sphere j belongs to slot j mod 7: slots 0-1 -> wave 1, 2-3 -> wave 2, 4-5 -> wave 3, slot 6 -> the list wave,', '#ifdef EXP_TABSTAMP\n  if (tab_wave) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); T1c = clock64(); }\n#endif\n  if (TABLE && wave_s <= 3) {')
k = k.replace('((k + 1) * kWave) / ntab);', """
#ifdef EXP_TABROWS
                          (k * kWave) / ntab + EXP_TABROWS);
#else
                          ((k + 1) * kWave) / ntab);
#endif
""")
k = k.replace('                          (wave_s & 3) - 1 + (wave_s >= 12 ? 3 : 0), 6);\n    init_zbuf(rh * (W + kRowPad));', """                          (wave_s & 3) - 1 + (wave_s >= 12 ? 3 : 0), 6);
#ifdef EXP_TABPRIO
    __builtin_amdgcn_s_setprio(0);
#endif
#ifndef EXP_TABNOINIT
    init_zbuf(rh * (W + kRowPad));
#endif""")
k = k.replace('  if (tab_wave) {\n#ifdef EXP_TABSTAMP', '  if (tab_wave) {\n#ifdef EXP_TABPRIO\n    __builtin_amdgcn_s_setprio(3);\n#endif\n#ifdef EXP_TABSTAMP')
tail = '  if (PERSIST && n + crop_step < N) __syncthreads();'
assert k.count(tail) == 1
k = k.replace(tail, '''  const long long T6 = clock64();
  if (lane == 0) {
    long long *t = tbuf + ((size_t)(blockIdx.x * gridDim.y + blockIdx.y) * 16 + wave) * 8;
    
#ifdef SHR_EXP_WALKCLK
    { long long *q = tbuf + (size_t)gridDim.x * gridDim.y * 129 + ((size_t)(blockIdx.x * gridDim.y + blockIdx.y) * 16 + wave) * 4; q[0] = wclk[0]; q[1] = wclk[1]; q[2] = wclk[2]; q[3] = wclk[3]; }
#endif
    t[0] = T0; t[1] = T1; t[2] = T2; t[3] = T3; t[4] = T4; t[5] = T1c ? T1c : T1b; t[6] = T6; t[7] = T3b; if (wave == 0) tbuf[(size_t)gridDim.x * gridDim.y * 128 + blockIdx.x] = 0;
  }
''' + tail)
exp = '''// timestamped copy of sphere_zbuf_fwd_kernel (generated by tools/gen_exp_ztime.py)
#include <stdlib.h>
#include "../spherehand_amd/csrc/sphere_zbuf.h"
#ifndef EXP_BOX
#define EXP_BOX false
#endif
#ifdef EXP_NOFIX
#define EXP_SQRT __builtin_amdgcn_sqrtf
#else
#define EXP_SQRT sqrt_rn
#endif
namespace shr {
''' + k + '''
}
extern "C" int exp_zfwd_t_launch(const float *spheres, int N, int J, int H, int W, float *depth, unsigned char *argmin,
                               int rows, int shares, long long *tbuf, void *stream) {
  using namespace shr;
  const int zcells = getenv("ZCELLS") ? atoi(getenv("ZCELLS")) : rows * max_box_pitch(W);
#ifdef EXP_TABLE
  constexpr bool TAB = true;
#else
  constexpr bool TAB = false;
#endif
  const size_t lds = kHdrBytes + (size_t)zcells * 8 + (TAB ? (size_t)J * 512 : 0);
  dim3 grid(N, (H + rows - 1) / rows), block(1024);
  int sh = 0; while ((1 << sh) < W / 4) sh++;
  hipFuncSetAttribute((const void *)exp_zfwd_t<true, true, true, false, EXP_BOX, TAB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL((exp_zfwd_t<true, true, true, false, EXP_BOX, TAB>), grid, block, lds, (hipStream_t)stream, (const float4 *)spheres, N, J, H, W, depth, argmin, rows, (sh & 0xff) | ((getenv("FLAGS") ? atoi(getenv("FLAGS")) : 0) << 8) | (16 << 16) | ((getenv("STREAM") ? atoi(getenv("STREAM")) : 0) << 24), shares, zcells, make_axis_k(W, H), tbuf);
  return (int)hipGetLastError();
}
'''
open('tools/exp_ztime.hip', 'w').write(exp)
