"""Generate tools/exp_mesh.hip: mesh_depth_kernel with per-phase clock64 accumulators."""
src = open('spherehand_amd/csrc/mesh_depth.hip').read()
a = src.index('template <int TO, int SL, bool EXACT>\n__global__')
b = src.index('}  // namespace shr')
k = src[a:b]
k = k.replace('mesh_depth_kernel(', 'exp_mesh(')
k = k.replace('float *__restrict__ depth) {', 'float *__restrict__ depth, long long *tb) {\n  long long tA = 0, tS = 0, tB = 0, tW = 0; int nitems = 0; const long long T0 = clock64();', 1)
k = k.replace('    int nk[kMeshFaces], dx0[kMeshFaces];', '    const long long a0 = clock64();\n    int nk[kMeshFaces], dx0[kMeshFaces];')
k = k.replace('    int incl = n;   // inclusive scan', '    const long long a1 = clock64(); tA += a1 - a0;\n    int incl = n;   // inclusive scan')
k = k.replace('    for (int w0 = 0; w0 < total; w0 += kMeshQueue) {', '    tS += clock64() - a1; nitems += total;\n    for (int w0 = 0; w0 < total; w0 += kMeshQueue) {')
k = k.replace('      const int count = min(kMeshQueue, total - w0);', '      const long long b0 = clock64();\n      const int count = min(kMeshQueue, total - w0);')
k = k.replace('''                if (zp == zp) atomicMin(&s_z[SL * (dy - ty0) + sy][SL * (dx - tx0) + sx], mkey(zp));
              }
            }
          }
        }
      }
    }
  }
  __syncthreads();''', '''                if (zp == zp) atomicMin(&s_z[SL * (dy - ty0) + sy][SL * (dx - tx0) + sx], mkey(zp));
              }
            }
          }
        }
      }
      const long long b1 = clock64(); tB += b1 - b0;
    }
  }
  const long long e0 = clock64();
  __syncthreads();''')
idx = k.rstrip().rfind('}')
k = k[:idx] + '''  if (lane == 0) {
    long long *t = tb + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16 + wave) * 8;
    t[0] = tA; t[1] = tS; t[2] = tB; t[3] = e0 - T0; t[4] = clock64() - T0; t[5] = nitems;
  }
}
'''
exp = '#include "../spherehand_amd/csrc/common.h"\nnamespace shr {\n' + src[src.index('__device__ __forceinline__ uint32_t mkey'):a] + k + '''}
extern "C" int exp_mesh_launch(const float *vertices, const int *faces, int B, int NV, int F, int src, int S, float *depth,
                               long long *tb, void *stream) {
  using namespace shr;
  if (S == 128) hipLaunchKernelGGL((exp_mesh<128, 1, true>), dim3(1, B), dim3(1024), 0, (hipStream_t)stream, (const float4 *)vertices, faces, NV, F, src, S, 100.f, depth, tb);
  else hipLaunchKernelGGL((exp_mesh<64, 2, true>), dim3(((S + 63) / 64) * ((S + 63) / 64), B), dim3(1024), 0, (hipStream_t)stream, (const float4 *)vertices, faces, NV, F, src, S, 100.f, depth, tb);
  return (int)hipGetLastError();
}
'''
open('tools/exp_mesh.hip', 'w').write(exp)
