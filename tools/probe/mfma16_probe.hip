// cycles per MFMA (one wave, back-to-back independent / dependent chains) for the shapes block 0 could use
//   hipcc --offload-arch=gfx950 -O3 tools/probe/mfma16_probe.hip -o /tmp/mfma16_probe && /tmp/mfma16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int DEP>
__global__ void probe(unsigned long long* out, float* sink) {
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  const float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  s16x4 as = {(short)threadIdx.x, 1, 2, 3}, bs = {4, 5, 6, (short)threadIdx.x};
  bf16x8 a8, b8;
  for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(a + i); b8[i] = (__bf16)(b - i); }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < 256; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      f32x4& c = acc[DEP ? 0 : u];
      if (KIND == 0) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
      else if (KIND == 1) c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(as, bs, c, 0, 0, 0);
      else c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, c, 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[0] = t1 - t0;
  sink[threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}

int main() {
  unsigned long long* d; float* s;
  hipMalloc(&d, 8); hipMalloc(&s, 64 * 4);
  const char* names[3] = {"v_mfma_f32_16x16x4_f32", "v_mfma_f32_16x16x16_bf16 (1k)", "v_mfma_f32_16x16x32_bf16"};
  for (int k = 0; k < 3; ++k)
    for (int dep = 0; dep < 2; ++dep) {
      unsigned long long h = 0;
      for (int rep = 0; rep < 2; ++rep) {
        if (k == 0 && dep == 0) hipLaunchKernelGGL((probe<0, 0>), 1, 64, 0, 0, d, s);
        if (k == 0 && dep == 1) hipLaunchKernelGGL((probe<0, 1>), 1, 64, 0, 0, d, s);
        if (k == 1 && dep == 0) hipLaunchKernelGGL((probe<1, 0>), 1, 64, 0, 0, d, s);
        if (k == 1 && dep == 1) hipLaunchKernelGGL((probe<1, 1>), 1, 64, 0, 0, d, s);
        if (k == 2 && dep == 0) hipLaunchKernelGGL((probe<2, 0>), 1, 64, 0, 0, d, s);
        if (k == 2 && dep == 1) hipLaunchKernelGGL((probe<2, 1>), 1, 64, 0, 0, d, s);
        hipDeviceSynchronize();
        hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
      }
      printf("%-34s %s chain: %.1f cycles per MFMA\n", names[k], dep ? "dependent  " : "independent", h / 1024.0);
    }
  return 0;
}
