// MFMA fragment-layout self test: C(32x32) = A(32xK) * B(Kx32) with v_mfma_f32_32x32x2_f32.
// Pins the lane maps every MFMA kernel in this library relies on (checked on the GPU by
// tests/test_igemm_gpu.py with asymmetric operands).
#include "bsed_common.h"
#include "../../include/bsed.h"

__global__ void mfma_selftest_kernel(const float* A, const float* B, float* C, int K) {
  const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  f32x16 acc = {0};
  for (int k = 0; k < K; k += 2) {
    const float a = A[li * K + k + lh];         // A[i = lane&31][k = lane>>5]
    const float b = B[(k + lh) * 32 + li];      // B[k = lane>>5][j = lane&31]
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;  // C row of register r
    C[row * 32 + li] = acc[r];
  }
}

extern "C" int bsed_selftest_mfma(const float* A, const float* B, float* C, int K, void* stream) {
  BSED_CHECK_ARG(A && B && C && K > 0 && K % 2 == 0, "bsed_selftest_mfma: bad argument");
  hipLaunchKernelGGL(mfma_selftest_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, A, B, C, K);
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}
