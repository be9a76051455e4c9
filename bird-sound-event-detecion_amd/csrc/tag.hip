// CNN-only tagging head (BASELINE configs[1]): what the reference's CRNN_pred.forward does after the CNN stack
// (/root/reference/src/models/CRNN_GRL.py:252-290):
//     strong = sigmoid(x)                                   x: (B,T',C) features of the last CNN block, C = 128
//     sof    = clamp(softmax_class(x Ws^T + bs), 1e-7, 1)   dense_softmax: Linear(2*n_RNN_cell = C, nclass = C)
//     weak   = sum_t strong*sof / sum_t sof
// The C x C logits come from the 1-tap contraction kernels (igemm3 / igemm); this file holds the per-frame softmax
// over the class axis and the attention pooling over time: one wave per frame (lane = classes lane, lane+64), the
// frames of a clip split over S workgroups whose numerator / denominator partials are summed in fixed order.
#include "bsed_common.h"
#include "../../include/bsed.h"

#define TAG_C 128
#define TAG_WAVES 4

__global__ __launch_bounds__(64 * TAG_WAVES) void tag_pool_kernel(
    const float* __restrict__ x, const float* __restrict__ logits, float* __restrict__ strong,
    float* __restrict__ part /*(B,S,2,C)*/, int T) {
  __shared__ float red[TAG_WAVES][2][TAG_C];
  const int b = blockIdx.y, s = blockIdx.x, S = gridDim.x;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int per = (T + S - 1) / S;
  const int f0 = s * per, f1 = min(T, f0 + per);
  float num0 = 0.f, num1 = 0.f, den0 = 0.f, den1 = 0.f;
  for (int f = f0 + wv; f < f1; f += TAG_WAVES) {
    const size_t o = ((size_t)b * T + f) * TAG_C;
    const float l0 = logits[o + lane], l1 = logits[o + 64 + lane];
    const float x0 = x[o + lane], x1 = x[o + 64 + lane];
    const float m = wave_max(fmaxf(l0, l1));
    const float e0 = expf(l0 - m), e1 = expf(l1 - m);
    const float inv = 1.0f / wave_sum(e0 + e1);
    const float a0 = fminf(fmaxf(e0 * inv, 1e-7f), 1.0f), a1 = fminf(fmaxf(e1 * inv, 1e-7f), 1.0f);
    const float s0 = sigmoidf_(x0), s1 = sigmoidf_(x1);
    strong[o + lane] = s0;
    strong[o + 64 + lane] = s1;
    num0 = fmaf(s0, a0, num0); num1 = fmaf(s1, a1, num1);
    den0 += a0; den1 += a1;
  }
  red[wv][0][lane] = num0; red[wv][0][lane + 64] = num1;
  red[wv][1][lane] = den0; red[wv][1][lane + 64] = den1;
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * TAG_C; i += 64 * TAG_WAVES) {
    const int which = i / TAG_C, c = i % TAG_C;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < TAG_WAVES; ++w) v += red[w][which][c];
    part[(((size_t)b * S + s) * 2 + which) * TAG_C + c] = v;
  }
}

__global__ void tag_weak_kernel(const float* __restrict__ part, float* __restrict__ weak, int B, int S) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * TAG_C) return;
  const int b = i / TAG_C, c = i % TAG_C;
  float num = 0.f, den = 0.f;
  for (int s = 0; s < S; ++s) {
    num += part[(((size_t)b * S + s) * 2 + 0) * TAG_C + c];
    den += part[(((size_t)b * S + s) * 2 + 1) * TAG_C + c];
  }
  weak[i] = num / den;
}

// time splits per clip: ~4 workgroups per CU, at least 8 frames per wave-group
extern "C" int bsed_tag_splits(int B, int T) {
  int S = 1;
  while (S < 64 && (long)B * S < 1024 && (T + 2 * S - 1) / (2 * S) >= 2 * TAG_WAVES) S *= 2;
  return S;
}

extern "C" int bsed_tag_head_fwd(const float* x, const float* logits, float* strong, float* weak, float* part, int B,
                                 int T, int C, void* stream) {
  BSED_CHECK_ARG(x && logits && strong && weak && part, "bsed_tag_head_fwd: null tensor");
  BSED_CHECK_ARG(B > 0 && B <= 65535 && T > 0, "bsed_tag_head_fwd: bad shape");
  BSED_CHECK_ARG(C == TAG_C, "bsed_tag_head_fwd: built for %d feature channels / classes (got %d)", TAG_C, C);
  const int S = bsed_tag_splits(B, T);
  hipLaunchKernelGGL(tag_pool_kernel, dim3(S, B), dim3(64 * TAG_WAVES), 0, (hipStream_t)stream, x, logits, strong, part, T);
  hipLaunchKernelGGL(tag_weak_kernel, dim3((B * TAG_C + 255) / 256), dim3(256), 0, (hipStream_t)stream, part, weak, B, S);
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}
