// Flat-buffer optimizer / EMA kernels: every parameter of CRNN + Predictor (+ discriminator) lives in one
// contiguous fp32 arena, so one launch updates the whole model and one RCCL all-reduce moves all gradients.
//
//   adam_kernel          torch.optim.Adam(lr, betas, eps, weight_decay) step   [src/main_baseline.py:861-867]
//   sgd_nesterov_kernel  torch.optim.SGD(momentum, nesterov=True, weight_decay) [src/main_scmt_ada_weak.py:854-866]
//   ema_kernel           update_ema_variables: ema = ema*alpha + p*(1-alpha)    [src/main_baseline.py:91-105]
#include "bsed_common.h"
#include "../../include/bsed.h"
#include <algorithm>

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long n, float lr, float b1, float b2, float eps, float wd,
                            int step, const int* __restrict__ step_add, float gscale) {
  // bias corrections from the step count ON THE DEVICE (step + *step_add): a captured launch replays with the count of
  // the step it is replayed for; eager and replayed steps run the same arithmetic, hence the same bits
  const float st = (float)(step + (step_add ? *step_add : 0));
  const float bc1 = 1.f - powf(b1, st), bc2_sqrt = sqrtf(1.f - powf(b2, st));
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float gi = g[i] * gscale;
    const float pi = p[i];
    if (wd != 0.f) gi = fmaf(wd, pi, gi);
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pi - (lr / bc1) * (mi / denom);
  }
}

__global__ void sgd_nesterov_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, long n,
                                    float lr, float momentum, float wd, int first, int nesterov, float gscale) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float gi = g[i] * gscale;
    const float pi = p[i];
    if (wd != 0.f) gi = fmaf(wd, pi, gi);
    if (momentum != 0.f) {
      const float bi = first ? gi : momentum * buf[i] + gi;
      buf[i] = bi;
      gi = nesterov ? gi + momentum * bi : bi;
    }
    p[i] = pi - lr * gi;
  }
}

__global__ void ema_kernel(float* __restrict__ ema, const float* __restrict__ p, long n, float alpha) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    ema[i] = ema[i] * alpha + p[i] * (1.f - alpha);
}

// int64 state entries (BatchNorm num_batches_tracked): float math then truncation, as load_state_dict does
__global__ void ema_i64_kernel(long long* __restrict__ ema, const long long* __restrict__ p, int n, float alpha) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) ema[i] = (long long)((float)ema[i] * alpha + (float)p[i] * (1.f - alpha));
}

__global__ void scale_kernel(float* __restrict__ x, long n, float s) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) x[i] *= s;
}

__global__ void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, long n, float a) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = fmaf(a, x[i], y[i]);
}

__global__ void roll_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int H, int W,
                            const int* __restrict__ sh, const int* __restrict__ sw) {
  const long per = (long)H * W, total = per * B;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i / per);
    const long r = i - (long)b * per;
    const int h = (int)(r / W), w = (int)(r % W);
    int hs = sh ? (h - sh[b]) % H : h, ws = sw ? (w - sw[b]) % W : w;
    if (hs < 0) hs += H;
    if (ws < 0) ws += W;
    out[i] = in[(long)b * per + (long)hs * W + ws];
  }
}

static inline unsigned flat_grid(long n) { return (unsigned)std::min<long>(std::max<long>(ceil_div(n, 256), 1), 4096); }

extern "C" int bsed_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2,
                              float eps, float weight_decay, long step, float grad_scale, void* stream) {
  BSED_CHECK_ARG(p && g && m && v && n > 0 && step >= 1, "bsed_adam_step: bad argument");
  hipLaunchKernelGGL(adam_kernel, dim3(flat_grid(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1,
                     beta2, eps, weight_decay, (int)step, bsed_step_add_ptr(), grad_scale);
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}

extern "C" int bsed_sgd_step(float* p, const float* g, float* buf, long n, float lr, float momentum, float weight_decay,
                             int first_step, int nesterov, float grad_scale, void* stream) {
  BSED_CHECK_ARG(p && g && buf && n > 0, "bsed_sgd_step: bad argument");
  hipLaunchKernelGGL(sgd_nesterov_kernel, dim3(flat_grid(n)), dim3(256), 0, (hipStream_t)stream, p, g, buf, n, lr,
                     momentum, weight_decay, first_step, nesterov, grad_scale);
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}

extern "C" int bsed_ema_update(float* ema, const float* p, long n, float alpha, void* stream) {
  BSED_CHECK_ARG(ema && p && n > 0, "bsed_ema_update: bad argument");
  hipLaunchKernelGGL(ema_kernel, dim3(flat_grid(n)), dim3(256), 0, (hipStream_t)stream, ema, p, n, alpha);
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}

extern "C" int bsed_ema_update_i64(long long* ema, const long long* p, int n, float alpha, void* stream) {
  BSED_CHECK_ARG(ema && p && n > 0, "bsed_ema_update_i64: bad argument");
  hipLaunchKernelGGL(ema_i64_kernel, dim3(ceil_div(n, 64)), dim3(64), 0, (hipStream_t)stream, ema, p, n, alpha);
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}

extern "C" int bsed_axpy(float* y, const float* x, long n, float a, void* stream) {
  BSED_CHECK_ARG(y && x && n > 0, "bsed_axpy: bad argument");
  hipLaunchKernelGGL(axpy_kernel, dim3(flat_grid(n)), dim3(256), 0, (hipStream_t)stream, y, x, n, a);
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}

extern "C" int bsed_roll(const float* in, float* out, int B, int H, int W, const int* sh, const int* sw, void* stream) {
  BSED_CHECK_ARG(in && out && in != out && B > 0 && H > 0 && W > 0, "bsed_roll: bad argument");
  hipLaunchKernelGGL(roll_kernel, dim3(flat_grid((long)B * H * W)), dim3(256), 0, (hipStream_t)stream, in, out, B, H, W, sh, sw);
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}

extern "C" int bsed_scale(float* x, long n, float s, void* stream) {
  BSED_CHECK_ARG(x && n > 0, "bsed_scale: bad argument");
  hipLaunchKernelGGL(scale_kernel, dim3(flat_grid(n)), dim3(256), 0, (hipStream_t)stream, x, n, s);
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}
