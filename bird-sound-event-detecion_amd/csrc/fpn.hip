// Feature-pyramid glue of CRNN_fpn (reference src/models/CRNN_GRL.py:333-336,378-384): nn.Upsample((T_out, 1),
// mode='bilinear', align_corners=True) on width-1 maps is a linear interpolation along time,
//   out[t] = (1 - w) * in[i0] + w * in[i0 + 1],   pos = t * (T_in - 1) / (T_out - 1), i0 = floor(pos), w = pos - i0,
// written straight into the right half of the concatenation buffer (out_pitch / out_offset), and its adjoint as a
// deterministic gather (every input frame sums the few output frames whose support contains it).
#include "bsed_common.h"
#include "../../include/bsed.h"
#include <algorithm>

__device__ __forceinline__ void up_coeff(int t, int T_in, int T_out, int& i0, float& w) {
  // PyTorch computes the source index in fp32: scale = (T_in - 1) / (T_out - 1), pos = scale * t
  const float scale = T_out > 1 ? (float)(T_in - 1) / (float)(T_out - 1) : 0.f;
  const float pos = scale * (float)t;
  i0 = min((int)pos, T_in - 1);
  w = pos - (float)i0;
}

__global__ void upsample_time_fwd_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int T_in,
                                         int T_out, int C, int in_pitch, int out_pitch) {
  const long total = (long)B * T_out * (C / 4);
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(e % (C / 4));
    const long r = e / (C / 4);
    const int t = (int)(r % T_out), b = (int)(r / T_out);
    int i0; float w;
    up_coeff(t, T_in, T_out, i0, w);
    const int i1 = min(i0 + 1, T_in - 1);
    const float4 a = *reinterpret_cast<const float4*>(in + ((size_t)b * T_in + i0) * in_pitch + 4 * c4);
    const float4 c = *reinterpret_cast<const float4*>(in + ((size_t)b * T_in + i1) * in_pitch + 4 * c4);
    const float w0 = 1.0f - w;
    *reinterpret_cast<float4*>(out + ((size_t)b * T_out + t) * out_pitch + 4 * c4) =
        make_float4(w0 * a.x + w * c.x, w0 * a.y + w * c.y, w0 * a.z + w * c.z, w0 * a.w + w * c.w);
  }
}

// d_in[b][i][c] = sum_t d_out[b][t][c] * ((i0(t) == i) * (1 - w(t)) + (i1(t) == i) * w(t))
__global__ void upsample_time_bwd_kernel(const float* __restrict__ dout, float* __restrict__ din, int B, int T_in,
                                         int T_out, int C, int dout_pitch, int din_pitch) {
  const long total = (long)B * T_in * (C / 4);
  const float inv = T_in > 1 ? (float)(T_out - 1) / (float)(T_in - 1) : 0.f;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(e % (C / 4));
    const long r = e / (C / 4);
    const int i = (int)(r % T_in), b = (int)(r / T_in);
    // candidates: output frames whose source position lies in (i - 1, i + 1); two frames of slack for fp32 rounding
    const int t_lo = max(0, (int)floorf((float)(i - 1) * inv) - 1);
    const int t_hi = min(T_out - 1, (int)ceilf((float)(i + 1) * inv) + 1);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = t_lo; t <= t_hi; ++t) {
      int i0; float w;
      up_coeff(t, T_in, T_out, i0, w);
      const int i1 = min(i0 + 1, T_in - 1);
      const float k = (i0 == i ? 1.0f - w : 0.f) + (i1 == i ? w : 0.f);
      if (k != 0.f) {
        const float4 g = *reinterpret_cast<const float4*>(dout + ((size_t)b * T_out + t) * dout_pitch + 4 * c4);
        s.x = fmaf(k, g.x, s.x); s.y = fmaf(k, g.y, s.y); s.z = fmaf(k, g.z, s.z); s.w = fmaf(k, g.w, s.w);
      }
    }
    *reinterpret_cast<float4*>(din + ((size_t)b * T_in + i) * din_pitch + 4 * c4) = s;
  }
}

extern "C" int bsed_upsample_time_fwd(const float* in, float* out, int B, int T_in, int T_out, int C, int in_pitch,
                                      int out_pitch, void* stream) {
  BSED_CHECK_ARG(in && out && B > 0 && T_in > 0 && T_out > 0 && C > 0 && C % 4 == 0 && in_pitch >= C &&
                     out_pitch >= C && in_pitch % 4 == 0 && out_pitch % 4 == 0, "bsed_upsample_time_fwd: bad argument");
  const long total = (long)B * T_out * (C / 4);
  hipLaunchKernelGGL(upsample_time_fwd_kernel, dim3((unsigned)std::min<long>(ceil_div(total, 256), 8192)), dim3(256), 0,
                     (hipStream_t)stream, in, out, B, T_in, T_out, C, in_pitch, out_pitch);
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}

extern "C" int bsed_upsample_time_bwd(const float* dout, float* din, int B, int T_in, int T_out, int C, int dout_pitch,
                                      int din_pitch, void* stream) {
  BSED_CHECK_ARG(dout && din && B > 0 && T_in > 0 && T_out > 0 && C > 0 && C % 4 == 0 && dout_pitch >= C &&
                     din_pitch >= C && dout_pitch % 4 == 0 && din_pitch % 4 == 0, "bsed_upsample_time_bwd: bad argument");
  const long total = (long)B * T_in * (C / 4);
  hipLaunchKernelGGL(upsample_time_bwd_kernel, dim3((unsigned)std::min<long>(ceil_div(total, 256), 8192)), dim3(256), 0,
                     (hipStream_t)stream, dout, din, B, T_in, T_out, C, dout_pitch, din_pitch);
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}
