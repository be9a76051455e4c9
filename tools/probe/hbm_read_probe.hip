// What a tuned READ-ONLY streaming kernel gets from HBM on this box (context for the read-only weight-gradient kernels).
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe/hbm_read_probe tools/probe/hbm_read_probe.hip
// Variants: U float4 loads in flight per lane (register destinations), W workgroups per CU, 256 threads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int U>
__global__ __launch_bounds__(256) void read_kernel(const float4* __restrict__ x, size_t n4, float* __restrict__ out) {
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const size_t stride = (size_t)gridDim.x * 256 * U;
  for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i + (U - 1) * 256 < n4; i += stride) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = x[i + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
  }
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = 1.f;
}

template <int U>
__global__ __launch_bounds__(256) void copy_kernel(const float4* __restrict__ x, float4* __restrict__ y, size_t n4) {
  const size_t stride = (size_t)gridDim.x * 256 * U;
  for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i + (U - 1) * 256 < n4; i += stride) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = x[i + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) y[i + u * 256] = v[u];
  }
}

template <int U>
void run(const float4* x, float4* y, size_t n4, float* out, int wpc) {
  hipEvent_t s, e;
  hipEventCreate(&s); hipEventCreate(&e);
  const int grid = 256 * wpc;
  for (int mode = 0; mode < 2; ++mode) {
    for (int w = 0; w < 2; ++w) {
      if (mode == 0) hipLaunchKernelGGL(read_kernel<U>, dim3(grid), dim3(256), 0, 0, x, n4, out);
      else hipLaunchKernelGGL(copy_kernel<U>, dim3(grid), dim3(256), 0, 0, x, y, n4);
    }
    hipEventRecord(s);
    const int reps = 5;
    for (int r = 0; r < reps; ++r) {
      if (mode == 0) hipLaunchKernelGGL(read_kernel<U>, dim3(grid), dim3(256), 0, 0, x, n4, out);
      else hipLaunchKernelGGL(copy_kernel<U>, dim3(grid), dim3(256), 0, 0, x, y, n4);
    }
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e); ms /= reps;
    const double bytes = (double)n4 * 16 * (mode == 0 ? 1 : 2);
    printf("%s U=%2d wg/CU=%d : %.3f ms  %.2f TB/s\n", mode == 0 ? "read" : "copy", U, wpc, ms, bytes / ms / 1e9);
  }
}

int main() {
  const size_t n4 = (size_t)1 << 27;   // 2 GiB
  float4 *x, *y; float* out;
  hipMalloc(&x, n4 * 16); hipMalloc(&y, n4 * 16); hipMalloc(&out, 4);
  hipMemset(x, 0x3c, n4 * 16); hipMemset(y, 0, n4 * 16);
  for (int wpc : {2, 4, 8}) {
    run<2>(x, y, n4, out, wpc);
    run<4>(x, y, n4, out, wpc);
    run<8>(x, y, n4, out, wpc);
    run<16>(x, y, n4, out, wpc);
  }
  return 0;
}
