// Development aid: occupy `grid` CUs' worth of workgroups (256 threads, 512 VGPRs per lane-quad: nothing else fits on the
// SIMD) for about `usec` microseconds - a stand-in for RCCL's kernels holding CUs while a persistent GEMM is launched.
#include <hip/hip_runtime.h>
extern "C" __global__ __launch_bounds__(256, 1) void hog_kernel(long cycles, float* out) {
  float acc[200];
#pragma unroll
  for (int i = 0; i < 200; ++i) acc[i] = threadIdx.x * 0.001f + i;
  const long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) {
#pragma unroll
    for (int i = 0; i < 200; ++i) acc[i] = acc[i] * 1.0001f + 0.5f;
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 200; ++i) s += acc[i];
  if (s == 12345.678f) out[0] = s;
}
extern "C" int hog_launch(int grid, int usec, float* out, void* stream) {
  // wall_clock64 ticks at 100 MHz on this part
  hipLaunchKernelGGL(hog_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (long)usec * 100, out);
  return (int)hipGetLastError();
}
