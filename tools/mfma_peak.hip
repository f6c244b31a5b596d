// Sustained fp32 MFMA rate of the device under DVFS: a register-only loop of v_mfma_f32_32x32x2_f32 on random
// operands (no LDS, no global memory in the loop), 4 independent accumulators per wave, W waves per SIMD.
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
// Prints TFLOP/s and the effective clock (s_memtime / s_memrealtime) for 1 and 2 waves per SIMD.
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wimplicit-const-int-float-conversion"
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void mfma_loop(const float* __restrict__ in, float* __restrict__ out, int iters,
                                                  unsigned long long* __restrict__ clk) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  float a[4], b[4];
  for (int j = 0; j < 4; ++j) { a[j] = in[(tid * 8 + j) & 0xFFFF]; b[j] = in[(tid * 8 + 4 + j) & 0xFFFF]; }
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t)
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int t = 0; t < 4; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[(j + t) & 3], acc[t], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int t = 0; t < 4; ++t)
    for (int r = 0; r < 16; ++r) s += acc[t][r];
  out[tid] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int main() {
  const int iters = 20000;
  std::vector<float> h(65536);
  srand(1);
  for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  float *din, *dout;
  unsigned long long* dclk;
  hipMalloc(&din, h.size() * 4);
  hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  for (int wps = 1; wps <= 2; ++wps) {
    const int blocks = 256 * wps;                       // 256 CUs x (4 waves per block) x wps
    hipMalloc(&dout, (size_t)blocks * 256 * 4);
    hipMalloc(&dclk, (size_t)blocks * 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, din, dout, iters, dclk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int reps = 5;
    for (int rep = 0; rep < reps; ++rep) hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, din, dout, iters, dclk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    std::vector<unsigned long long> c(2 * blocks);
    hipMemcpy(c.data(), dclk, c.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0;
    for (int i = 0; i < blocks; ++i) { cyc += c[2 * i]; rt += c[2 * i + 1]; }
    const double flops = (double)blocks * 4 /*waves*/ * iters * 16 /*mfma*/ * 32.0 * 32 * 2 * 2;
    printf("waves/SIMD %d: %.3f ms  %.1f TFLOP/s  clock %.2f GHz  cycles per MFMA %.1f\n", wps, ms, flops / ms / 1e9,
           cyc / rt * 0.1, cyc / blocks / ((double)iters * 16) / wps * (wps));
    hipFree(dout); hipFree(dclk);
  }
  return 0;
}
