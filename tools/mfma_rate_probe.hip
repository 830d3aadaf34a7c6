// Dev probe: issue rate of the two MFMA shapes the GEMM family could use on gfx950 — bf16 16x16x32 (today) and fp8 (e4m3) 16x16x128
// through the f8f6f4 scaled instruction (configs[4] of BASELINE.json) — with 1 and 2 waves per SIMD, no memory traffic.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_rate_probe.hip -o build/mfma_rate_probe && build/mfma_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf8_t __attribute__((ext_vector_type(8)));
typedef int i8_t __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int FP8>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  f4_t acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f4_t{0.f, 0.f, 0.f, 0.f};
  bf8_t a, b;
  i8_t a8, b8;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); a8[i] = 0x38383838 + threadIdx.x; b8[i] = 0x3c3c3c3c; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (FP8) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, acc[i], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int FP8>
int run(int threads, const char* tag) {
  float* out; CK(hipMalloc(&out, 1024 * 512 * 4));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000, blocks = 256;
  k<FP8><<<blocks, threads>>>(out, 100);
  hipEventRecord(e0);
  k<FP8><<<blocks, threads>>>(out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  CK(hipGetLastError());
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = 2.0 * 16 * 16 * (FP8 ? 128 : 32) * 8.0 * iters * (threads / 64) * blocks;
  printf("%-28s %d waves/CU: %8.3f ms  %8.1f TFLOP/s  (%.1f clk per MFMA per SIMD at 2.4 GHz)\n", tag, threads / 64, ms, flops / ms / 1e9,
         ms * 1e-3 * 2.4e9 / (8.0 * iters * (threads / 64) / 4));
  hipFree(out);
  return 0;
}
typedef float f16_t __attribute__((ext_vector_type(16)));
// bf16 32x32x16 (round 3: the shape with half the MFMA instructions per FLOP), 8 independent accumulators like the others
__global__ __launch_bounds__(512) void k32(float* out, int iters) {
  f16_t acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  bf8_t a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int run32(int threads) {
  float* out; CK(hipMalloc(&out, 1024 * 512 * 4));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 10000, blocks = 256;
  k32<<<blocks, threads>>>(out, 100);
  hipEventRecord(e0);
  k32<<<blocks, threads>>>(out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  CK(hipGetLastError());
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = 2.0 * 32 * 32 * 16 * 8.0 * iters * (threads / 64) * blocks;
  printf("%-28s %d waves/CU: %8.3f ms  %8.1f TFLOP/s  (%.1f clk per MFMA per SIMD at 2.4 GHz)\n", "bf16 32x32x16", threads / 64, ms, flops / ms / 1e9,
         ms * 1e-3 * 2.4e9 / (8.0 * iters * (threads / 64) / 4));
  hipFree(out);
  return 0;
}
int main() {
  run<0>(256, "bf16 16x16x32"); run<0>(512, "bf16 16x16x32");
  run32(256); run32(512);
  run<1>(256, "fp8 f8f6f4 16x16x128"); run<1>(512, "fp8 f8f6f4 16x16x128");
  return 0;
}
