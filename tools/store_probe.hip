// Dev probe: HBM write bandwidth for (a) linear 16-B stores, (b) GEMM-epilogue-shaped 128-B row segments at a row stride.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1;} }while(0)
__global__ void k_linear(uint4* p, size_t n16) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  uint4 v = make_uint4(i, 1, 2, 3);
  for (; i < n16; i += st) p[i] = v;
}
// tile (tm,tn): 128 rows x 128 cols bf16 (256 B per row): wave w writes rows 32w..32w+31; lane -> (row = pass*4 + lane/16, col 8B*(lane%16)) per 64-col half
__global__ void k_tile(char* p, int tiles_n, long long ld_bytes, int bytes_per_lane) {
  int tile = blockIdx.x, tm = tile / tiles_n, tn = tile % tiles_n;
  int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  char* base = p + ((long long)tm * 128) * ld_bytes + tn * 256;
  int lpr = 128 / bytes_per_lane;            // lanes per 128-B half-row segment
  int rpp = 64 / lpr;
  for (int half = 0; half < 2; ++half)
    for (int ps = 0; ps < 64 / rpp; ++ps) {   // wave covers 64 rows x 128 B for its (w&1) column half... 4 waves: 2x2
      int row = (w >> 1) * 64 + ps * rpp + lane / lpr;
      char* q = base + (long long)row * ld_bytes + (w & 1) * 128 + (lane % lpr) * bytes_per_lane;
      if (half == 1) break;
      if (bytes_per_lane == 8) *reinterpret_cast<uint2*>(q) = make_uint2(row, lane);
      else *reinterpret_cast<uint4*>(q) = make_uint4(row, lane, 0, 0);
    }
}
int main() {
  const int M = 50432, N = 1536;
  size_t bytes = (size_t)M * N * 2;
  char* d; CK(hipMalloc(&d, bytes));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0); for (int i = 0; i < 10; ++i) k_linear<<<2048, 256>>>((uint4*)d, bytes / 16); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1); printf("linear 16B stores: %.1f us  %.2f TB/s\n", ms * 100, bytes / (ms / 10 * 1e-3) / 1e12);
    for (int bpl = 8; bpl <= 16; bpl += 8) {
      int tiles = (M / 128) * (N / 128);
      hipEventRecord(e0); for (int i = 0; i < 10; ++i) k_tile<<<tiles, 256>>>(d, N / 128, (long long)N * 2, bpl); hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1); printf("tile-shaped %2d B/lane stores: %.1f us  %.2f TB/s\n", bpl, ms * 100, bytes / (ms / 10 * 1e-3) / 1e12);
    }
  }
  return 0;
}
