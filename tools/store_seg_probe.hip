// Per-CU store path: one 512-thread workgroup per CU stores its 256 x 256 bf16 tile (128 KiB) of a row-major [M][N] matrix with 16-byte
// stores whose 64 lanes cover  (a) 8 rows x 128 B  (the wave-private epilogue strip of gemm.hip)  (b) 4 rows x 256 B  (c) 2 rows x 512 B
// (whole tile rows: would need a cross-wave strip)  (d) 1 KiB contiguous (not a tile shape)  (e) 16 rows x four 16-B chunks at a 32-B stride, the other
// chunks by a second instruction (accumulator fragments stored as they lie when the weight rows are permuted so that a lane owns 16 contiguous columns).  Clocks of workgroup 0 and kernel time.
// build: hipcc --offload-arch=gfx950 -O2 tools/store_seg_probe.hip -o build/store_seg_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ unsigned long long g_clk[5];
template <int SEG>   // bytes of a row segment covered by consecutive lanes
__global__ __launch_bounds__(512) void k(char* C, long long ld_bytes, int tiles_n, int which) {
  const int tile = blockIdx.x, tm = tile / tiles_n, tn = tile % tiles_n;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int LPR = SEG >= 1024 ? 64 : SEG / 16, RPI = 64 / LPR;           // lanes per row segment, rows per instruction
  char* base = C + (long long)tm * 256 * ld_bytes + tn * 512;
  uint4 v = make_uint4(tile, w, lane, 1);
  const unsigned long long t0 = __builtin_readcyclecounter();
  if constexpr (SEG <= 512) {
    // wave w owns column block (w % (512 / SEG)) and a row range; 128 KiB / 8 waves = 16 KiB per wave = 16 instructions
    constexpr int CB = 512 / SEG, RW = 256 / (8 / CB);    // column blocks, rows per wave
    const int cb = w % CB, r0 = (w / CB) * RW;
#pragma unroll 4
    for (int i = 0; i < RW / RPI; ++i) {
      const int row = r0 + i * RPI + lane / LPR;
      *reinterpret_cast<uint4*>(base + (long long)row * ld_bytes + cb * SEG + (lane % LPR) * 16) = v;
    }
  } else if constexpr (SEG == 2048) {
    // wave w: rows 128 (w / 4) .., columns 64 (w % 4) ..; fragment i = 16 rows; lane (t, g): row t, bytes 32 g + {0, 16}
    const int t = lane & 15, g = lane >> 4;
    char* wb = base + (long long)((w / 4) * 128 + t) * ld_bytes + (w % 4) * 128 + g * 32;
#pragma unroll 4
    for (int i = 0; i < 8; ++i) {
      *reinterpret_cast<uint4*>(wb + (long long)i * 16 * ld_bytes) = v;
      *reinterpret_cast<uint4*>(wb + (long long)i * 16 * ld_bytes + 16) = v;
    }
  } else {
    char* lin = C + (long long)tile * 131072 + w * 16384;
#pragma unroll 4
    for (int i = 0; i < 16; ++i) *reinterpret_cast<uint4*>(lin + i * 1024 + lane * 16) = v;
  }
  __builtin_amdgcn_s_waitcnt(0);
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (blockIdx.x == 100 && threadIdx.x == 0) g_clk[which] = t1 - t0;
}
int main() {
  const int M = 50432, N = 1536;   // decoder qkv output
  char* d; hipMalloc(&d, (size_t)M * N * 2 + (1 << 20));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int tiles_n = N / 256, tiles = 256 * 4;   // four rounds of the chip
  for (int rep = 0; rep < 2; ++rep)
    for (int which = 0; which < 5; ++which) {
      hipEventRecord(e0);
      for (int i = 0; i < 10; ++i) {
        if (which == 0) k<128><<<tiles, 512>>>(d, (long long)N * 2, tiles_n, which);
        else if (which == 1) k<256><<<tiles, 512>>>(d, (long long)N * 2, tiles_n, which);
        else if (which == 2) k<512><<<tiles, 512>>>(d, (long long)N * 2, tiles_n, which);
        else if (which == 3) k<1024><<<tiles, 512>>>(d, (long long)N * 2, tiles_n, which);
        else k<2048><<<tiles, 512>>>(d, (long long)N * 2, tiles_n, which);
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long c[5]; hipMemcpyFromSymbol(c, HIP_SYMBOL(g_clk), sizeof(c));
      const char* nm[] = {"8 rows x 128 B", "4 rows x 256 B", "2 rows x 512 B", "1 KiB contiguous", "16 rows x 4 x 16 B"};
      printf("%-18s per instruction: workgroup %6llu clocks for 128 KiB (%.1f B/clk/CU), kernel %.1f us = %.2f TB/s\n", nm[which], c[which], 131072.0 / c[which],
             ms * 100, tiles * 131072.0 / (ms / 10 * 1e-3) / 1e12);
    }
  return 0;
}
