// Dev probe: L2 -> LDS staging rate of `buffer_load_dwordx4 ... lds` as a function of the contiguous row-segment length per K-tile
// (64 B = 32 bf16, 128 B = one cache line, 256 B) and of the row stride.  Same ring / wait / barrier structure as the GEMM main loop,
// no fragment reads and no MFMAs.   hipcc --offload-arch=gfx950 -O3 tools/dma_probe.hip -o /tmp/dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1;} }while(0)
#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

template <int SEG, int STAGES, int STAGE_BYTES>
__global__ __launch_bounds__(512) void k_dma(const char* src, unsigned src_bytes, int ld, int npanels, int kt_total, int* sink) {
  constexpr int PIECES = STAGE_BYTES / 1024, PPW = PIECES / 8, LPR = SEG / 16, RPP = 64 / LPR, ROWS = STAGE_BYTES / SEG;
  __shared__ __attribute__((aligned(16))) char smem[STAGES * STAGE_BYTES];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, src_bytes, 0x00020000);
  const int panel = (blockIdx.x >> 3) % npanels;
  unsigned off[PPW];
#pragma unroll
  for (int q = 0; q < PPW; ++q) {
    int piece = w * PPW + q, row = piece * RPP + lane / LPR;
    off[q] = (unsigned)((panel * ROWS + row) * ld + (lane % LPR) * 16);
  }
  const int segs_per_row = ld / SEG;
  auto stage = [&](int kt) {
    char* base = smem + (kt % STAGES) * STAGE_BYTES;
    unsigned koff = (unsigned)((kt % segs_per_row) * SEG);
#pragma unroll
    for (int q = 0; q < PPW; ++q)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(void, base + (w * PPW + q) * 1024), 16, (int)(off[q] + koff), 0, 0, 0);
  };
  for (int kt = 0; kt < STAGES - 1; ++kt) stage(kt);
  for (int kt = 0; kt < kt_total; ++kt) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * PPW) : "memory");
    __builtin_amdgcn_s_barrier();
    stage(kt + STAGES - 1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0 && smem[0] == 123 && smem[1] == 77) sink[0] = 1;
}

template <int SEG, int STAGES, int STAGE_BYTES>
int run(const char* d, unsigned bytes, int ld, int npanels, int* sink, const char* tag) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int KT = 512, blocks = 1024;
  k_dma<SEG, STAGES, STAGE_BYTES><<<blocks, 512>>>(d, bytes, ld, npanels, KT, sink);
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) k_dma<SEG, STAGES, STAGE_BYTES><<<blocks, 512>>>(d, bytes, ld, npanels, KT, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  CK(hipGetLastError());
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  double tot = (double)blocks * KT * STAGE_BYTES;
  printf("%-10s seg=%3d B stages=%d stage=%2d KB ld=%5d panels=%3d : %7.1f us  %6.2f TB/s  %5.1f B/clk/CU (2.4 GHz)  %6.0f clk per 32 KB\n", tag, SEG, STAGES,
         STAGE_BYTES / 1024, ld, npanels, ms * 1e3, tot / (ms * 1e-3) / 1e12, tot / (ms * 1e-3) / 256 / 2.4e9, 32768.0 / (tot / (ms * 1e-3) / 256 / 2.4e9));
  return 0;
}

int main() {
  const unsigned bytes = 256u << 20;
  char* d; CK(hipMalloc(&d, bytes)); CK(hipMemset(d, 1, bytes));
  int* sink; CK(hipMalloc(&sink, 4));
  for (int ld : {1024, 1536, 2048, 4096, 6144, 4096 + 128}) {
    for (int np : {4, 64}) {
      run<64, 4, 32768>(d, bytes, ld, np, sink, "bk32");
      run<128, 4, 32768>(d, bytes, ld, np, sink, "bk64-half");
      run<128, 2, 65536>(d, bytes, ld, np, sink, "bk64");
      run<256, 4, 32768>(d, bytes, ld, np, sink, "seg256");
    }
  }
  return 0;
}
