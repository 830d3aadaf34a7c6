// What does ds_read_b64_tr_b8 return?  LDS holds bytes with value = (row << 4) | col for a [16 rows][16 cols] block per 16-lane group (row pitch 16 B);
// every lane supplies an address and gets 8 bytes back.  Printed per lane for two addressing hypotheses.  hipcc --offload-arch=gfx950 tools/tr8_probe.hip -o build/tr8_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void probe(uint64_t* out, int pattern) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[4 * 256 * 4];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned char)(i & 0xff);   // byte value = offset within a 256-B block: (row << 4) | col at pitch 16
  __syncthreads();
  const int lane = threadIdx.x, grp = lane >> 4, t = lane & 15;
  int off;
  if (pattern == 0) off = (t >> 1) * 16 + (t & 1) * 8;        // lane -> row t / 2, 8-byte half t & 1 (8 rows x 16 B)
  else if (pattern == 1) off = (t & 7) * 16 + (t >> 3) * 8;   // lane -> row t % 8, half t / 8
  else off = t * 16;                                          // lane -> row t, first 8 bytes (16 rows x 8 B)
  const unsigned addr = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)lds) + grp * 256 + off;
  uint64_t v;
  asm volatile("ds_read_b64_tr_b8 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
  out[lane] = v;
}
int main() {
  uint64_t* d; hipMalloc(&d, 64 * 8);
  uint64_t h[64];
  for (int p = 0; p < 3; ++p) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, p);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("pattern %d (bytes shown as row.col of the 16-lane group's block)\n", p);
    for (int l = 0; l < 20; ++l) {
      printf("  lane %2d:", l);
      for (int b = 0; b < 8; ++b) { unsigned v = (h[l] >> (8 * b)) & 0xff; printf(" %x.%x", v >> 4, v & 15); }
      printf("\n");
    }
  }
  return 0;
}
