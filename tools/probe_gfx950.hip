// Dev probe (not product): pins down gfx950 fragment layouts empirically.
//  1. ds_read_b64_tr_b16 lane/element mapping
//  2. mfma_f32_16x16x32_bf16 / 32x32x16_bf16 / 16x16x4f32 A,B,C layouts
//  3. buffer_load ... lds out-of-bounds zero fill
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef short s4 __attribute__((ext_vector_type(4)));
typedef short s8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
#define LDSP(T, p) ((__attribute__((address_space(3))) T*)(p))
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

static unsigned short f2bf(float f){ unsigned u; memcpy(&u,&f,4); return (unsigned short)(u>>16); }

__global__ void k_tr(short* out, int mode) {
  __shared__ __attribute__((aligned(16))) short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
  __syncthreads();
  int lane = threadIdx.x;
  int addr;
  if (mode == 0) addr = lane * 4;                       // contiguous 8B chunks
  else { int g = lane >> 4, t = lane & 15; addr = g * 1024 + (t >> 2) * 128 + (t & 3) * 4; } // 4 rows x stride 128 elems
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDSP(s4, lds + addr));
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = v[j];
}

// C[16x16] = A[16x32] * B[32x16]; A row-major [i][k], Bt row-major [j][k]
__global__ void k_mfma16(const unsigned short* A, const unsigned short* Bt, float* C) {
  int l = threadIdx.x;
  s8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = A[(l & 15) * 32 + (l >> 4) * 8 + j]; b[j] = Bt[(l & 15) * 32 + (l >> 4) * 8 + j]; }
  f4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) C[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}
// C[32x32] = A[32x16] * B[16x32]
__global__ void k_mfma32(const unsigned short* A, const unsigned short* Bt, float* C) {
  int l = threadIdx.x;
  s8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = A[(l & 31) * 16 + (l >> 5) * 8 + j]; b[j] = Bt[(l & 31) * 16 + (l >> 5) * 8 + j]; }
  f16v c; for (int r = 0; r < 16; ++r) c[r] = 0;
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}
// f32: C[16x16] = A[16x4] * B[4x16]
__global__ void k_mfma_f32(const float* A, const float* Bt, float* C) {
  int l = threadIdx.x;
  float a = A[(l & 15) * 4 + (l >> 4)], b = Bt[(l & 15) * 4 + (l >> 4)];
  f4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) C[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}
// K-strided operands through tr-read: A stored [k][i] (32x16), B stored [k][j] (32x16); C = A^T B
__global__ void k_mfma16_tr(const unsigned short* Ak, const unsigned short* Bk, float* C) {
  __shared__ __attribute__((aligned(16))) short la[32 * 16], lb[32 * 16];
  int l = threadIdx.x;
  for (int i = l; i < 512; i += 64) { la[i] = Ak[i]; lb[i] = Bk[i]; }
  __syncthreads();
  int g = l >> 4, t = l & 15;
  // hypothesis: within a 16-lane group lane t supplies the 8B chunk (row t>>2, cols 4*(t&3)..+3) of a 4x16 block and
  // receives column t of that block. k-assignment (same for A and B): j<4 -> k = 4g + j ; j>=4 -> k = 16 + 4g + (j-4)
  int r0 = 4 * g + (t >> 2), r1 = 16 + 4 * g + (t >> 2), c0 = (t & 3) * 4;
  s4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDSP(s4, la + r0 * 16 + c0));
  s4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDSP(s4, la + r1 * 16 + c0));
  s4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDSP(s4, lb + r0 * 16 + c0));
  s4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDSP(s4, lb + r1 * 16 + c0));
  s8 a = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
  s8 b = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
  f4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) C[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}
// buffer_load..lds OOB zero-fill: n valid shorts, lanes read 16B each at lane*16; lanes past the end must read 0
__global__ void k_buflds(const short* g, short* out, int nbytes, int hugelane) {
  __shared__ __attribute__((aligned(16))) short lds[64 * 8];
  for (int i = threadIdx.x; i < 512; i += 64) lds[i] = (short)0x7777;
  __syncthreads();
  auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, nbytes, 0x00020000);
  int voff = threadIdx.x * 16;
  if ((int)threadIdx.x == hugelane) voff = 0x7fffff00;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDSP(void, lds), 16, voff, 0, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 64) out[i] = lds[i];
}

int main() {
  int fails = 0;
  { // 1. tr-read dump
    short* d; CK(hipMalloc(&d, 256 * 2)); short h[256];
    for (int mode = 0; mode < 2; ++mode) {
      k_tr<<<1, 64>>>(d, mode); CK(hipMemcpy(h, d, 512, hipMemcpyDeviceToHost));
      printf("tr-read mode %d (lane: 4 values = lds element index)\n", mode);
      for (int l = 0; l < 64; ++l) { printf("  l%02d: %4d %4d %4d %4d%s", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3], (l & 3) == 3 ? "\n" : " |"); }
    }
    // check hypothesis on mode 0: lane (g,t): elem j == g*64 + j*16 + t
    k_tr<<<1, 64>>>(d, 0); CK(hipMemcpy(h, d, 512, hipMemcpyDeviceToHost));
    bool ok = true; for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) ok &= (h[l*4+j] == (l>>4)*64 + j*16 + (l&15));
    printf("TR_HYPOTHESIS_MODE0 %s\n", ok ? "PASS" : "FAIL"); fails += !ok;
    k_tr<<<1, 64>>>(d, 1); CK(hipMemcpy(h, d, 512, hipMemcpyDeviceToHost));
    ok = true; for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) ok &= (h[l*4+j] == (l>>4)*1024 + j*128 + (l&15));
    printf("TR_HYPOTHESIS_MODE1 %s\n", ok ? "PASS" : "FAIL"); fails += !ok;
  }
  auto rnd = [](){ return (float)((rand() % 9) - 4); };
  { // 2a. mfma 16x16x32
    std::vector<unsigned short> A(16*32), Bt(16*32); std::vector<float> Af(16*32), Bf(16*32), C(256), R(256);
    for (int i = 0; i < 512; ++i) { Af[i] = rnd(); Bf[i] = rnd(); A[i] = f2bf(Af[i]); Bt[i] = f2bf(Bf[i]); }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { float s = 0; for (int k = 0; k < 32; ++k) s += Af[i*32+k]*Bf[j*32+k]; R[i*16+j] = s; }
    unsigned short *dA, *dB; float* dC; CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1024)); CK(hipMalloc(&dC, 1024));
    CK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, Bt.data(), 1024, hipMemcpyHostToDevice));
    k_mfma16<<<1, 64>>>(dA, dB, dC); CK(hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost));
    bool ok = true; for (int i = 0; i < 256; ++i) ok &= (C[i] == R[i]);
    printf("MFMA_16x16x32_BF16 %s\n", ok ? "PASS" : "FAIL"); fails += !ok;
    // 2d. tr-read fed (A stored [k][i], B stored [k][j])
    std::vector<unsigned short> Ak(512), Bk(512);
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 32; ++k) { Ak[k*16+i] = A[i*32+k]; Bk[k*16+i] = Bt[i*32+k]; }
    CK(hipMemcpy(dA, Ak.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, Bk.data(), 1024, hipMemcpyHostToDevice));
    k_mfma16_tr<<<1, 64>>>(dA, dB, dC); CK(hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost));
    ok = true; for (int i = 0; i < 256; ++i) ok &= (C[i] == R[i]);
    printf("MFMA_16x16x32_BF16_TRREAD %s\n", ok ? "PASS" : "FAIL"); fails += !ok;
  }
  { // 2b. mfma 32x32x16
    std::vector<unsigned short> A(512), Bt(512); std::vector<float> Af(512), Bf(512), C(1024), R(1024);
    for (int i = 0; i < 512; ++i) { Af[i] = rnd(); Bf[i] = rnd(); A[i] = f2bf(Af[i]); Bt[i] = f2bf(Bf[i]); }
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float s = 0; for (int k = 0; k < 16; ++k) s += Af[i*16+k]*Bf[j*16+k]; R[i*32+j] = s; }
    unsigned short *dA, *dB; float* dC; CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1024)); CK(hipMalloc(&dC, 4096));
    CK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, Bt.data(), 1024, hipMemcpyHostToDevice));
    k_mfma32<<<1, 64>>>(dA, dB, dC); CK(hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost));
    bool ok = true; for (int i = 0; i < 1024; ++i) ok &= (C[i] == R[i]);
    printf("MFMA_32x32x16_BF16 %s\n", ok ? "PASS" : "FAIL"); fails += !ok;
  }
  { // 2c. f32 mfma
    std::vector<float> A(64), Bt(64), C(256), R(256);
    for (int i = 0; i < 64; ++i) { A[i] = rnd(); Bt[i] = rnd(); }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { float s = 0; for (int k = 0; k < 4; ++k) s += A[i*4+k]*Bt[j*4+k]; R[i*16+j] = s; }
    float *dA, *dB, *dC; CK(hipMalloc(&dA, 256)); CK(hipMalloc(&dB, 256)); CK(hipMalloc(&dC, 1024));
    CK(hipMemcpy(dA, A.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, Bt.data(), 256, hipMemcpyHostToDevice));
    k_mfma_f32<<<1, 64>>>(dA, dB, dC); CK(hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost));
    bool ok = true; for (int i = 0; i < 256; ++i) ok &= (C[i] == R[i]);
    printf("MFMA_16x16x4_F32 %s\n", ok ? "PASS" : "FAIL"); fails += !ok;
  }
  { // 3. buffer_load lds OOB
    short hsrc[512]; for (int i = 0; i < 512; ++i) hsrc[i] = (short)(i + 1);
    short *dsrc, *dout; CK(hipMalloc(&dsrc, 1024)); CK(hipMalloc(&dout, 1024));
    CK(hipMemcpy(dsrc, hsrc, 1024, hipMemcpyHostToDevice));
    int nbytes = 40 * 16;  // lanes >= 40 are out of bounds
    k_buflds<<<1, 64>>>(dsrc, dout, nbytes, 5); short h[512]; CK(hipMemcpy(h, dout, 1024, hipMemcpyDeviceToHost));
    bool ok = true;
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 8; ++j) {
      short want = (l < 40 && l != 5) ? (short)(l * 8 + j + 1) : 0;
      if (h[l*8+j] != want) { ok = false; if (j == 0) printf("  buflds lane %d got %d want %d\n", l, h[l*8+j], want); }
    }
    printf("BUFFER_LOAD_LDS_OOB_ZERO %s\n", ok ? "PASS" : "FAIL"); fails += !ok;
  }
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s CUs %d clock %d kHz memclk %d kHz L2 %d lds/blk %zu gcn %s\n", p.name, p.multiProcessorCount, p.clockRate, p.memoryClockRate, p.l2CacheSize, p.sharedMemPerBlock, p.gcnArchName);
  printf("PROBE_DONE fails=%d\n", fails);
  return 0;
}
