// GEMM family for the ViT encoder/decoder blocks (QKV / proj / fc1 / fc2 and their
// backward products).  C[M,N] = sum_k A(m,k) * B(k,n) with fused epilogues.
//
//   transA = 0 : A stored [M][K]  (K contiguous)      transA = 1 : A stored [K][M]
//   transB = 0 : B stored [N][K]  (K contiguous)      transB = 1 : B stored [K][N]
//
//   forward   y = x W^T        : transA=0, transB=0  (torch Linear weight is [out,in] = [N][K])
//   dX = dY W                  : transA=0, transB=1  (W is [K=out][N=in])
//   dW = dY^T X (reduce tokens): transA=1, transB=1  (dY is [K=tok][M=out], X is [K=tok][N=in])
//
// bf16 path (gfx950): 128x128x64 block tile, 4 waves (2x2), each wave 64x64 = 4x4 MFMA 16x16x32 tiles,
// operands staged HBM->LDS by `buffer_load ... lds` (16 B/lane, no VGPR round trip, OOB -> 0 gives free
// edge handling), double-buffered LDS, one barrier per K-tile.  K-contiguous operands are read with
// ds_read_b128 from an XOR-swizzled image (conflict free); K-strided operands are read with the gfx950
// transpose read ds_read_b64_tr_b16 from a 32-B-granule swizzled image, so no transposed copies of
// weights/activations are ever materialised.  MFMA operands are swapped (D = B_frag x A_frag) so each lane
// ends with 4 consecutive output columns -> 8/16-byte epilogue accesses.
// fp32 path: exact-fp32 FMA tile kernel used for the 1e-4 parity mode (not the throughput path).
#include "common.h"

#define EPI_NONE 0    // C = acc + bias
#define EPI_GELU 1    // aux = acc + bias ; C = gelu(aux)
#define EPI_RESID 2   // C = acc + bias + resid (resid fp32)
#define EPI_DGELU 3   // C = acc * gelu'(aux)
#define EPI_ATOMIC 4  // C(fp32) += acc   (split-K weight gradients)

struct GemmArgs {
  const void* A; const void* B; void* C; const float* bias; void* aux; const float* resid;
  long long lda, ldb, ldc, ldaux, ldr;
  int M, N, K;
  int c_dtype, epi, splitk, tiles_m, tiles_n, ktiles, ktiles_per_split;
  unsigned a_bytes, b_bytes;
};

// ------------------------------------------------------------------------------------ epilogue
template <typename TC>
__device__ __forceinline__ void epi_store4(const GemmArgs& p, int m, int n, f4_t v) {
  if (p.bias) { f4_t b = *reinterpret_cast<const f4_t*>(p.bias + n); v += b; }
  TC* c = reinterpret_cast<TC*>(p.C) + (long long)m * p.ldc + n;
  if (p.epi == EPI_GELU) {
    // the saved pre-activation is what backward sees: evaluate gelu on the value as stored (rounded for bf16)
    v = round4<TC>(v);
    st4<TC>(reinterpret_cast<TC*>(p.aux) + (long long)m * p.ldaux + n, v);
    v[0] = gelu_erf(v[0]); v[1] = gelu_erf(v[1]); v[2] = gelu_erf(v[2]); v[3] = gelu_erf(v[3]);
  } else if (p.epi == EPI_RESID) {
    f4_t r = *reinterpret_cast<const f4_t*>(p.resid + (long long)m * p.ldr + n);
    v += r;
  } else if (p.epi == EPI_DGELU) {
    f4_t q = ld4<TC>(reinterpret_cast<const TC*>(p.aux) + (long long)m * p.ldaux + n);
    v[0] *= gelu_erf_grad(q[0]); v[1] *= gelu_erf_grad(q[1]); v[2] *= gelu_erf_grad(q[2]); v[3] *= gelu_erf_grad(q[3]);
  }
  st4<TC>(c, v);
}

__device__ __forceinline__ void epi_dispatch(const GemmArgs& p, int m, int n, f4_t v) {
  if (m >= p.M || n >= p.N) return;
  if (p.epi == EPI_ATOMIC) {
    float* c = reinterpret_cast<float*>(p.C) + (long long)m * p.ldc + n;
    unsafeAtomicAdd(c + 0, v[0]); unsafeAtomicAdd(c + 1, v[1]); unsafeAtomicAdd(c + 2, v[2]); unsafeAtomicAdd(c + 3, v[3]);
  } else if (p.c_dtype == CSMAE_BF16) {
    epi_store4<bf16_t>(p, m, n, v);
  } else {
    epi_store4<float>(p, m, n, v);
  }
}

// XCD-aware, bijective block remap: blocks that share the A row panel land on the same XCD/L2.
__device__ __forceinline__ int xcd_remap(int b, int nwg) {
  int xcd = b & 7, idx = b >> 3, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// ------------------------------------------------------------------------------------ bf16 MFMA
#define OOB_OFF 0xFFFFFFF0u
__device__ __forceinline__ int tr_key(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }

template <bool TA, bool TB>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(GemmArgs p) {
  __shared__ __attribute__((aligned(16))) char smem[65536];  // [2 buffers][A 16 KiB | B 16 KiB]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int t = lane & 15, g = lane >> 4;
  const int tiles = p.tiles_m * p.tiles_n;
  const int wg = xcd_remap(blockIdx.x, tiles * p.splitk);
  const int split = wg / tiles, tile = wg - split * tiles;
  const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
  const int m0 = tm * 128, n0 = tn * 128;
  const int kt_begin = split * p.ktiles_per_split;
  const int kt_end = min(kt_begin + p.ktiles_per_split, p.ktiles);

  auto rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, p.a_bytes, 0x00020000);
  auto rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.B), 0, p.b_bytes, 0x00020000);

  // per-lane staging descriptors: 4 DMA pieces (1 KiB each) per operand per wave
  unsigned offA[4], offB[4];   // byte offset at kt = 0 (without the k advance)
  int kA[4], kB[4];            // k index (within tile) whose validity must be checked against K
  bool rowokA[4], rowokB[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int piece = w * 4 + q;
    if (!TA) {  // tile image [128 rows m][64 k], 128-B rows, 16-B chunk swizzle c ^ (row & 7)
      int row = piece * 8 + (lane >> 3), c = (lane & 7) ^ (row & 7);
      long long gr = (long long)m0 + row;
      rowokA[q] = gr < p.M; kA[q] = c * 8;
      offA[q] = (unsigned)((gr * p.lda + c * 8) * 2);
    } else {    // tile image [64 rows k][128 m], 256-B rows, 32-B granule swizzle
      int row = piece * 4 + (lane >> 4), s16 = lane & 15;
      int ch = ((((s16 >> 1) ^ tr_key(row)) << 1) | (s16 & 1));
      rowokA[q] = true; kA[q] = row;
      offA[q] = (unsigned)(((long long)row * p.lda + m0 + ch * 8) * 2);
    }
    if (!TB) {
      int row = piece * 8 + (lane >> 3), c = (lane & 7) ^ (row & 7);
      long long gr = (long long)n0 + row;
      rowokB[q] = gr < p.N; kB[q] = c * 8;
      offB[q] = (unsigned)((gr * p.ldb + c * 8) * 2);
    } else {
      int row = piece * 4 + (lane >> 4), s16 = lane & 15;
      int ch = ((((s16 >> 1) ^ tr_key(row)) << 1) | (s16 & 1));
      rowokB[q] = true; kB[q] = row;
      offB[q] = (unsigned)(((long long)row * p.ldb + n0 + ch * 8) * 2);
    }
  }
  const unsigned kstepA = TA ? (unsigned)(64 * p.lda * 2) : 128u;
  const unsigned kstepB = TB ? (unsigned)(64 * p.ldb * 2) : 128u;

  auto stage = [&](int buf, int kt) {
    char* sa = smem + buf * 32768;
    char* sb = sa + 16384;
    const int k0 = kt * 64;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      unsigned oa = (rowokA[q] && (k0 + kA[q] < p.K)) ? offA[q] + (unsigned)kt * kstepA : OOB_OFF;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(void, sa + (w * 4 + q) * 1024), 16, (int)oa, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      unsigned ob = (rowokB[q] && (k0 + kB[q] < p.K)) ? offB[q] + (unsigned)kt * kstepB : OOB_OFF;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, LDS_PTR(void, sb + (w * 4 + q) * 1024), 16, (int)ob, 0, 0, 0);
    }
  };

  f4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f4_t{0.f, 0.f, 0.f, 0.f};

  const int wm = (w >> 1) * 64, wn = (w & 1) * 64;
  // fragment read offsets (bytes, within an operand image), k-slice 0; slice 1 adds a constant
  int rdA[4], rdB[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (!TA) { int row = wm + i * 16 + t; rdA[i] = row * 128; }
    else     { rdA[i] = (wm >> 4) + i; }
    if (!TB) { int row = wn + i * 16 + t; rdB[i] = row * 128; }
    else     { rdB[i] = (wn >> 4) + i; }
  }

  if (kt_begin < kt_end) stage(0, kt_begin);
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const int buf = (kt - kt_begin) & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < kt_end) stage(buf ^ 1, kt + 1);
    const char* sa = smem + buf * 32768;
    const char* sb = sa + 16384;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      s8_t fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (!TA) {
          int row7 = (wm + i * 16 + t) & 7;
          fa[i] = *reinterpret_cast<const s8_t*>(sa + rdA[i] + (((ks * 4 + g) ^ row7) << 4));
        } else {
          int kr = ks * 32 + 8 * g + (t >> 2);
          int key = (t >> 2) | ((g & 1) << 2);
          const char* base = sa + ((rdA[i] ^ key) << 5) + (t & 3) * 8;
          s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s4_t, base + kr * 256));
          s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s4_t, base + (kr + 4) * 256));
          fa[i] = s8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
        if (!TB) {
          int row7 = (wn + i * 16 + t) & 7;
          fb[i] = *reinterpret_cast<const s8_t*>(sb + rdB[i] + (((ks * 4 + g) ^ row7) << 4));
        } else {
          int kr = ks * 32 + 8 * g + (t >> 2);
          int key = (t >> 2) | ((g & 1) << 2);
          const char* base = sb + ((rdB[i] ^ key) << 5) + (t & 3) * 8;
          s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s4_t, base + kr * 256));
          s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s4_t, base + (kr + 4) * 256));
          fb[i] = s8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8_t, fb[j]), __builtin_bit_cast(bf8_t, fa[i]), acc[i][j], 0, 0, 0);
    }
  }
  if (kt_begin >= kt_end && p.epi == EPI_ATOMIC) return;
  // lane (t,g) holds C[m = .. + t][n = .. + 4g + r]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      epi_dispatch(p, m0 + wm + i * 16 + t, n0 + wn + j * 16 + 4 * g, acc[i][j]);
}

// ------------------------------------------------------------------------------------ fp32 exact
template <bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs p) {
  __shared__ float As[16][68], Bs[16][68];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int tiles = p.tiles_m * p.tiles_n;
  const int wg = blockIdx.x;
  const int split = wg / tiles, tile = wg - split * tiles;
  const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
  const int m0 = tm * 64, n0 = tn * 64;
  const float* A = reinterpret_cast<const float*>(p.A);
  const float* B = reinterpret_cast<const float*>(p.B);
  const int kt_begin = split * p.ktiles_per_split, kt_end = min(kt_begin + p.ktiles_per_split, p.ktiles);
  float acc[4][4] = {};
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const int k0 = kt * 16;
#pragma unroll
    for (int e = tid; e < 1024; e += 256) {
      int mm, kk;
      if (!TA) { mm = e >> 4; kk = e & 15; } else { kk = e >> 6; mm = e & 63; }
      long long gm = m0 + mm, gk = k0 + kk;
      float v = 0.f;
      if (gm < p.M && gk < p.K) v = TA ? A[gk * p.lda + gm] : A[gm * p.lda + gk];
      As[kk][mm] = v;
      int nn;
      if (!TB) { nn = e >> 4; kk = e & 15; } else { kk = e >> 6; nn = e & 63; }
      long long gn = n0 + nn; gk = k0 + kk;
      v = 0.f;
      if (gn < p.N && gk < p.K) v = TB ? B[gk * p.ldb + gn] : B[gn * p.ldb + gk];
      Bs[kk][nn] = v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { a[i] = As[k][ty * 4 + i]; b[i] = Bs[k][tx * 4 + i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
  if (kt_begin >= kt_end && p.epi == EPI_ATOMIC) return;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    epi_dispatch(p, m0 + ty * 4 + i, n0 + tx * 4, f4_t{acc[i][0], acc[i][1], acc[i][2], acc[i][3]});
}

// ------------------------------------------------------------------------------------ C ABI
extern "C" int csmae_gemm(int dtype, int transA, int transB, long long M, long long N, long long K,
                          const void* A, long long lda, const void* B, long long ldb,
                          void* C, long long ldc, int c_dtype, const float* bias, int epilogue,
                          void* aux, long long ldaux, const float* resid, long long ldr,
                          int splitk, void* stream) {
  CSMAE_REQUIRE(M > 0 && N > 0 && K > 0, "csmae_gemm: empty problem M=%lld N=%lld K=%lld", M, N, K);
  CSMAE_REQUIRE(N % 4 == 0 && ldc % 4 == 0, "csmae_gemm: N and ldc must be multiples of 4 (N=%lld ldc=%lld)", N, ldc);
  CSMAE_REQUIRE(epilogue >= EPI_NONE && epilogue <= EPI_ATOMIC, "csmae_gemm: bad epilogue %d", epilogue);
  CSMAE_REQUIRE(epilogue != EPI_ATOMIC || c_dtype == CSMAE_F32, "csmae_gemm: atomic accumulate needs fp32 C");
  CSMAE_REQUIRE(!(epilogue == EPI_GELU || epilogue == EPI_DGELU) || (aux && ldaux % 4 == 0), "csmae_gemm: gelu epilogues need aux");
  CSMAE_REQUIRE(epilogue != EPI_RESID || (resid && ldr % 4 == 0), "csmae_gemm: residual epilogue needs resid");
  CSMAE_REQUIRE(splitk >= 1 && (splitk == 1 || epilogue == EPI_ATOMIC), "csmae_gemm: split-K only with atomic accumulate");
  GemmArgs p;
  p.A = A; p.B = B; p.C = C; p.bias = (epilogue == EPI_ATOMIC || epilogue == EPI_DGELU) ? nullptr : bias; p.aux = aux; p.resid = resid;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldaux = ldaux; p.ldr = ldr;
  p.M = (int)M; p.N = (int)N; p.K = (int)K; p.c_dtype = c_dtype; p.epi = epilogue; p.splitk = splitk;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CSMAE_BF16) {
    CSMAE_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && K % 8 == 0, "csmae_gemm(bf16): lda, ldb, K must be multiples of 8 (lda=%lld ldb=%lld K=%lld)", lda, ldb, K);
    CSMAE_REQUIRE((((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15) == 0, "csmae_gemm(bf16): A, B, C must be 16-byte aligned");
    // K-strided operands are fetched in 8-column chunks: the row allocation must cover the rounded-up width
    CSMAE_REQUIRE(!transA || lda >= (M + 7) / 8 * 8, "csmae_gemm(bf16): transposed A needs lda >= roundup8(M)");
    CSMAE_REQUIRE(!transB || ldb >= (N + 7) / 8 * 8, "csmae_gemm(bf16): transposed B needs ldb >= roundup8(N)");
    long long abytes = (transA ? K : M) * lda * 2, bbytes = (transB ? K : N) * ldb * 2;
    CSMAE_REQUIRE(abytes < 0xFFFFFFF0ll && bbytes < 0xFFFFFFF0ll, "csmae_gemm(bf16): operand larger than 4 GiB");
    p.a_bytes = (unsigned)abytes; p.b_bytes = (unsigned)bbytes;
    p.tiles_m = cdiv(M, 128); p.tiles_n = cdiv(N, 128); p.ktiles = cdiv(K, 64);
    p.ktiles_per_split = cdiv(p.ktiles, splitk);
    p.splitk = cdiv(p.ktiles, p.ktiles_per_split);
    dim3 grid(p.tiles_m * p.tiles_n * p.splitk), block(256);
    if (!transA && !transB) hipLaunchKernelGGL((gemm_bf16_kernel<false, false>), grid, block, 0, st, p);
    else if (!transA && transB) hipLaunchKernelGGL((gemm_bf16_kernel<false, true>), grid, block, 0, st, p);
    else if (transA && transB) hipLaunchKernelGGL((gemm_bf16_kernel<true, true>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((gemm_bf16_kernel<true, false>), grid, block, 0, st, p);
  } else if (dtype == CSMAE_F32) {
    p.a_bytes = p.b_bytes = 0;
    p.tiles_m = cdiv(M, 64); p.tiles_n = cdiv(N, 64); p.ktiles = cdiv(K, 16);
    p.ktiles_per_split = cdiv(p.ktiles, splitk);
    p.splitk = cdiv(p.ktiles, p.ktiles_per_split);
    dim3 grid(p.tiles_m * p.tiles_n * p.splitk), block(256);
    if (!transA && !transB) hipLaunchKernelGGL((gemm_f32_kernel<false, false>), grid, block, 0, st, p);
    else if (!transA && transB) hipLaunchKernelGGL((gemm_f32_kernel<false, true>), grid, block, 0, st, p);
    else if (transA && transB) hipLaunchKernelGGL((gemm_f32_kernel<true, true>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((gemm_f32_kernel<true, false>), grid, block, 0, st, p);
  } else {
    csmae_set_error("csmae_gemm: unsupported dtype %d", dtype);
    return CSMAE_ERR_UNSUPPORTED;
  }
  return csmae_check_launch("csmae_gemm");
}
