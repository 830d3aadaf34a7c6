// Optimizer-side and bookkeeping kernels (HBM-bound, 16-B accesses):
//   adamw      fused multi-tensor AdamW over the flat fp32 parameter/grad/moment buffers (main_pretrain.py:426-427:
//              torch.optim.AdamW semantics, betas (0.9, 0.95), decoupled weight decay per tile), optionally refreshing the
//              bf16 weight mirror the MFMA GEMMs read.
//   cast       fp32 -> bf16 mirror refresh
//   colsum     bias gradients: db[n] += sum_m dY[m, n]
#include "common.h"

// hyper-parameters travel by value (kernel arguments): no device-side staging buffer whose host copy could be overwritten by a
// CPU that runs a step ahead of the GPU
__global__ __launch_bounds__(256) void adamw_kernel(const long long* __restrict__ tile_off, const int* __restrict__ tile_cnt,
                                                    const float* __restrict__ tile_wd, float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, float lr, float b1, float b2, float eps,
                                                    float bc1, float bc2, bf16_t* __restrict__ p_lp) {
  const long long off = tile_off[blockIdx.x];
  const int cnt = tile_cnt[blockIdx.x];
  const float wd = tile_wd[blockIdx.x];
  const float step_size = lr / bc1, rbc2 = 1.f / sqrtf(bc2), decay = 1.f - lr * wd;
  for (int i = threadIdx.x * 4; i < cnt; i += blockDim.x * 4) {
    if (i + 4 <= cnt && ((off + i) & 3) == 0) {
      f4_t pp = *reinterpret_cast<f4_t*>(p + off + i), gg = *reinterpret_cast<const f4_t*>(g + off + i);
      f4_t mm = *reinterpret_cast<f4_t*>(m + off + i), vv = *reinterpret_cast<f4_t*>(v + off + i);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        pp[k] *= decay;
        mm[k] = mm[k] + (1.f - b1) * (gg[k] - mm[k]);   // torch: exp_avg.lerp_(grad, 1 - beta1)
        vv[k] = b2 * vv[k] + (1.f - b2) * gg[k] * gg[k];
        pp[k] -= step_size * (mm[k] / (sqrtf(vv[k]) * rbc2 + eps));
      }
      *reinterpret_cast<f4_t*>(p + off + i) = pp; *reinterpret_cast<f4_t*>(m + off + i) = mm; *reinterpret_cast<f4_t*>(v + off + i) = vv;
      if (p_lp) st4<bf16_t>(p_lp + off + i, pp);
    } else {
      for (int k = i; k < cnt && k < i + 4; ++k) {
        float pp = p[off + k] * decay, gg = g[off + k];
        float mm = m[off + k] + (1.f - b1) * (gg - m[off + k]);
        float vv = b2 * v[off + k] + (1.f - b2) * gg * gg;
        pp -= step_size * (mm / (sqrtf(vv) * rbc2 + eps));
        p[off + k] = pp; m[off + k] = mm; v[off + k] = vv;
        if (p_lp) p_lp[off + k] = f2bf(pp);
      }
    }
  }
}
extern "C" int csmae_adamw(long long ntiles, const long long* tile_off, const int* tile_cnt, const float* tile_wd, float* p, const float* g,
                           float* m, float* v, float lr, float beta1, float beta2, float eps, float bias_correction1, float bias_correction2,
                           void* p_lp, void* stream) {
  CSMAE_REQUIRE(ntiles > 0 && tile_off && tile_cnt && tile_wd && p && g && m && v, "csmae_adamw: null argument");
  CSMAE_REQUIRE(bias_correction1 > 0.f && bias_correction2 > 0.f, "csmae_adamw: bias corrections must be positive (step >= 1)");
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)ntiles), dim3(256), 0, (hipStream_t)stream, tile_off, tile_cnt, tile_wd, p, g, m, v, lr, beta1, beta2, eps,
                     bias_correction1, bias_correction2, (bf16_t*)p_lp);
  return csmae_check_launch("csmae_adamw");
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(long long n, const float* __restrict__ src, bf16_t* __restrict__ dst) {
  const long long n4 = n >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x)
    st4<bf16_t>(dst + i * 4, *reinterpret_cast<const f4_t*>(src + i * 4));
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) dst[n4 * 4 + threadIdx.x] = f2bf(src[n4 * 4 + threadIdx.x]);
}
extern "C" int csmae_cast_f32_to_bf16(long long n, const float* src, void* dst, void* stream) {
  CSMAE_REQUIRE(n > 0 && src && dst && (((uintptr_t)src & 15) == 0) && (((uintptr_t)dst & 7) == 0), "csmae_cast_f32_to_bf16: bad args");
  hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)fmin((double)cdiv(n, 1024), 4096.0)), dim3(256), 0, (hipStream_t)stream, n, src, (bf16_t*)dst);
  return csmae_check_launch("csmae_cast_f32_to_bf16");
}

// out[n] += sum_m x[m, n].  block = 64 column-quads x 4 row lanes... 256 threads: tx = column quad (0..63), ty = row lane (0..3)
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(long long M, int N, const T* __restrict__ x, long long ld, float* __restrict__ out) {
  __shared__ float red[4][64][4];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int col = (blockIdx.x * 64 + tx) * 4;
  f4_t s = {0.f, 0.f, 0.f, 0.f};
  if (col < N)
    for (long long r = (long long)blockIdx.y * 4 + ty; r < M; r += (long long)gridDim.y * 4) s += ld4<T>(x + r * ld + col);
  for (int k = 0; k < 4; ++k) red[ty][tx][k] = s[k];
  __syncthreads();
  if (ty == 0 && col < N)
    for (int k = 0; k < 4; ++k) unsafeAtomicAdd(out + col + k, red[0][tx][k] + red[1][tx][k] + red[2][tx][k] + red[3][tx][k]);
}
int csmae_colsum_launch(int dtype, long long M, int N, const void* x, long long ld, float* out, void* stream);
extern "C" int csmae_colsum(int dtype, long long M, int N, const void* x, long long ld, float* out, void* stream) {
  return csmae_colsum_launch(dtype, M, N, x, ld, out, stream);
}
int csmae_colsum_launch(int dtype, long long M, int N, const void* x, long long ld, float* out, void* stream) {
  CSMAE_REQUIRE(M > 0 && N > 0 && N % 4 == 0 && ld % 4 == 0, "csmae_colsum: N and ld must be multiples of 4");
  int gx = cdiv(N, 256), gy = (int)fmin((double)cdiv(M, 64), fmax(1.0, 1024.0 / gx));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CSMAE_BF16) hipLaunchKernelGGL((colsum_kernel<bf16_t>), dim3(gx, gy), dim3(256), 0, st, M, N, (const bf16_t*)x, ld, out);
  else if (dtype == CSMAE_F32) hipLaunchKernelGGL((colsum_kernel<float>), dim3(gx, gy), dim3(256), 0, st, M, N, (const float*)x, ld, out);
  else { csmae_set_error("csmae_colsum: bad dtype %d", dtype); return CSMAE_ERR_UNSUPPORTED; }
  return csmae_check_launch("csmae_colsum");
}
