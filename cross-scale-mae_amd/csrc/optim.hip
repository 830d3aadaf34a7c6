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
                                                    float bc1, float bc2, bf16_t* __restrict__ p_lp, const float* __restrict__ gate,
                                                    const long long* __restrict__ tile_ks, bf16_t* __restrict__ p_ks) {
  // update gate (device scalar, nullable): the step's loss, summed over micro-steps and averaged over ranks with the gradients (it
  // rides in the tail slot of the flat gradient buffer).  A non-finite value means non-finite gradients on every rank: the update
  // is skipped as a whole — weights, moments and the bf16 mirror stay as they are — and the host raises at its next loss drain.
  if (gate != nullptr && !isfinite(gate[0])) return;
  const long long off = tile_off[blockIdx.x];
  const int cnt = tile_cnt[blockIdx.x];
  const float wd = tile_wd[blockIdx.x];
  const float step_size = lr / bc1, rbc2 = 1.f / sqrtf(bc2), decay = 1.f - lr * wd;
  // K-slab mirror (csmae.h csmae_gemm_ks) of a block Linear weight [N][K], written here instead of by a csmae_weights_kslab launch per step:
  // tile_ks[tile] = {flat offset of the weight's first element, N, K} (K = 0: this tile's parameter has none).  K % 32 == 0 and slots are
  // 8-aligned, so the 4 consecutive elements a thread owns stay inside one 32-wide K group: one 8-byte store.
  long long wbase = 0; int ksN = 0, ksK = 0;
  if (tile_ks != nullptr && p_ks != nullptr) { wbase = tile_ks[blockIdx.x * 3]; ksN = (int)tile_ks[blockIdx.x * 3 + 1]; ksK = (int)tile_ks[blockIdx.x * 3 + 2]; }
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
      if (ksK) {
        const long long e = off + i - wbase;
        const int n = (int)(e / ksK), k = (int)(e - (long long)n * ksK);
        if ((k & 3) == 0 && k + 4 <= ksK) st4<bf16_t>(p_ks + wbase + ((long long)(k >> 5) * ksN + n) * 32 + (k & 31), pp);
        else {   // (K % 4 != 0 or an unaligned slot: the four elements straddle a row or a 32-wide K group)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const long long eq = e + q;
            const int nq = (int)(eq / ksK), kq = (int)(eq - (long long)nq * ksK);
            p_ks[wbase + ((long long)(kq >> 5) * ksN + nq) * 32 + (kq & 31)] = f2bf(pp[q]);
          }
        }
      }
    } else {
      for (int k = i; k < cnt && k < i + 4; ++k) {
        float pp = p[off + k] * decay, gg = g[off + k];
        float mm = m[off + k] + (1.f - b1) * (gg - m[off + k]);
        float vv = b2 * v[off + k] + (1.f - b2) * gg * gg;
        pp -= step_size * (mm / (sqrtf(vv) * rbc2 + eps));
        p[off + k] = pp; m[off + k] = mm; v[off + k] = vv;
        if (p_lp) p_lp[off + k] = f2bf(pp);
        if (ksK) {
          const long long e = off + k - wbase;
          const int n = (int)(e / ksK), kk = (int)(e - (long long)n * ksK);
          p_ks[wbase + ((long long)(kk >> 5) * ksN + n) * 32 + (kk & 31)] = f2bf(pp);
        }
      }
    }
  }
}
extern "C" int csmae_adamw(long long ntiles, const long long* tile_off, const int* tile_cnt, const float* tile_wd, float* p, const float* g,
                           float* m, float* v, float lr, float beta1, float beta2, float eps, float bias_correction1, float bias_correction2,
                           void* p_lp, const float* gate, const long long* tile_ks, void* p_ks, void* stream) {
  CSMAE_REQUIRE(ntiles > 0 && tile_off && tile_cnt && tile_wd && p && g && m && v, "csmae_adamw: null argument");
  CSMAE_REQUIRE(bias_correction1 > 0.f && bias_correction2 > 0.f, "csmae_adamw: bias corrections must be positive (step >= 1)");
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)ntiles), dim3(256), 0, (hipStream_t)stream, tile_off, tile_cnt, tile_wd, p, g, m, v, lr, beta1, beta2, eps,
                     bias_correction1, bias_correction2, (bf16_t*)p_lp, gate, tile_ks, (bf16_t*)p_ks);
  return csmae_check_launch("csmae_adamw");
}

// ---- the same step for the block Linear weights of an fp8-mode model, writing their fp8 mirrors on the way (BASELINE.json configs[4]): W8 [out][in] for the
// forward products and W8^T [in][out] for dX (csmae_gemm_fp8 reads two K-contiguous operands).  A workgroup owns a 64 x 64 sub-block of one weight
// (tile8[tile] = {flat offset of the weight, N = out, K = in, n0, k0, weight index}; N, K multiples of 64): its rows are 256-B segments of the fp32
// buffers, the quantised block crosses LDS once so that both mirrors leave in 64-byte row segments.  DELAYED scaling, as for the activations: the scale
// is 448 / amax of the weight ONE STEP EARLIER (amax_prev [nw][64] partial maxima; a weight moves by ~lr per step, far below an e4m3 step), the new
// maximum is folded into amax_next (zeroed by the caller), dq[w] = amax_prev / 448.  Replaces csmae_fp8_weights' three launches per step
// (a second read of the 2.5 GB of masters).  A tripped gate leaves weights and mirrors alone and carries the old maxima over.
__global__ __launch_bounds__(256) void adamw_f8_kernel(const long long* __restrict__ tile8, float wd, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, float lr, float b1, float b2, float eps, float bc1, float bc2, bf16_t* __restrict__ p_lp,
                                                       const float* __restrict__ gate, unsigned char* __restrict__ w8, unsigned char* __restrict__ w8t,
                                                       const float* __restrict__ amax_prev, float* __restrict__ amax_next, float* __restrict__ dq) {
  const long long* t = tile8 + (long long)blockIdx.x * 6;
  const long long wbase = t[0];
  const int N = (int)t[1], K = (int)t[2], n0 = (int)t[3], k0 = (int)t[4], wi = (int)t[5];
  const int lane = threadIdx.x & 63;
  const float am = wave_max(amax_prev[(long long)wi * 64 + lane]);
  if (gate != nullptr && !isfinite(gate[0])) {
    if (n0 == 0 && k0 == 0 && threadIdx.x == 0 && am > 0.f) atomicMax(reinterpret_cast<unsigned*>(amax_next) + (long long)wi * 64, __float_as_uint(am));
    return;
  }
  __shared__ unsigned q8[64][17];   // the block's fp8 bytes: row r = 64 bytes (+ pad)
  __shared__ float red[4];
  const float scale = am > 0.f ? 448.0f / am : 1.f;
  if (n0 == 0 && k0 == 0 && threadIdx.x == 0) dq[wi] = am > 0.f ? am / 448.0f : 1.f;
  const float step_size = lr / bc1, rbc2 = 1.f / sqrtf(bc2), decay = 1.f - lr * wd;
  const int c4 = (threadIdx.x & 15) * 4, r0 = threadIdx.x >> 4;   // 16 threads x 4 columns per row, 16 rows per pass
  float seen = 0.f;
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int r = pass * 16 + r0;
    const long long e = wbase + (long long)(n0 + r) * K + k0 + c4;
    f4_t pp = *reinterpret_cast<f4_t*>(p + e), gg = *reinterpret_cast<const f4_t*>(g + e);
    f4_t mm = *reinterpret_cast<f4_t*>(m + e), vv = *reinterpret_cast<f4_t*>(v + e);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      pp[k] *= decay;
      mm[k] = mm[k] + (1.f - b1) * (gg[k] - mm[k]);
      vv[k] = b2 * vv[k] + (1.f - b2) * gg[k] * gg[k];
      pp[k] -= step_size * (mm[k] / (sqrtf(vv[k]) * rbc2 + eps));
      const float a = fabsf(pp[k]);
      seen = fmaxf(seen, a == a ? a : INFINITY);
    }
    *reinterpret_cast<f4_t*>(p + e) = pp; *reinterpret_cast<f4_t*>(m + e) = mm; *reinterpret_cast<f4_t*>(v + e) = vv;
    if (p_lp) st4<bf16_t>(p_lp + e, pp);
    f4_t q = pp * scale;
#pragma unroll
    for (int k = 0; k < 4; ++k) q[k] = q[k] != q[k] ? q[k] : fminf(fmaxf(q[k], -448.0f), 448.0f);
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(q[0], q[1], w, false); w = __builtin_amdgcn_cvt_pk_fp8_f32(q[2], q[3], w, true);
    *reinterpret_cast<unsigned*>(w8 + e) = (unsigned)w;
    q8[r][c4 >> 2] = (unsigned)w;
  }
  __syncthreads();
  // transposed mirror: thread -> column c of the block (a row of W8^T), 4 consecutive block rows r4 .. r4 + 3 (bytes along n)
  const unsigned char* qb = reinterpret_cast<const unsigned char*>(&q8[0][0]);
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int c = pass * 16 + r0, r4 = (threadIdx.x & 15) * 4;
    const unsigned b0 = qb[(r4 + 0) * 68 + c], b1_ = qb[(r4 + 1) * 68 + c], b2_ = qb[(r4 + 2) * 68 + c], b3 = qb[(r4 + 3) * 68 + c];
    *reinterpret_cast<unsigned*>(w8t + wbase + (long long)(k0 + c) * N + n0 + r4) = b0 | (b1_ << 8) | (b2_ << 16) | (b3 << 24);
  }
  seen = wave_max(seen);
  if (lane == 0) red[threadIdx.x >> 6] = seen;
  __syncthreads();
  if (threadIdx.x == 0) {
    seen = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    amax_publish(amax_next + (long long)wi * 64 + (blockIdx.x & 63), seen);
  }
}
extern "C" int csmae_adamw_fp8(long long ntiles, const long long* tile8, float weight_decay, float* p, const float* g, float* m, float* v, float lr, float beta1,
                               float beta2, float eps, float bias_correction1, float bias_correction2, void* p_lp, const float* gate, void* w8, void* w8t,
                               const float* amax_prev, float* amax_next, float* dq, void* stream) {
  CSMAE_REQUIRE(ntiles > 0 && tile8 && p && g && m && v && w8 && w8t && amax_prev && amax_next && dq, "csmae_adamw_fp8: null argument");
  CSMAE_REQUIRE(bias_correction1 > 0.f && bias_correction2 > 0.f, "csmae_adamw_fp8: bias corrections must be positive (step >= 1)");
  hipLaunchKernelGGL(adamw_f8_kernel, dim3((unsigned)ntiles), dim3(256), 0, (hipStream_t)stream, tile8, weight_decay, p, g, m, v, lr, beta1, beta2, eps, bias_correction1,
                     bias_correction2, (bf16_t*)p_lp, gate, (unsigned char*)w8, (unsigned char*)w8t, amax_prev, amax_next, dq);
  return csmae_check_launch("csmae_adamw_fp8");
}

// ---- global gradient norm + clip over the flat gradient buffer (util/misc.py:299-335: `torch.nn.utils.clip_grad_norm_(parameters, clip_grad)`
// or `get_grad_norm_`).  All gradients are one contiguous fp32 buffer (slots of frozen / unused parameters hold zeros), so the
// 2-norm is one streaming pass in a fixed order (deterministic two-stage sum, fp32 like torch's foreach norm) and the clip is one
// in-place scale by min(1, max_norm / (norm + 1e-6)) read from device memory: no host synchronisation anywhere.
#define NORM_BLOCKS 1024
__global__ __launch_bounds__(256) void sqnorm_partial_kernel(long long n, const float* __restrict__ g, float* __restrict__ partial) {
  __shared__ float red[4];
  const long long n4 = n >> 2;
  float s = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const f4_t x = *reinterpret_cast<const f4_t*>(g + i * 4);
    s += x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3];
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float x = g[n4 * 4 + threadIdx.x]; s += x * x; }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void norm_finish_kernel(const float* __restrict__ partial, float max_norm, float* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < NORM_BLOCKS; i += 256) s += partial[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float nrm = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
    out[0] = nrm;
    out[1] = max_norm > 0.f ? fminf(max_norm / (nrm + 1.0e-6f), 1.0f) : 1.0f;   // torch: clamp(max_norm / (total_norm + 1e-6), max=1.0)
  }
}
__global__ __launch_bounds__(256) void grad_scale_kernel(long long n, float* __restrict__ g, const float* __restrict__ coef) {
  const float c = coef[1];
  if (c == 1.0f) return;   // (x * 1.0f is x: nothing to write; a NaN coefficient — non-finite norm — is applied, as torch does)
  const long long n4 = n >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    f4_t x = *reinterpret_cast<f4_t*>(g + i * 4);
    x *= c;
    *reinterpret_cast<f4_t*>(g + i * 4) = x;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) g[n4 * 4 + threadIdx.x] *= c;
}
extern "C" int csmae_clip_grad_norm(long long n, float* g, float max_norm, float* scratch, float* out, void* stream) {
  CSMAE_REQUIRE(n > 0 && g && scratch && out && (((uintptr_t)g & 15) == 0), "csmae_clip_grad_norm: bad arguments (g must be 16-byte aligned, scratch >= 1024 floats)");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(NORM_BLOCKS), dim3(256), 0, st, n, g, scratch);
  hipLaunchKernelGGL(norm_finish_kernel, dim3(1), dim3(256), 0, st, scratch, max_norm, out);
  if (max_norm > 0.f) hipLaunchKernelGGL(grad_scale_kernel, dim3(2048), dim3(256), 0, st, n, g, out);
  return csmae_check_launch("csmae_clip_grad_norm");
}

// update gate bookkeeping: slot (+)= loss (accumulate = 0: slot = loss).  The slot is the tail element of the flat gradient buffer.
__global__ void gate_kernel(const float* __restrict__ loss, float* __restrict__ slot, int accumulate) {
  slot[0] = accumulate ? slot[0] + loss[0] : loss[0];
}
extern "C" int csmae_gate_accumulate(const float* loss, float* slot, int accumulate, void* stream) {
  CSMAE_REQUIRE(loss && slot, "csmae_gate_accumulate: null argument");
  hipLaunchKernelGGL(gate_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, loss, slot, accumulate);
  return csmae_check_launch("csmae_gate_accumulate");
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

__global__ __launch_bounds__(256) void cast_f32_kernel(long long n, const bf16_t* __restrict__ src, float* __restrict__ dst) {
  const long long n4 = n >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x)
    *reinterpret_cast<f4_t*>(dst + i * 4) = ld4<bf16_t>(src + i * 4);
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) dst[n4 * 4 + threadIdx.x] = bf2f(src[n4 * 4 + threadIdx.x]);
}
extern "C" int csmae_cast_bf16_to_f32(long long n, const void* src, float* dst, void* stream) {
  CSMAE_REQUIRE(n > 0 && src && dst && (((uintptr_t)src & 7) == 0) && (((uintptr_t)dst & 15) == 0), "csmae_cast_bf16_to_f32: bad args");
  hipLaunchKernelGGL(cast_f32_kernel, dim3((unsigned)fmin((double)cdiv(n, 1024), 4096.0)), dim3(256), 0, (hipStream_t)stream, n, (const bf16_t*)src, dst);
  return csmae_check_launch("csmae_cast_bf16_to_f32");
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
