// OCP fp8 (e4m3 / e5m2) quantisation for the fp8 MFMA GEMM path (BASELINE.json configs[4]; csmae_gemm_fp8 in gemm.hip).
// Per-tensor scaling: amax of the tensor (one atomic max per workgroup into 64 device slots the caller zeroed), then
// q = cvt_fp8(x * FMAX / amax) and the de-quantisation factor amax / FMAX for the GEMM epilogue.  "Current" scaling runs both passes on
// the tensor; "delayed" scaling (activations from the second step on) quantises in ONE pass with the amax the same tensor had in the
// previous step — values beyond it saturate — and records this step's amax for the next (amax_next).  Nothing is staged on the host.
#include "common.h"

// An amax is kept as FP8_SLOTS partial maxima (producers hash their workgroup id into a slot, readers take the maximum of the slots):
// thousands of same-address atomic maxima per launch serialise in L2 (measured: 16 k of them cost 100 us), 64 addresses do not.
#define FP8_SLOTS 64
__device__ __forceinline__ float fp8_amax_read(const float* slots) { return wave_max(slots[threadIdx.x & (FP8_SLOTS - 1)]); }
#define FP8_E4M3_MAX 448.0f
#define FP8_E5M2_MAX 57344.0f

template <typename T>
__device__ __forceinline__ void fp8_amax_body(long long rows, int cols, const T* __restrict__ src, long long ld, float* __restrict__ amax) {
  __shared__ float red[4];
  const int cv = cols >> 2;
  float m = 0.f;
  for (long long r = blockIdx.x; r < rows; r += gridDim.x)
    for (int c = threadIdx.x; c < cv; c += blockDim.x) {
      const f4_t v = ld4<T>(src + r * ld + c * 4);
      m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    amax_publish(amax + (blockIdx.x & (FP8_SLOTS - 1)), m);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void fp8_amax_kernel(long long rows, int cols, const T* __restrict__ src, long long ld, float* __restrict__ amax) {
  fp8_amax_body<T>(rows, cols, src, ld, amax);
}

template <int FMT> __device__ __forceinline__ unsigned pack4_fp8(f4_t v) {
  int p = 0;
  if (FMT == 0) { p = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], p, false); p = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], p, true); }
  else { p = __builtin_amdgcn_cvt_pk_bf8_f32(v[0], v[1], p, false); p = __builtin_amdgcn_cvt_pk_bf8_f32(v[2], v[3], p, true); }
  return (unsigned)p;
}

// dst[r][c] (TR = 0) or dst[c][r] (TR = 1: the pre-transposed weight mirror the dX products read) = fp8(src[r][c] * scale)
template <typename T, int FMT, int TR>
__device__ __forceinline__ void fp8_quant_body(long long rows, int cols, const T* __restrict__ src, long long ld, unsigned char* __restrict__ dst,
                                               long long ldd, const float* __restrict__ amax, float* __restrict__ dq, float* __restrict__ amax_next) {
  const float fmax = FMT == 0 ? FP8_E4M3_MAX : FP8_E5M2_MAX;
  const float am = fp8_amax_read(amax);
  float seen = 0.f;   // delayed scaling: max|x| of THIS tensor, for the next step's scale (values beyond the old amax saturate)
  const float scale = am > 0.f ? fmax / am : 1.f;
  if (blockIdx.x == 0 && threadIdx.x == 0) dq[0] = am > 0.f ? am / fmax : 1.f;
  if (!TR) {
    const int cv = cols >> 2;
    for (long long r = blockIdx.x; r < rows; r += gridDim.x)
      for (int c = threadIdx.x; c < cv; c += blockDim.x) {
        f4_t v = ld4<T>(src + r * ld + c * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) { const float a = fabsf(v[k]); seen = fmaxf(seen, a == a ? a : INFINITY); }   // NaN / Inf -> an amax of +Inf (the next step's scale is NaN: the loss gate trips)
        v *= scale;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = v[k] != v[k] ? v[k] : fminf(fmaxf(v[k], -fmax), fmax);   // (NaN is not clamped away)
        *reinterpret_cast<unsigned*>(dst + r * ldd + c * 4) = pack4_fp8<FMT>(v);
      }
    if (amax_next) {   // one atomic per workgroup (same-address atomics serialise in L2)
      __shared__ float red[4];
      seen = wave_max(seen);
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = seen;
      __syncthreads();
      if (threadIdx.x == 0) {
        seen = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        amax_publish(amax_next + (blockIdx.x & (FP8_SLOTS - 1)), seen);
      }
    }
  } else {  // 64 x 64 tiles through LDS: reads along source rows, writes along destination rows
    __shared__ float tile[64][65];
    const long long tiles_c = (cols + 63) / 64, ntiles = ((rows + 63) / 64) * tiles_c;
    for (long long tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
      const long long r0 = (tl / tiles_c) * 64; const int c0 = (int)(tl % tiles_c) * 64;
      __syncthreads();
      for (int e = threadIdx.x; e < 64 * 16; e += 256) {
        const int rr = e >> 4, c4 = (e & 15) * 4;
        f4_t v = {0.f, 0.f, 0.f, 0.f};
        if (r0 + rr < rows && c0 + c4 < cols) v = ld4<T>(src + (r0 + rr) * ld + c0 + c4);
#pragma unroll
        for (int k = 0; k < 4; ++k) tile[rr][c4 + k] = v[k];
      }
      __syncthreads();
      for (int e = threadIdx.x; e < 64 * 16; e += 256) {
        const int cc = e >> 4, r4 = (e & 15) * 4;
        if (c0 + cc < cols && r0 + r4 < rows) {
          f4_t v = f4_t{tile[r4][cc], tile[r4 + 1][cc], tile[r4 + 2][cc], tile[r4 + 3][cc]} * scale;
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] = v[k] != v[k] ? v[k] : fminf(fmaxf(v[k], -fmax), fmax);
          *reinterpret_cast<unsigned*>(dst + (long long)(c0 + cc) * ldd + r0 + r4) = pack4_fp8<FMT>(v);
        }
      }
    }
  }
}

template <typename T, int FMT, int TR>
__global__ __launch_bounds__(256) void fp8_quant_kernel(long long rows, int cols, const T* __restrict__ src, long long ld, unsigned char* __restrict__ dst,
                                                        long long ldd, const float* __restrict__ amax, float* __restrict__ dq, float* __restrict__ amax_next) {
  fp8_quant_body<T, FMT, TR>(rows, cols, src, ld, dst, ldd, amax, dq, amax_next);
}

// ---- all fp8 weight mirrors of a model in three launches (blockIdx.y = weight): per-tensor amax of the fp32 masters, W8 [out][in] (e4m3) for
// the forward products, W8^T [in][out] for dX.  desc[k] = {offset of weight k in the flat parameter buffer (= in both byte mirrors), out, in};
// amax [count][64] (zeroed by the caller), dq [count].  Replaces four launches per weight and step (640 for ViT-H/14: ~2 % of its fp8 step).
__global__ __launch_bounds__(256) void fp8_weights_amax_kernel(const long long* __restrict__ desc, const float* __restrict__ p, float* __restrict__ amax) {
  const long long* d = desc + blockIdx.y * 3;
  fp8_amax_body<float>(d[1], (int)d[2], p + d[0], d[2], amax + (long long)blockIdx.y * FP8_SLOTS);
}
template <int TR>
__global__ __launch_bounds__(256) void fp8_weights_quant_kernel(const long long* __restrict__ desc, const float* __restrict__ p, unsigned char* __restrict__ w8,
                                                                const float* __restrict__ amax, float* __restrict__ dq) {
  const long long* d = desc + blockIdx.y * 3;
  fp8_quant_body<float, 0, TR>(d[1], (int)d[2], p + d[0], d[2], w8 + d[0], TR ? d[1] : d[2], amax + (long long)blockIdx.y * FP8_SLOTS, dq + blockIdx.y, nullptr);
}
extern "C" int csmae_fp8_weights(int count, const long long* desc, const float* p, void* w8, void* w8t, float* amax, float* dq, void* stream) {
  CSMAE_REQUIRE(count > 0 && desc && p && w8 && w8t && amax && dq, "csmae_fp8_weights: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(fp8_weights_amax_kernel, dim3(64, count), dim3(256), 0, st, desc, p, amax);
  hipLaunchKernelGGL(fp8_weights_quant_kernel<0>, dim3(64, count), dim3(256), 0, st, desc, p, (unsigned char*)w8, amax, dq);
  hipLaunchKernelGGL(fp8_weights_quant_kernel<1>, dim3(128, count), dim3(256), 0, st, desc, p, (unsigned char*)w8t, amax, dq);
  return csmae_check_launch("csmae_fp8_weights");
}

extern "C" int csmae_fp8_amax(int in_dtype, long long rows, int cols, const void* src, long long ld, float* amax, void* stream) {
  CSMAE_REQUIRE(rows > 0 && cols > 0 && cols % 4 == 0 && ld % 4 == 0 && src && amax, "csmae_fp8_amax: bad arguments (cols, ld multiples of 4)");
  const dim3 grid((unsigned)fmin((double)rows, 1024.0));
  if (in_dtype == CSMAE_BF16) hipLaunchKernelGGL(fp8_amax_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, rows, cols, (const bf16_t*)src, ld, amax);
  else if (in_dtype == CSMAE_F32) hipLaunchKernelGGL(fp8_amax_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, rows, cols, (const float*)src, ld, amax);
  else { csmae_set_error("csmae_fp8_amax: bad dtype %d", in_dtype); return CSMAE_ERR_UNSUPPORTED; }
  return csmae_check_launch("csmae_fp8_amax");
}

extern "C" int csmae_fp8_quantize(int in_dtype, int fmt, int transpose, long long rows, int cols, const void* src, long long ld, void* dst,
                                  long long ldd, const float* amax, float* dq, float* amax_next, void* stream) {
  CSMAE_REQUIRE(rows > 0 && cols > 0 && cols % 4 == 0 && ld % 4 == 0 && ldd % 4 == 0 && src && dst && amax && dq && (fmt == 0 || fmt == 1),
                "csmae_fp8_quantize: bad arguments");
  CSMAE_REQUIRE(!transpose || rows % 4 == 0, "csmae_fp8_quantize: a transposed mirror needs rows %% 4 == 0");
  CSMAE_REQUIRE(!transpose || !amax_next, "csmae_fp8_quantize: delayed scaling is not wired for the transposed mirror");
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)fmin((double)(transpose ? ((rows + 63) / 64) * ((cols + 63) / 64) : rows), transpose ? 4096.0 : 1024.0));
#define QGO(T_, F_, R_) hipLaunchKernelGGL((fp8_quant_kernel<T_, F_, R_>), grid, dim3(256), 0, st, rows, cols, (const T_*)src, ld, (unsigned char*)dst, ldd, amax, dq, amax_next)
  if (in_dtype == CSMAE_BF16) { if (transpose) { if (fmt) QGO(bf16_t, 1, 1); else QGO(bf16_t, 0, 1); } else { if (fmt) QGO(bf16_t, 1, 0); else QGO(bf16_t, 0, 0); } }
  else if (in_dtype == CSMAE_F32) { if (transpose) { if (fmt) QGO(float, 1, 1); else QGO(float, 0, 1); } else { if (fmt) QGO(float, 1, 0); else QGO(float, 0, 0); } }
  else { csmae_set_error("csmae_fp8_quantize: bad dtype %d", in_dtype); return CSMAE_ERR_UNSUPPORTED; }
#undef QGO
  return csmae_check_launch("csmae_fp8_quantize");
}
