// Shared device/host helpers for libcsmae_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <math.h>

#define CSMAE_F32 0
#define CSMAE_BF16 1

#define CSMAE_OK 0
#define CSMAE_ERR_ARG -1
#define CSMAE_ERR_LAUNCH -2
#define CSMAE_ERR_UNSUPPORTED -3

typedef unsigned short bf16_t;  // raw bfloat16 bits
typedef short s4_t __attribute__((ext_vector_type(4)));
typedef short s8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf8_t __attribute__((ext_vector_type(8)));
typedef float f4_t __attribute__((ext_vector_type(4)));
#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

// two transposed 4x16-bit reads -> one 8x16-bit MFMA operand by register naming only (element-wise vector construction made
// hipcc emit v_perm/v_or shuffles per fragment)
__device__ __forceinline__ s8_t join_s4(s4_t lo, s4_t hi) {
  const uint2 l = __builtin_bit_cast(uint2, lo), h = __builtin_bit_cast(uint2, hi);
  return __builtin_bit_cast(s8_t, make_uint4(l.x, l.y, h.x, h.y));
}

// ---- one gate for the settled A/B aids: CSMAE_DEBUG="key[=value],key,..." (the Python side reads the same variable: csmae_hip.debug_opt).
// csmae_debug_opt returns the value of `key` ("1" for a bare key) or nullptr; host only, evaluated once per call site (static const).
const char* csmae_debug_opt(const char* key);
// ---- error plumbing (thread-local message; see include/csmae.h csmae_last_error)
void csmae_set_error(const char* fmt, ...);
int csmae_check_launch(const char* what);
// Completion event of the NEXT kernel launch of this host thread (csmae_next_launch_event).  A launch site that goes through CSMAE_LAUNCH
// attaches the event to its dispatch packet (hipExtLaunchKernelGGL's stopEvent: the packet's own completion signal) instead of leaving the
// caller to put a marker packet behind the kernel — a marker costs the stream it is recorded on 3-5 us (tools/event_probe.hip).
#include <hip/hip_ext.h>
extern thread_local hipEvent_t g_csmae_launch_event;
#define CSMAE_LAUNCH(kernel, grid, block, shmem, stream, ...)                                                        \
  do {                                                                                                               \
    hipEvent_t ev_ = g_csmae_launch_event;                                                                           \
    if (ev_) { g_csmae_launch_event = nullptr; hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, nullptr, ev_, 0, __VA_ARGS__); } \
    else hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                                        \
  } while (0)
#define CSMAE_REQUIRE(cond, ...)                     \
  do {                                               \
    if (!(cond)) {                                   \
      csmae_set_error(__VA_ARGS__);                  \
      return CSMAE_ERR_ARG;                          \
    }                                                \
  } while (0)

// ---- scalar conversions (round-to-nearest-even, NaN preserved)
__device__ __forceinline__ float bf2f(bf16_t b) { return __uint_as_float(((unsigned)b) << 16); }
typedef __bf16 bf2_t __attribute__((ext_vector_type(2)));
// native conversions: hipcc lowers them to gfx950's v_cvt_pk_bf16_f32 (round-to-nearest-even, NaN preserving) — one instruction
// per PAIR instead of the ~8-instruction integer sequence
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ unsigned pack2bf(float lo, float hi) { bf2_t v = {(__bf16)lo, (__bf16)hi}; return __builtin_bit_cast(unsigned, v); }

template <typename T> __device__ __forceinline__ float ld_as_f32(const T* p);
template <> __device__ __forceinline__ float ld_as_f32<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld_as_f32<bf16_t>(const bf16_t* p) { return bf2f(*p); }
template <typename T> __device__ __forceinline__ void st_from_f32(T* p, float v);
template <> __device__ __forceinline__ void st_from_f32<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st_from_f32<bf16_t>(bf16_t* p, float v) { *p = f2bf(v); }

// 4 consecutive elements (pointer must be 4-element aligned)
template <typename T> __device__ __forceinline__ f4_t ld4(const T* p);
template <> __device__ __forceinline__ f4_t ld4<float>(const float* p) { return *reinterpret_cast<const f4_t*>(p); }
template <> __device__ __forceinline__ f4_t ld4<bf16_t>(const bf16_t* p) {
  uint2 u = *reinterpret_cast<const uint2*>(p);
  f4_t r = {__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u)};
  return r;
}
template <typename T> __device__ __forceinline__ void st4(T* p, f4_t v);
template <> __device__ __forceinline__ void st4<float>(float* p, f4_t v) { *reinterpret_cast<f4_t*>(p) = v; }
template <> __device__ __forceinline__ void st4<bf16_t>(bf16_t* p, f4_t v) {
  uint2 u; u.x = pack2bf(v[0], v[1]); u.y = pack2bf(v[2], v[3]);
  *reinterpret_cast<uint2*>(p) = u;
}

template <typename T> __device__ __forceinline__ f4_t round4(f4_t v);
template <> __device__ __forceinline__ f4_t round4<float>(f4_t v) { return v; }
template <> __device__ __forceinline__ f4_t round4<bf16_t>(f4_t v) {
  return f4_t{bf2f(f2bf(v[0])), bf2f(f2bf(v[1])), bf2f(f2bf(v[2])), bf2f(f2bf(v[3]))};
}

// ---- wave64 reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// The same sum for a FULL wave (all 64 lanes active — the result is read from lane 63) on DPP moves only: quad swaps, half-row and row mirrors leave the row's
// sum in each of its 16 lanes, row_bcast:15 / :31 carry it across the four rows.  `wave_sum` above goes through ds_bpermute_b32 (the LDS crossbar): six
// dependent round trips of ~100 clocks per sum — more than a LayerNorm row's arithmetic.  Different summation tree: last-bit differences against wave_sum.
__device__ __forceinline__ float wave_sum64(float v) {
#define CSMAE_DPP_ADD(ctrl, rmask, bctl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rmask, 0xF, bctl))
  CSMAE_DPP_ADD(0xB1, 0xF, true);    // quad_perm [1,0,3,2]
  CSMAE_DPP_ADD(0x4E, 0xF, true);    // quad_perm [2,3,0,1]
  CSMAE_DPP_ADD(0x141, 0xF, true);   // row_half_mirror
  CSMAE_DPP_ADD(0x140, 0xF, true);   // row_mirror
  CSMAE_DPP_ADD(0x142, 0xA, false);  // row_bcast:15 into rows 1 and 3
  CSMAE_DPP_ADD(0x143, 0xC, false);  // row_bcast:31 into rows 2 and 3
#undef CSMAE_DPP_ADD
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// ... and the maximum of a FULL wave the same way (fp8 amax bookkeeping of the producing kernels' epilogues: every lane is active there)
__device__ __forceinline__ float wave_max64(float v) {
#define CSMAE_DPP_MAX(ctrl, rmask, bctl) v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), ctrl, rmask, 0xF, bctl)))
  CSMAE_DPP_MAX(0xB1, 0xF, false);
  CSMAE_DPP_MAX(0x4E, 0xF, false);
  CSMAE_DPP_MAX(0x141, 0xF, false);
  CSMAE_DPP_MAX(0x140, 0xF, false);
  CSMAE_DPP_MAX(0x142, 0xA, false);   // (rows outside the mask keep their own value: old = v)
  CSMAE_DPP_MAX(0x143, 0xC, false);
#undef CSMAE_DPP_MAX
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// max / sum over the four 16-lane rows of a wave, lane by lane (lanes t, t + 16, t + 32, t + 48: what `x = op(x, __shfl_xor(x, 16)); x = op(x, __shfl_xor(x, 32))`
// computes, bit for bit — the same pairs are combined) on gfx950's v_permlane16_swap / v_permlane32_swap instead of two dependent ds_bpermute round trips
// through the LDS crossbar: swapping a register with a copy of itself leaves every lane's partner value in one of the two.
__device__ __forceinline__ float xrow_max(float v) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xrow_sum(float v) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// block reduction of one float over blockDim.x threads (<=1024); result valid in every thread
__device__ __forceinline__ float block_sum(float v, float* smem /* >= 17 floats */) {
  v = wave_sum(v);
  int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) smem[w] = v;
  __syncthreads();
  if (threadIdx.x == 0) { float s = 0; for (int i = 0; i < nw; ++i) s += smem[i]; smem[16] = s; }
  __syncthreads();
  return smem[16];
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
  return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * __expf(-0.5f * x * x);
}

// Abramowitz-Stegun 7.1.26 erf (|error| <= 1.5e-7): one exp + one rcp instead of libm's ulp-exact erff.  Used by the bf16
// epilogues, where the result is rounded to 8 mantissa bits anyway; the fp32 parity path keeps erff.
__device__ __forceinline__ float erf_fast(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float r = 1.0f - poly * t * __expf(-ax * ax);
  return copysignf(r, x);
}
template <typename T> __device__ __forceinline__ float gelu_fwd(float x);
template <> __device__ __forceinline__ float gelu_fwd<float>(float x) { return gelu_erf(x); }
template <> __device__ __forceinline__ float gelu_fwd<bf16_t>(float x) { return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752440f)); }
template <typename T> __device__ __forceinline__ float gelu_bwd(float x);
template <> __device__ __forceinline__ float gelu_bwd<float>(float x) { return gelu_erf_grad(x); }
template <> __device__ __forceinline__ float gelu_bwd<bf16_t>(float x) {
  return 0.5f * (1.0f + erf_fast(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * __expf(-0.5f * x * x);
}

// GELU and its derivative from ONE exponential: e = exp(-x^2/2) feeds both erf (A&S 7.1.26: erfc(z) = t*poly(t)*exp(-z^2),
// z = |x|/sqrt2) and the Gaussian density.  The fc1 epilogue stores h = gelu(x) and g = gelu'(x); the fc2-backward epilogue is then
// a single multiply (it used to recompute erf + exp from the saved pre-activation).
template <typename T> __device__ __forceinline__ void gelu_both(float x, float& h, float& gp);
template <> __device__ __forceinline__ void gelu_both<bf16_t>(float x, float& h, float& gp) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.23164189f, ax, 1.0f));  // 0.3275911 / sqrt(2)
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-0.72134752f * x * x);          // exp(-x^2/2)
  const float q = 0.5f * poly * t * e;                                   // 0.5*erfc(|x|/sqrt2) = Phi(-|x|)
  const float phi = x >= 0.f ? 1.0f - q : q;                             // Phi(x)
  h = x * phi;
  gp = fmaf(x * 0.39894228f, e, phi);
}
template <> __device__ __forceinline__ void gelu_both<float>(float x, float& h, float& gp) { h = gelu_erf(x); gp = gelu_erf_grad(x); }
template <typename T> __device__ __forceinline__ void gelu_both4(f4_t x, f4_t& h, f4_t& gp) {
#pragma unroll
  for (int k = 0; k < 4; ++k) { float a, b; gelu_both<T>(x[k], a, b); h[k] = a; gp[k] = b; }
}
// bf16 outputs: the same formula on two-element vectors, so the multiplies and fmas become v_pk_mul_f32 / v_pk_fma_f32 (the GELU
// epilogue of fc1 is VALU-bound: ~22 op-equivalents per element scalar, ~15 packed)
typedef float f2_t __attribute__((ext_vector_type(2)));
// Phi(x) for bf16 outputs without the reciprocal and the sign select of the A&S form: Phi = 0.5 + xc * P(xc^2) with xc = clamp(x, +-3.5)
// and an odd minimax polynomial constrained to reach exactly +-0.5 at the clamp (Phi saturates at 0 / 1 beyond; max |Phi error| 2.3e-4,
// max |gelu error| 8e-4 at x = 3.5 where a bf16 ulp is 1.6e-2).  The fc1 epilogue is VALU-bound: per pair of elements this is
// 16 packed / scalar ops + 2 transcendentals (the Gaussian density of gelu') instead of 21 + 4.
__device__ __forceinline__ void gelu_both2_fast(f2_t x, f2_t& h, f2_t& gp) {
  const f2_t xc = {__builtin_amdgcn_fmed3f(x[0], -3.5f, 3.5f), __builtin_amdgcn_fmed3f(x[1], -3.5f, 3.5f)};
  const f2_t u = xc * xc;
  f2_t poly = u * -6.819443001e-07f + 3.422239752e-05f;
  poly = poly * u + -7.210712065e-04f;
  poly = poly * u + 8.507518098e-03f;
  poly = poly * u + -6.439825892e-02f;
  poly = poly * u + 3.980685472e-01f;
  const f2_t phi = xc * poly + 0.5f;                                     // Phi(x)
  const f2_t xx = x * x * -0.72134752f;
  const f2_t e = {__builtin_amdgcn_exp2f(xx[0]), __builtin_amdgcn_exp2f(xx[1])};   // exp(-x^2/2)
  h = x * phi;
  gp = x * 0.39894228f * e + phi;
}
template <> __device__ __forceinline__ void gelu_both4<bf16_t>(f4_t x, f4_t& h, f4_t& gp) {
  f2_t h0, g0, h1, g1;
  gelu_both2_fast(f2_t{x[0], x[1]}, h0, g0);
  gelu_both2_fast(f2_t{x[2], x[3]}, h1, g1);
  h = f4_t{h0[0], h0[1], h1[0], h1[1]};
  gp = f4_t{g0[0], g0[1], g1[0], g1[1]};
}

// fp8 mode, delayed scaling (csrc/fp8.hip, csmae_gemm_fp8): a kernel that produces a GEMM's A operand also writes it as fp8 bytes, scaled
// with the amax the tensor had one step earlier (64 partial maxima), and records the new amax — no separate quantisation pass.
struct Fp8Emit { unsigned char* q; const float* amax_prev; float* amax_next; float* dq; int fmt; };
__device__ __forceinline__ float fp8_emit_scale(const Fp8Emit& e, int lane, float& qmax) {
  qmax = e.fmt == 0 ? 448.0f : 57344.0f;
  const float am = wave_max64(e.amax_prev[lane & 63]);
  if (blockIdx.x == 0 && threadIdx.x == 0) e.dq[0] = am > 0.f ? am / qmax : 1.f;
  return am > 0.f ? qmax / am : 1.f;
}
// |x| as an unsigned integer: non-negative floats order like their bit patterns, +Inf (0x7f800000) sorts above every finite value and the NaN patterns above
// +Inf — so the running maximum of a tensor WITH its non-finite values is two integer ops per element (v_and + a share of v_max3_u32) instead of a compare, a
// select and a float max (the emitting epilogues are VALU-bound: csmae_gemm_fp8's GELU epilogue spends ~25 ops per element)
__device__ __forceinline__ unsigned abs_bits(float x) { return __float_as_uint(x) & 0x7fffffffu; }
__device__ __forceinline__ unsigned umax3(unsigned a, unsigned b, unsigned c) { return max(max(a, b), c); }
__device__ __forceinline__ float amax_of_bits(unsigned s) { return s > 0x7f800000u ? INFINITY : __uint_as_float(s); }   // (NaN records +Inf, like Inf)
__device__ __forceinline__ unsigned fp8_pack4(f4_t v, float scale, float qmax, int fmt, float& seen) {
  int p = 0;
  // non-finite values are not hidden: NaN passes through the clamp, NaN / Inf record an amax of +Inf (next step: scale 0 x Inf = NaN -> the loss gate trips)
  const unsigned m = max(umax3(abs_bits(v[0]), abs_bits(v[1]), abs_bits(v[2])), abs_bits(v[3]));
  seen = fmaxf(seen, amax_of_bits(m));
  const f4_t q = v * scale;
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = __builtin_amdgcn_fmed3f(q[k], -qmax, qmax);
  if (__builtin_amdgcn_ballot_w64(m >= 0x7f800000u) != 0ull) {   // (rare, wave-uniform branch: an Inf or a NaN somewhere in the wave — v_med3 would turn a NaN, also the NaN of Inf x 0, into -qmax)
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = q[k] != q[k] ? q[k] : v[k];
  }
  if (fmt == 0) { p = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], p, false); p = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], p, true); }
  else { p = __builtin_amdgcn_cvt_pk_bf8_f32(v[0], v[1], p, false); p = __builtin_amdgcn_cvt_pk_bf8_f32(v[2], v[3], p, true); }
  return (unsigned)p;
}
// Publish a partial maximum (non-negative floats order like their bit patterns).  The atomics of a launch onto one slot are serialised at the memory side
// (~0.3 us each: 8 192 waves on 64 slots added 36 us to a LayerNorm launch of 10), so a wave first LOOKS (a device-scope load, not serialised) and only
// publishes a value that raises the slot: a handful per slot instead of every wave's.
__device__ __forceinline__ void amax_publish(float* slot, float seen) {
  if (seen > 0.f) {
    const unsigned cur = __hip_atomic_load(reinterpret_cast<unsigned*>(slot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__float_as_uint(seen) > cur) atomicMax(reinterpret_cast<unsigned*>(slot), __float_as_uint(seen));
  }
}
__device__ __forceinline__ void fp8_emit_amax(const Fp8Emit& e, float seen, int lane) {
  seen = wave_max64(seen);
  if (lane == 0) amax_publish(e.amax_next + ((blockIdx.x * 4 + (threadIdx.x >> 6)) & 63), seen);
}


static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
