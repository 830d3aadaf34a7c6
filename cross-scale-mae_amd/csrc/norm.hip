// LayerNorm (timm Block norm1/norm2, decoder_norm; eps 1e-6 — MAE_ViT_Baseline.py:43-45) and the predictor's
// BatchNorm1d-over-token-positions + ReLU (models_mae/MLP.py:4-10).  All HBM-bound: one wave per LayerNorm row
// (wave-shuffle reductions, 16-B accesses), one workgroup per BatchNorm channel.
#include "common.h"
#include <type_traits>
typedef __amdgpu_buffer_rsrc_t brsrc_t;
typedef unsigned int u2_t __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------------------ LayerNorm forward
template <typename TX, typename TO, int NV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(long long M, int D, const TX* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float eps, TO* __restrict__ y, float* __restrict__ y32,
                                                     float* __restrict__ mean, float* __restrict__ rstd, Fp8Emit em) {
  const int lane = threadIdx.x & 63;
  float qmax = 0.f, qseen = 0.f;
  const float qscale = em.q ? fp8_emit_scale(em, lane, qmax) : 1.f;
  const int nv = D >> 2;
  // one row per wave and pass; the grid covers all rows in one pass unless an fp8 copy is emitted: then a wave walks several rows, so that
  // reading the scale (a dependent global load + wave reduction) and the amax atomic are paid once per wave, not once per row
  for (long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); row < M; row += (long long)gridDim.x * 4) {
    f4_t v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int c = lane + i * 64;
      v[i] = c < nv ? ld4<TX>(x + row * D + c * 4) : f4_t{0.f, 0.f, 0.f, 0.f};
      s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    }
    const float mu = wave_sum64(s) / D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int c = lane + i * 64;
      if (c < nv) { f4_t d = v[i] - mu; q += d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]; }
    }
    const float rs = rsqrtf(wave_sum64(q) / D + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int c = lane + i * 64;
      if (c < nv) {
        f4_t gm = *reinterpret_cast<const f4_t*>(gamma + c * 4), bt = *reinterpret_cast<const f4_t*>(beta + c * 4);
        f4_t o = (v[i] - mu) * rs * gm + bt;
        if (y) st4<TO>(y + row * D + c * 4, o);   // (y == null: only the fp8 copy leaves — fp8 mode, where nothing reads the bf16 output)
        if (y32) *reinterpret_cast<f4_t*>(y32 + row * D + c * 4) = o;
        if (em.q) *reinterpret_cast<unsigned*>(em.q + row * D + c * 4) = fp8_pack4(o, qscale, qmax, em.fmt, qseen);
      }
    }
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
  }
  if (em.q) fp8_emit_amax(em, qseen, lane);
}

// ------------------------------------------------------------------------------------------ LayerNorm backward
// dx = rstd * (g - mean(g) - xhat * mean(g*xhat)),  g = dy*gamma ;  dres_out = dres_in + dx ;  dgamma += dy*xhat ; dbeta += dy
// TX = type of the residual stream: x, and the residual-gradient stream dres_in / dx_out (fp32, or bf16 in throughput mode, where
// dx_out itself is the next GEMM's operand and dx_lp is null)
template <typename TDY, typename TX, typename TLP, int NV>   // (occupancy steps: NV = 3, D = 768, held to 128 registers = four waves per SIMD — at 130, three, every encoder launch ran 20 % longer —; NV = 2, D = 512, to 96 = five)
__global__ __launch_bounds__(256, NV == 3 ? 4 : (NV == 2 ? 5 : 1)) void ln_bwd_kernel(long long M, int D, const TDY* __restrict__ dy, const TX* __restrict__ x,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, const TX* __restrict__ dres_in,
                                                     TX* __restrict__ dx_out, TLP* __restrict__ dx_lp,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ part, Fp8Emit em) {
  __shared__ float red[4 * 64 * 4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float qmax = 0.f, qseen = 0.f;
  const float qscale = em.q ? fp8_emit_scale(em, lane, qmax) : 1.f;
  const int nv = D >> 2;
  f4_t gm[NV], ag[NV], ab[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    int c = lane + i * 64;
    gm[i] = c < nv ? *reinterpret_cast<const f4_t*>(gamma + c * 4) : f4_t{0.f, 0.f, 0.f, 0.f};
    ag[i] = f4_t{0.f, 0.f, 0.f, 0.f}; ab[i] = f4_t{0.f, 0.f, 0.f, 0.f};
  }
  for (long long row = (long long)blockIdx.x * 4 + w; row < M; row += (long long)gridDim.x * 4) {
    const float mu = mean[row], rs = rstd[row];
    f4_t g[NV], xh[NV], dr[NV];
    float s1 = 0.f, s2 = 0.f;
    // every load of the row is issued before the first reduction: the residual gradient does not depend on the row statistics,
    // and fetching it only after the two wave reductions left each wave with a second, serialised HBM round trip per row
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int c = lane + i * 64;
      dr[i] = (dres_in && c < nv) ? ld4<TX>(dres_in + row * D + c * 4) : f4_t{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int c = lane + i * 64;
      if (c < nv) {
        f4_t d = ld4<TDY>(dy + row * D + c * 4);
        xh[i] = (ld4<TX>(x + row * D + c * 4) - mu) * rs;
        ag[i] += d * xh[i]; ab[i] += d;
        g[i] = d * gm[i];
        s1 += g[i][0] + g[i][1] + g[i][2] + g[i][3];
        f4_t t = g[i] * xh[i];
        s2 += t[0] + t[1] + t[2] + t[3];
      }
    }
    s1 = wave_sum64(s1) / D; s2 = wave_sum64(s2) / D;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int c = lane + i * 64;
      if (c < nv) {
        f4_t dx = (g[i] - s1 - xh[i] * s2) * rs + dr[i];
        st4<TX>(dx_out + row * D + c * 4, dx);
        if (dx_lp) st4<TLP>(dx_lp + row * D + c * 4, dx);
        if (em.q) *reinterpret_cast<unsigned*>(em.q + row * D + c * 4) = fp8_pack4(dx, qscale, qmax, em.fmt, qseen);
      }
    }
  }
  if (em.q) fp8_emit_amax(em, qseen, lane);
  if (!dgamma && !part) return;
  // fold the 4 waves' column partials into this block's partial row (or, without a workspace, one atomic per column per block)
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    int c = lane + i * 64;
    __syncthreads();
    *reinterpret_cast<f4_t*>(red + (w * 64 + lane) * 4) = ag[i];
    __syncthreads();
    if (w == 0 && c < nv) {
      f4_t s = *reinterpret_cast<f4_t*>(red + lane * 4) + *reinterpret_cast<f4_t*>(red + (64 + lane) * 4) +
               *reinterpret_cast<f4_t*>(red + (128 + lane) * 4) + *reinterpret_cast<f4_t*>(red + (192 + lane) * 4);
      if (part) *reinterpret_cast<f4_t*>(part + (long long)blockIdx.x * 2 * D + c * 4) = s;
      else for (int k = 0; k < 4; ++k) unsafeAtomicAdd(dgamma + c * 4 + k, s[k]);
    }
    __syncthreads();
    *reinterpret_cast<f4_t*>(red + (w * 64 + lane) * 4) = ab[i];
    __syncthreads();
    if (w == 0 && c < nv) {
      f4_t s = *reinterpret_cast<f4_t*>(red + lane * 4) + *reinterpret_cast<f4_t*>(red + (64 + lane) * 4) +
               *reinterpret_cast<f4_t*>(red + (128 + lane) * 4) + *reinterpret_cast<f4_t*>(red + (192 + lane) * 4);
      if (part) *reinterpret_cast<f4_t*>(part + (long long)blockIdx.x * 2 * D + D + c * 4) = s;
      else for (int k = 0; k < 4; ++k) unsafeAtomicAdd(dbeta + c * 4 + k, s[k]);
    }
  }
}

// The throughput mode's instance (dy, x, dres_in, dx_out all bf16, no second copy) holding a row's three inputs PACKED (8 bytes per lane and load) through
// the two reductions and unpacking them at each use: 54 instead of 90 live registers at D = 768, i.e. six instead of four waves per SIMD — a wave handles one
// row at a time and is bound by its row's HBM round trip, so rows in flight per CU are what the kernel's bandwidth consists of (3.7 TB/s alone at four waves).
template <int NV, int MINW, bool EMIT>
__global__ __launch_bounds__(256, MINW) void ln_bwd_bf16_kernel(long long M, int D, const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, const float* __restrict__ mean,
                                                                const float* __restrict__ rstd, const float* __restrict__ gamma, const bf16_t* __restrict__ dres_in,
                                                                bf16_t* __restrict__ dx_out, float* __restrict__ part, Fp8Emit em) {
  __shared__ float red[4 * 64 * 4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int nv = D >> 2;
  // EMIT (fp8 mode): the outgoing gradient also leaves as fp8 bytes (the next product's operand), scaled with the previous step's maximum (common.h Fp8Emit)
  float qmax = 0.f, qseen = 0.f;
  const float qscale = EMIT ? fp8_emit_scale(em, lane, qmax) : 1.f;
  f4_t gm[NV], ag[NV], ab[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + i * 64;
    gm[i] = c < nv ? *reinterpret_cast<const f4_t*>(gamma + c * 4) : f4_t{0.f, 0.f, 0.f, 0.f};
    ag[i] = f4_t{0.f, 0.f, 0.f, 0.f}; ab[i] = f4_t{0.f, 0.f, 0.f, 0.f};
  }
  auto up = [](u2_t u) { return f4_t{__uint_as_float(u[0] << 16), __uint_as_float(u[0] & 0xffff0000u), __uint_as_float(u[1] << 16), __uint_as_float(u[1] & 0xffff0000u)}; };
  // the four [M][D] bf16 tensors share one element offset: buffer resources + ONE 32-bit byte offset per row (lane's first column group; group i is 512 B further)
  // instead of a 64-bit address per load (18 of the first version's registers); columns beyond D are masked to an out-of-range offset (zeros in, nothing out)
  const long long bytes = M * (long long)D * 2;
  const brsrc_t rD = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(dy), 0, (int)(unsigned)bytes, 0x00020000);
  const brsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(x), 0, (int)(unsigned)bytes, 0x00020000);
  const brsrc_t rR = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(dres_in ? dres_in : dy), 0, dres_in ? (int)(unsigned)bytes : 0, 0x00020000);
  const brsrc_t rO = __builtin_amdgcn_make_buffer_rsrc(dx_out, 0, (int)(unsigned)bytes, 0x00020000);
  const brsrc_t rQ = __builtin_amdgcn_make_buffer_rsrc(EMIT ? em.q : (unsigned char*)dx_out, 0, EMIT ? (int)(unsigned)(bytes >> 1) : 0, 0x00020000);
  for (long long row = (long long)blockIdx.x * 4 + w; row < M; row += (long long)gridDim.x * 4) {
    const float mu = mean[row], rs = rstd[row];
    const unsigned voff = (unsigned)((row * D + lane * 4) * 2);
    u2_t rd[NV], rx[NV], rr[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const unsigned o = (lane + i * 64 < nv) ? voff + (unsigned)(i * 512) : 0xFFFFFFF0u;
      rr[i] = __builtin_bit_cast(u2_t, __builtin_amdgcn_raw_buffer_load_b64(rR, o, 0, 0));
      rd[i] = __builtin_bit_cast(u2_t, __builtin_amdgcn_raw_buffer_load_b64(rD, o, 0, 0));
      rx[i] = __builtin_bit_cast(u2_t, __builtin_amdgcn_raw_buffer_load_b64(rX, o, 0, 0));
    }
    float s1 = 0.f, s2 = 0.f;
    const float c0 = -mu * rs;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (lane + i * 64 < nv) {
        const f4_t d = up(rd[i]), xh = up(rx[i]) * rs + c0, g = d * gm[i];
        ag[i] += d * xh; ab[i] += d;
        s1 += (g[0] + g[1]) + (g[2] + g[3]);
        const f4_t t = g * xh;
        s2 += (t[0] + t[1]) + (t[2] + t[3]);
      }
      __builtin_amdgcn_sched_barrier(0);   // (one column group at a time: unpacked, all groups at once are the 36 registers this instance exists to save)
    }
    s1 = wave_sum64(s1) / D; s2 = wave_sum64(s2) / D;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const unsigned o = (lane + i * 64 < nv) ? voff + (unsigned)(i * 512) : 0xFFFFFFF0u;
      const f4_t xh = up(rx[i]) * rs + c0, g = up(rd[i]) * gm[i];
      const f4_t o4 = (g - s1 - xh * s2) * rs + up(rr[i]);
      __builtin_amdgcn_raw_buffer_store_b64(u2_t{pack2bf(o4[0], o4[1]), pack2bf(o4[2], o4[3])}, rO, o, 0, 0);
      if (EMIT) __builtin_amdgcn_raw_buffer_store_b32(fp8_pack4(lane + i * 64 < nv ? o4 : f4_t{0.f, 0.f, 0.f, 0.f}, qscale, qmax, em.fmt, qseen), rQ, o >> 1, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (EMIT) fp8_emit_amax(em, qseen, lane);
  if (!part) return;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + i * 64;
    __syncthreads();
    *reinterpret_cast<f4_t*>(red + (w * 64 + lane) * 4) = ag[i];
    __syncthreads();
    if (w == 0 && c < nv)
      *reinterpret_cast<f4_t*>(part + (long long)blockIdx.x * 2 * D + c * 4) = *reinterpret_cast<f4_t*>(red + lane * 4) + *reinterpret_cast<f4_t*>(red + (64 + lane) * 4) +
                                                                             *reinterpret_cast<f4_t*>(red + (128 + lane) * 4) + *reinterpret_cast<f4_t*>(red + (192 + lane) * 4);
    __syncthreads();
    *reinterpret_cast<f4_t*>(red + (w * 64 + lane) * 4) = ab[i];
    __syncthreads();
    if (w == 0 && c < nv)
      *reinterpret_cast<f4_t*>(part + (long long)blockIdx.x * 2 * D + D + c * 4) = *reinterpret_cast<f4_t*>(red + lane * 4) + *reinterpret_cast<f4_t*>(red + (64 + lane) * 4) +
                                                                                 *reinterpret_cast<f4_t*>(red + (128 + lane) * 4) + *reinterpret_cast<f4_t*>(red + (192 + lane) * 4);
  }
}

// dgamma/dbeta: per-block partial rows -> column sums.  One workgroup owns 64 columns of one LayerNorm and adds its sum to the gradient
// in a fixed order (no atomics: the result is bit-reproducible).  blockIdx.y walks a batch of LayerNorms whose partial rows lie
// `stride` floats apart and whose dgamma / dbeta sit at goff[2k], goff[2k+1] floats behind `gbase` (the flat gradient buffer), so the
// whole backward pass needs one launch per block stack instead of one per LayerNorm on its critical path.
__global__ __launch_bounds__(1024) void ln_param_reduce_kernel(int nblk, int D, const float* __restrict__ part, long long stride,
                                                               float* __restrict__ gbase, const long long* __restrict__ goff,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta) {
  // 64 columns x 16 row lanes; a lane walks rows ty, ty + 16, ... with four loads in flight (a serial walk of 256 dependent rows per
  // thread made this kernel 120 us long), the lanes are folded through LDS in lane order
  __shared__ float red[16][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  const int k = blockIdx.y;
  part += (long long)k * stride;
  if (goff) { dgamma = gbase + goff[2 * k]; dbeta = gbase + goff[2 * k + 1]; }
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < 2 * D) {
    int b = ty;
    for (; b + 48 < nblk; b += 64) {
      s0 += part[(long long)b * 2 * D + c]; s1 += part[(long long)(b + 16) * 2 * D + c];
      s2 += part[(long long)(b + 32) * 2 * D + c]; s3 += part[(long long)(b + 48) * 2 * D + c];
    }
    for (; b < nblk; b += 16) s0 += part[(long long)b * 2 * D + c];
  }
  red[ty][tx] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (ty == 0 && c < 2 * D) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += red[j][tx];
    float* dst = c < D ? dgamma + c : dbeta + (c - D);
    *dst += s;
  }
}

template <typename TX, typename TO>
static int ln_fwd_launch(long long M, int D, const void* x, const float* g, const float* b, float eps, void* y, float* y32, float* mean, float* rstd, Fp8Emit em, hipStream_t st) {
  dim3 grid(em.q ? (unsigned)fmin((double)cdiv(M, 4), 2048.0) : (unsigned)cdiv(M, 4)), block(256);
  int nv = cdiv(D, 256);
#define LNF(NVV) hipLaunchKernelGGL((ln_fwd_kernel<TX, TO, NVV>), grid, block, 0, st, M, D, (const TX*)x, g, b, eps, (TO*)y, y32, mean, rstd, em)
  switch (nv) { case 1: LNF(1); break; case 2: LNF(2); break; case 3: LNF(3); break; case 4: LNF(4); break; case 5: LNF(5); break; default: LNF(8); }
#undef LNF
  return CSMAE_OK;
}

static int fp8_emit_check(const char* who, void* q, int fmt, const float* prev, float* next, float* dq) {
  CSMAE_REQUIRE(!q || (prev && next && dq && (fmt == 0 || fmt == 1) && ((uintptr_t)q & 3) == 0), "%s: the fused fp8 copy needs amax_prev, amax_next, dq and fmt 0 / 1", who);
  return CSMAE_OK;
}
extern "C" int csmae_layernorm_fwd(int x_dtype, int out_dtype, long long M, int D, const void* x, const float* gamma, const float* beta, float eps,
                                   void* y, float* y32, float* mean, float* rstd, void* q_out, int q_fmt, const float* q_amax_prev,
                                   float* q_amax_next, float* q_dq, void* stream) {
  CSMAE_REQUIRE(M > 0 && D > 0 && D % 4 == 0 && D <= 2048, "csmae_layernorm_fwd: need 0 < D <= 2048, D %% 4 == 0 (D=%d)", D);
  if (int rc = fp8_emit_check("csmae_layernorm_fwd", q_out, q_fmt, q_amax_prev, q_amax_next, q_dq)) return rc;
  CSMAE_REQUIRE(x && mean && rstd && (y || q_out), "csmae_layernorm_fwd: null argument (y may be null only when the fp8 copy is asked for)");
  const Fp8Emit em{(unsigned char*)q_out, q_amax_prev, q_amax_next, q_dq, q_fmt};
  hipStream_t st = (hipStream_t)stream;
  if (x_dtype == CSMAE_F32 && out_dtype == CSMAE_BF16) ln_fwd_launch<float, bf16_t>(M, D, x, gamma, beta, eps, y, y32, mean, rstd, em, st);
  else if (x_dtype == CSMAE_F32 && out_dtype == CSMAE_F32) ln_fwd_launch<float, float>(M, D, x, gamma, beta, eps, y, y32, mean, rstd, em, st);
  else if (x_dtype == CSMAE_BF16 && out_dtype == CSMAE_BF16) ln_fwd_launch<bf16_t, bf16_t>(M, D, x, gamma, beta, eps, y, y32, mean, rstd, em, st);
  else { csmae_set_error("csmae_layernorm_fwd: bad dtypes %d -> %d", x_dtype, out_dtype); return CSMAE_ERR_UNSUPPORTED; }
  return csmae_check_launch("csmae_layernorm_fwd");
}

// partial rows a launch leaves in its workspace (the caller of the deferred reduce needs the same number)
static int ln_bwd_blocks(long long M, int D, long long part_elems) {
  int blocks = (int)fmin((double)cdiv(M, 4), 1024.0);
  if (part_elems / (2 * D) < blocks) blocks = (int)(part_elems / (2 * D));
  return blocks;
}

template <typename TDY, typename TX, typename TLP>
static void ln_bwd_launch(long long M, int D, const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                          const void* dres_in, void* dx_out, void* dx_lp, float* dgamma, float* dbeta, float* part, long long part_elems,
                          Fp8Emit em, hipStream_t st) {
  int blocks = (int)fmin((double)cdiv(M, 4), 1024.0);
  if (part) {
    blocks = ln_bwd_blocks(M, D, part_elems);
    if (blocks < 1) part = nullptr, blocks = (int)fmin((double)cdiv(M, 4), 1024.0);
  }
  dim3 grid(blocks), block(256);
  int nv = cdiv(D, 256);
#define LNB(NVV) hipLaunchKernelGGL((ln_bwd_kernel<TDY, TX, TLP, NVV>), grid, block, 0, st, M, D, (const TDY*)dy, (const TX*)x, mean, rstd, gamma, (const TX*)dres_in, (TX*)dx_out, (TLP*)dx_lp, dgamma, dbeta, part, em)
  // the all-bf16 stream with deferred parameter gradients and no fp8 copy (every block LayerNorm of the bf16 step): the packed-register instance
  static const bool no_packed = csmae_debug_opt("ln_bwd_unpacked") != nullptr;   // A/B aid
  if (std::is_same<TDY, bf16_t>::value && std::is_same<TX, bf16_t>::value && !dx_lp && !dgamma && part && !no_packed && D % 4 == 0 && nv >= 2 && nv <= 5 &&
      (M + 4) * (long long)D * 2 < 0xFFFFFFF0ll) {
#define LNBP(NVV, MW, EM) hipLaunchKernelGGL((ln_bwd_bf16_kernel<NVV, MW, EM>), grid, block, 0, st, M, D, (const bf16_t*)dy, (const bf16_t*)x, mean, rstd, gamma, (const bf16_t*)dres_in, (bf16_t*)dx_out, part, em)
    if (em.q) { if (nv == 2) LNBP(2, 6, true); else if (nv == 3) LNBP(3, 5, true); else if (nv == 4) LNBP(4, 4, true); else LNBP(5, 3, true); }
    else { if (nv == 2) LNBP(2, 6, false); else if (nv == 3) LNBP(3, 5, false); else if (nv == 4) LNBP(4, 4, false); else LNBP(5, 3, false); }
#undef LNBP
    return;
  }
  switch (nv) { case 1: LNB(1); break; case 2: LNB(2); break; case 3: LNB(3); break; case 4: LNB(4); break; case 5: LNB(5); break; default: LNB(8); }
#undef LNB
  if (part && dgamma) hipLaunchKernelGGL(ln_param_reduce_kernel, dim3(cdiv(2 * D, 64), 1), dim3(1024), 0, st, blocks, D, part, 0ll, (float*)nullptr, (const long long*)nullptr, dgamma, dbeta);
}

extern "C" int csmae_layernorm_bwd(int dy_dtype, int x_dtype, int lp_dtype, long long M, int D, const void* dy, const void* x, const float* mean,
                                   const float* rstd, const float* gamma, const void* dres_in, void* dx_out, void* dx_lp,
                                   float* dgamma, float* dbeta, float* partial_ws, long long partial_elems, void* q_out, int q_fmt,
                                   const float* q_amax_prev, float* q_amax_next, float* q_dq, void* stream) {
  CSMAE_REQUIRE(M > 0 && D > 0 && D % 4 == 0 && D <= 2048, "csmae_layernorm_bwd: need 0 < D <= 2048, D %% 4 == 0 (D=%d)", D);
  if (int rc = fp8_emit_check("csmae_layernorm_bwd", q_out, q_fmt, q_amax_prev, q_amax_next, q_dq)) return rc;
  const Fp8Emit em{(unsigned char*)q_out, q_amax_prev, q_amax_next, q_dq, q_fmt};
  CSMAE_REQUIRE(dgamma || !partial_ws || partial_elems >= 2ll * D, "csmae_layernorm_bwd: deferred parameter gradients need a workspace of at least one partial row");
  hipStream_t st = (hipStream_t)stream;
#define GO(A, X, L) ln_bwd_launch<A, X, L>(M, D, dy, x, mean, rstd, gamma, dres_in, dx_out, dx_lp, dgamma, dbeta, partial_ws, partial_elems, em, st)
  if (x_dtype == CSMAE_BF16) {       // bf16 residual streams (throughput mode): dx_out is the GEMM operand itself
    if (dy_dtype == CSMAE_BF16) GO(bf16_t, bf16_t, bf16_t);
    else if (dy_dtype == CSMAE_F32) GO(float, bf16_t, bf16_t);
    else { csmae_set_error("csmae_layernorm_bwd: bad dtypes %d/%d/%d", dy_dtype, x_dtype, lp_dtype); return CSMAE_ERR_UNSUPPORTED; }
  } else if (x_dtype != CSMAE_F32) { csmae_set_error("csmae_layernorm_bwd: bad x dtype %d", x_dtype); return CSMAE_ERR_UNSUPPORTED; }
  else if (dy_dtype == CSMAE_BF16 && lp_dtype == CSMAE_BF16) GO(bf16_t, float, bf16_t);
  else if (dy_dtype == CSMAE_F32 && lp_dtype == CSMAE_BF16) GO(float, float, bf16_t);
  else if (dy_dtype == CSMAE_F32 && lp_dtype == CSMAE_F32) GO(float, float, float);
  else if (dy_dtype == CSMAE_BF16 && lp_dtype == CSMAE_F32) GO(bf16_t, float, float);
  else { csmae_set_error("csmae_layernorm_bwd: bad dtypes %d/%d", dy_dtype, lp_dtype); return CSMAE_ERR_UNSUPPORTED; }
#undef GO
  return csmae_check_launch("csmae_layernorm_bwd");
}

// Deferred dgamma / dbeta of a batch of LayerNorm backward launches that were called with dgamma == NULL and their own workspace
// slice: LayerNorm k's partial rows start at partials + k * stride (M, D and the slice size must be those of the launches).
extern "C" int csmae_ln_param_reduce(int count, long long M, int D, const float* partials, long long stride, long long slice_elems,
                                     float* gbase, const long long* goff, void* stream) {
  CSMAE_REQUIRE(count > 0 && M > 0 && D > 0 && partials && gbase && goff && slice_elems >= 2ll * D, "csmae_ln_param_reduce: bad arguments");
  const int blocks = ln_bwd_blocks(M, D, slice_elems);
  hipLaunchKernelGGL(ln_param_reduce_kernel, dim3(cdiv(2 * D, 64), count), dim3(1024), 0, (hipStream_t)stream, blocks, D, partials, stride, gbase, goff,
                     (float*)nullptr, (float*)nullptr);
  return csmae_check_launch("csmae_ln_param_reduce");
}

// The same fold for partial rows that a GEMM epilogue left (csmae_gemm_ln_bwd: one row per 128-row tile): the caller names the row count.
extern "C" int csmae_ln_param_reduce_rows(int count, int rows, int D, const float* partials, long long stride, float* gbase, const long long* goff, void* stream) {
  CSMAE_REQUIRE(count > 0 && rows > 0 && D > 0 && partials && gbase && goff && stride >= 2ll * D * rows, "csmae_ln_param_reduce_rows: bad arguments");
  hipLaunchKernelGGL(ln_param_reduce_kernel, dim3(cdiv(2 * D, 64), count), dim3(1024), 0, (hipStream_t)stream, rows, D, partials, stride, gbase, goff,
                     (float*)nullptr, (float*)nullptr);
  return csmae_check_launch("csmae_ln_param_reduce_rows");
}

// ------------------------------------------------------------------------------------------ BatchNorm(token axis)+ReLU
// u is [N*L, Hp]; channel = token position l; statistics over the N*Hp values {u[n*L + l, :]} (models_mae/MLP.py:7).
template <typename T>
__global__ __launch_bounds__(1024) void bnrelu_fwd_kernel(int N, int L, int Hp, const T* __restrict__ u, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float eps, float momentum, T* __restrict__ r,
                                                         float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ run_mean,
                                                         float* __restrict__ run_var, long long* __restrict__ nbt, int training) {
  __shared__ float red[32];
  const int l = blockIdx.x, hv = Hp >> 2;
  const long long cnt = (long long)N * Hp;
  if (!training) {  // eval(): normalise with the running statistics, touch nothing
    const float mu = run_mean[l], rs = rsqrtf(run_var[l] + eps), gm = gamma[l], bt = beta[l];
    for (long long e = threadIdx.x; e < (long long)N * hv; e += blockDim.x) {
      long long n = e / hv; int c = (int)(e - n * hv);
      f4_t v = (ld4<T>(u + (n * L + l) * Hp + c * 4) - mu) * rs * gm + bt;
      v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
      st4<T>(r + (n * L + l) * Hp + c * 4, v);
    }
    if (threadIdx.x == 0) { mean[l] = mu; rstd[l] = rs; }
    return;
  }
  float s = 0.f;
  for (long long e = threadIdx.x; e < (long long)N * hv; e += blockDim.x) {
    long long n = e / hv; int c = (int)(e - n * hv);
    f4_t v = ld4<T>(u + (n * L + l) * Hp + c * 4);
    s += v[0] + v[1] + v[2] + v[3];
  }
  const float mu = block_sum(s, red) / cnt;
  float q = 0.f;
  for (long long e = threadIdx.x; e < (long long)N * hv; e += blockDim.x) {
    long long n = e / hv; int c = (int)(e - n * hv);
    f4_t v = ld4<T>(u + (n * L + l) * Hp + c * 4) - mu;
    q += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  const float var = block_sum(q, red) / cnt;
  const float rs = rsqrtf(var + eps), gm = gamma[l], bt = beta[l];
  for (long long e = threadIdx.x; e < (long long)N * hv; e += blockDim.x) {
    long long n = e / hv; int c = (int)(e - n * hv);
    f4_t v = (ld4<T>(u + (n * L + l) * Hp + c * 4) - mu) * rs * gm + bt;
    v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
    st4<T>(r + (n * L + l) * Hp + c * 4, v);
  }
  if (threadIdx.x == 0) {
    mean[l] = mu; rstd[l] = rs;
    if (run_mean) {
      run_mean[l] = (1.f - momentum) * run_mean[l] + momentum * mu;
      run_var[l] = (1.f - momentum) * run_var[l] + momentum * var * ((float)cnt / (float)(cnt - 1));
    }
    if (nbt && l == 0) *nbt += 1;
  }
}

template <typename T>
__global__ __launch_bounds__(1024) void bnrelu_bwd_kernel(int N, int L, int Hp, const T* __restrict__ u, const T* __restrict__ dr,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd, T* __restrict__ du,
                                                         float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ float red[32];
  const int l = blockIdx.x, hv = Hp >> 2;
  const long long cnt = (long long)N * Hp;
  const float mu = mean[l], rs = rstd[l], gm = gamma[l], bt = beta[l];
  float s1 = 0.f, s2 = 0.f;
  for (long long e = threadIdx.x; e < (long long)N * hv; e += blockDim.x) {
    long long n = e / hv; int c = (int)(e - n * hv);
    long long off = (n * L + l) * Hp + c * 4;
    f4_t xh = (ld4<T>(u + off) - mu) * rs, g = ld4<T>(dr + off);
#pragma unroll
    for (int k = 0; k < 4; ++k) { float gg = (xh[k] * gm + bt) > 0.f ? g[k] : 0.f; s1 += gg; s2 += gg * xh[k]; }
  }
  s1 = block_sum(s1, red); s2 = block_sum(s2, red);
  const float m1 = s1 / cnt, m2 = s2 / cnt;
  for (long long e = threadIdx.x; e < (long long)N * hv; e += blockDim.x) {
    long long n = e / hv; int c = (int)(e - n * hv);
    long long off = (n * L + l) * Hp + c * 4;
    f4_t xh = (ld4<T>(u + off) - mu) * rs, g = ld4<T>(dr + off), o;
#pragma unroll
    for (int k = 0; k < 4; ++k) { float gg = (xh[k] * gm + bt) > 0.f ? g[k] : 0.f; o[k] = gm * rs * (gg - m1 - xh[k] * m2); }
    st4<T>(du + off, o);
  }
  if (threadIdx.x == 0) { dgamma[l] += s2; dbeta[l] += s1; }
}

// ---- bf16 fast paths of the two kernels above (Hp / 8 divides 1024: the predictor's 2048, the test sizes).  The generic kernels walk
// the channel's N x Hp values with 8-byte loads, a 64-bit division per element and one load in flight per thread: 100 / 150 us for
// 100 MB.  Here thread (cq, rl) owns the 16-byte column group cq of rows rl, rl + RL, ...: four independent 16-byte loads in flight, no
// divisions, and the statistics in ONE pass as shifted sums (shift = the channel's first value, so the subtraction of the two moments
// does not cancel): forward 3 -> 2 sweeps over the channel, same two-pass-equivalent arithmetic to ~1e-7 relative.
__device__ __forceinline__ void bf8_unpack(const uint4& a, float (&v)[8]) {
  v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u); v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
  v[4] = __uint_as_float(a.z << 16); v[5] = __uint_as_float(a.z & 0xffff0000u); v[6] = __uint_as_float(a.w << 16); v[7] = __uint_as_float(a.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 bf8_pack(const float (&v)[8]) {
  return make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
}
__device__ __forceinline__ float block_sum1024(float v, float* red /* 16 */) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += red[i];
  return s;
}
__global__ __launch_bounds__(1024) void bnrelu_fwd_fast_kernel(int N, int L, int Hp, const bf16_t* __restrict__ u, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float eps, float momentum, bf16_t* __restrict__ r,
                                                              float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ run_mean,
                                                              float* __restrict__ run_var, long long* __restrict__ nbt) {
  __shared__ float red[16];
  const int l = blockIdx.x, tpr = Hp >> 3, rl_n = 1024 / tpr;
  const int cq = threadIdx.x % tpr, rl = threadIdx.x / tpr;
  const long long cnt = (long long)N * Hp, rstride = (long long)L * Hp;
  const bf16_t* base = u + (long long)l * Hp + cq * 8;
  const float shift = bf2f(u[(long long)l * Hp]);
  float s = 0.f, q = 0.f;
  int n = rl;
  for (; n + 3 * rl_n < N; n += 4 * rl_n) {
    uint4 a[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] = *reinterpret_cast<const uint4*>(base + (n + k * rl_n) * rstride);
#pragma unroll
    for (int k = 0; k < 4; ++k) { float v[8]; bf8_unpack(a[k], v);
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[e] - shift; s += d; q += d * d; } }
  }
  for (; n < N; n += rl_n) { float v[8]; bf8_unpack(*reinterpret_cast<const uint4*>(base + n * rstride), v);
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float d = v[e] - shift; s += d; q += d * d; } }
  const float ms = block_sum1024(s, red) / cnt;
  const float var = fmaxf(block_sum1024(q, red) / cnt - ms * ms, 0.f);
  const float mu = shift + ms;
  const float rs = rsqrtf(var + eps), gm = gamma[l], bt = beta[l];
  const float a1 = rs * gm, a0 = bt - mu * rs * gm;
  bf16_t* ob = r + (long long)l * Hp + cq * 8;
  n = rl;
  for (; n + 3 * rl_n < N; n += 4 * rl_n) {
    uint4 a[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] = *reinterpret_cast<const uint4*>(base + (n + k * rl_n) * rstride);
#pragma unroll
    for (int k = 0; k < 4; ++k) { float v[8]; bf8_unpack(a[k], v);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = fmaxf((v[e] - mu) * rs * gm + bt, 0.f);
      *reinterpret_cast<uint4*>(ob + (n + k * rl_n) * rstride) = bf8_pack(v); }
  }
  for (; n < N; n += rl_n) { float v[8]; bf8_unpack(*reinterpret_cast<const uint4*>(base + n * rstride), v);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = fmaxf((v[e] - mu) * rs * gm + bt, 0.f);
    *reinterpret_cast<uint4*>(ob + n * rstride) = bf8_pack(v); }
  (void)a0; (void)a1;
  if (threadIdx.x == 0) {
    mean[l] = mu; rstd[l] = rs;
    if (run_mean) {
      run_mean[l] = (1.f - momentum) * run_mean[l] + momentum * mu;
      run_var[l] = (1.f - momentum) * run_var[l] + momentum * var * ((float)cnt / (float)(cnt - 1));
    }
    if (nbt && l == 0) *nbt += 1;
  }
}
__global__ __launch_bounds__(1024) void bnrelu_bwd_fast_kernel(int N, int L, int Hp, const bf16_t* __restrict__ u, const bf16_t* __restrict__ dr,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              const float* __restrict__ mean, const float* __restrict__ rstd, bf16_t* __restrict__ du,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ float red[16];
  const int l = blockIdx.x, tpr = Hp >> 3, rl_n = 1024 / tpr;
  const int cq = threadIdx.x % tpr, rl = threadIdx.x / tpr;
  const long long cnt = (long long)N * Hp, rstride = (long long)L * Hp, off0 = (long long)l * Hp + cq * 8;
  const float mu = mean[l], rs = rstd[l], gm = gamma[l], bt = beta[l];
  float s1 = 0.f, s2 = 0.f;
  int n = rl;
  for (; n + rl_n < N; n += 2 * rl_n) {
    uint4 a[2], b[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) { a[k] = *reinterpret_cast<const uint4*>(u + off0 + (n + k * rl_n) * rstride); b[k] = *reinterpret_cast<const uint4*>(dr + off0 + (n + k * rl_n) * rstride); }
#pragma unroll
    for (int k = 0; k < 2; ++k) { float x[8], g[8]; bf8_unpack(a[k], x); bf8_unpack(b[k], g);
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float xh = (x[e] - mu) * rs; const float gg = (xh * gm + bt) > 0.f ? g[e] : 0.f; s1 += gg; s2 += gg * xh; } }
  }
  for (; n < N; n += rl_n) { float x[8], g[8]; bf8_unpack(*reinterpret_cast<const uint4*>(u + off0 + n * rstride), x); bf8_unpack(*reinterpret_cast<const uint4*>(dr + off0 + n * rstride), g);
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float xh = (x[e] - mu) * rs; const float gg = (xh * gm + bt) > 0.f ? g[e] : 0.f; s1 += gg; s2 += gg * xh; } }
  s1 = block_sum1024(s1, red); s2 = block_sum1024(s2, red);
  const float m1 = s1 / cnt, m2 = s2 / cnt;
  n = rl;
  for (; n + rl_n < N; n += 2 * rl_n) {
    uint4 a[2], b[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) { a[k] = *reinterpret_cast<const uint4*>(u + off0 + (n + k * rl_n) * rstride); b[k] = *reinterpret_cast<const uint4*>(dr + off0 + (n + k * rl_n) * rstride); }
#pragma unroll
    for (int k = 0; k < 2; ++k) { float x[8], g[8]; bf8_unpack(a[k], x); bf8_unpack(b[k], g);
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float xh = (x[e] - mu) * rs; const float gg = (xh * gm + bt) > 0.f ? g[e] : 0.f; x[e] = gm * rs * (gg - m1 - xh * m2); }
      *reinterpret_cast<uint4*>(du + off0 + (n + k * rl_n) * rstride) = bf8_pack(x); }
  }
  for (; n < N; n += rl_n) { float x[8], g[8]; bf8_unpack(*reinterpret_cast<const uint4*>(u + off0 + n * rstride), x); bf8_unpack(*reinterpret_cast<const uint4*>(dr + off0 + n * rstride), g);
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float xh = (x[e] - mu) * rs; const float gg = (xh * gm + bt) > 0.f ? g[e] : 0.f; x[e] = gm * rs * (gg - m1 - xh * m2); }
    *reinterpret_cast<uint4*>(du + off0 + n * rstride) = bf8_pack(x); }
  if (threadIdx.x == 0) { dgamma[l] += s2; dbeta[l] += s1; }
}
static bool bn_fast_ok(int dtype, int Hp, const void* a, const void* b) {
  return dtype == CSMAE_BF16 && Hp % 8 == 0 && Hp / 8 <= 1024 && 1024 % (Hp / 8) == 0 && ((((uintptr_t)a | (uintptr_t)b) & 15) == 0);
}

extern "C" int csmae_bnrelu_fwd(int dtype, int N, int L, int Hp, const void* u, const float* gamma, const float* beta, float eps,
                                float momentum, void* r, float* mean, float* rstd, float* running_mean, float* running_var,
                                long long* num_batches_tracked, int training, void* stream) {
  CSMAE_REQUIRE(N > 0 && L > 0 && Hp > 0 && Hp % 4 == 0, "csmae_bnrelu_fwd: bad geometry N=%d L=%d Hp=%d", N, L, Hp);
  CSMAE_REQUIRE(!training || (long long)N * Hp > 1, "csmae_bnrelu_fwd: need more than one value per channel (torch raises the same)");
  CSMAE_REQUIRE(training || (running_mean && running_var), "csmae_bnrelu_fwd: eval mode needs running statistics");
  hipStream_t st = (hipStream_t)stream;
  if (training && bn_fast_ok(dtype, Hp, u, r)) {
    hipLaunchKernelGGL(bnrelu_fwd_fast_kernel, dim3(L), dim3(1024), 0, st, N, L, Hp, (const bf16_t*)u, gamma, beta, eps, momentum, (bf16_t*)r, mean, rstd, running_mean, running_var, num_batches_tracked);
    return csmae_check_launch("csmae_bnrelu_fwd");
  }
  if (dtype == CSMAE_BF16) hipLaunchKernelGGL((bnrelu_fwd_kernel<bf16_t>), dim3(L), dim3(1024), 0, st, N, L, Hp, (const bf16_t*)u, gamma, beta, eps, momentum, (bf16_t*)r, mean, rstd, running_mean, running_var, num_batches_tracked, training);
  else if (dtype == CSMAE_F32) hipLaunchKernelGGL((bnrelu_fwd_kernel<float>), dim3(L), dim3(1024), 0, st, N, L, Hp, (const float*)u, gamma, beta, eps, momentum, (float*)r, mean, rstd, running_mean, running_var, num_batches_tracked, training);
  else { csmae_set_error("csmae_bnrelu_fwd: bad dtype %d", dtype); return CSMAE_ERR_UNSUPPORTED; }
  return csmae_check_launch("csmae_bnrelu_fwd");
}

extern "C" int csmae_bnrelu_bwd(int dtype, int N, int L, int Hp, const void* u, const void* dr, const float* gamma, const float* beta,
                                const float* mean, const float* rstd, void* du, float* dgamma, float* dbeta, void* stream) {
  CSMAE_REQUIRE(N > 0 && L > 0 && Hp > 0 && Hp % 4 == 0, "csmae_bnrelu_bwd: bad geometry N=%d L=%d Hp=%d", N, L, Hp);
  hipStream_t st = (hipStream_t)stream;
  if (bn_fast_ok(dtype, Hp, u, dr) && (((uintptr_t)du & 15) == 0)) {
    hipLaunchKernelGGL(bnrelu_bwd_fast_kernel, dim3(L), dim3(1024), 0, st, N, L, Hp, (const bf16_t*)u, (const bf16_t*)dr, gamma, beta, mean, rstd, (bf16_t*)du, dgamma, dbeta);
    return csmae_check_launch("csmae_bnrelu_bwd");
  }
  if (dtype == CSMAE_BF16) hipLaunchKernelGGL((bnrelu_bwd_kernel<bf16_t>), dim3(L), dim3(1024), 0, st, N, L, Hp, (const bf16_t*)u, (const bf16_t*)dr, gamma, beta, mean, rstd, (bf16_t*)du, dgamma, dbeta);
  else if (dtype == CSMAE_F32) hipLaunchKernelGGL((bnrelu_bwd_kernel<float>), dim3(L), dim3(1024), 0, st, N, L, Hp, (const float*)u, (const float*)dr, gamma, beta, mean, rstd, (float*)du, dgamma, dbeta);
  else { csmae_set_error("csmae_bnrelu_bwd: bad dtype %d", dtype); return CSMAE_ERR_UNSUPPORTED; }
  return csmae_check_launch("csmae_bnrelu_bwd");
}
