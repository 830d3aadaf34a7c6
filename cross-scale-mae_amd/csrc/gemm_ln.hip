// Full-row bf16 GEMM with LayerNorm in its epilogue (round 6).  See gemm_common.h / gemm_k2.hip for the pieces it shares.
//
// timm's pre-norm Block (MAE_ViT_Baseline.py:160-188) puts a LayerNorm on either side of four of its eight products:
//   forward   x_mid = x + proj(o)        -> norm2(x_mid) -> fc1          x_out = x_mid + fc2(h) -> the NEXT block's norm1 -> qkv
//   backward  d(norm2 out) = dpre W_fc1  -> norm2' + residual gradient   d(norm1 out) = dqkv W_qkv -> norm1' + residual gradient
// As kernels of their own these LayerNorms are 123 launches per ViT-B step on the critical chain, 8 B per element of HBM traffic each in the
// backward pass, and they run at half their speed beside the weight-gradient launches.  A LayerNorm needs the WHOLE output row, and a
// 256-column tile holds half of a decoder row: so this kernel's tile is 128 x 512 — EIGHT waves side by side along N (1 x 8), each the 128 x 64 wave
// tile, the wave-private B ring and the hand-scheduled K step of k2_tile; the A image (128 rows x 64 k) is shared by all eight, two 16-KiB slots,
// one barrier per K step.  LDS: A 2 x 16 KiB + 8 x 12 KiB of B halves + 10 KiB of row statistics = 138 KiB: one workgroup per CU, two waves per
// SIMD, the same per-wave instruction stream as two k2 workgroups on a CU (and 80 instead of 96 KiB staged per K step).  With the row in ONE
// workgroup the epilogue does the LayerNorm itself:
//   FWD  x_new = acc + bias + resid (bf16 stream, written);  y = LN(x_new) * gamma + beta (written);  mean / rstd (written)
//        — the row statistics by Chan's merge of the eight waves' (mean, M2) pairs through LDS: one exchange, no cancellation;
//   BWD  dx = rstd * (g - mean(g) - xhat * mean(g * xhat)) + dres_in,  g = bf16(acc) * gamma  (exactly ln_bwd_kernel's arithmetic on the rounded
//        product, which never reaches HBM), xhat from the saved bf16 stream and statistics; dgamma / dbeta partial rows per tile for
//        csmae_ln_param_reduce_rows — the two row sums cross the eight waves through LDS in a fixed order (bit-reproducible).
// N <= 512 with N % 64 == 0 (the decoders of every preset: 512); waves beyond N idle through the barriers.  D = 768 and up (the encoders) do not
// fit: 128 x 768 needs 192 accumulator registers per lane at two waves per SIMD.
#include "gemm_common.h"

#ifdef GEMM_TIMING
extern "C" int csmae_debug_k8_ts(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gemm_ts), sizeof(g_gemm_ts)); }
#endif

struct LnEpi {
  const float* gamma; const float* beta; float eps;
  void* y; long long ldy;              // FWD: the normalised rows (bf16)
  float* mean; float* rstd;            // FWD: written;  BWD: read
  const void* x; long long ldx;        // BWD: the LayerNorm's input rows (bf16 residual stream)
  const void* dres; long long ldd;     // BWD: incoming residual gradient (bf16; nullable)
  float* part;                         // BWD: [tiles_m][2][N] partial rows of dgamma / dbeta
};

#define K8_ASLOT (128 * 64 * 2)
#define K8_BSLOT (64 * 32 * 2)
#define K8_BWAVE (3 * K8_BSLOT)
#define K8_BBASE (2 * K8_ASLOT)
#define K8_RED1 (K8_BBASE + 8 * K8_BWAVE)          // [8 waves][128 rows] float2: a wave's share of a row's two sums
#define K8_TOT (K8_RED1 + 8 * 128 * 8)             // [128 rows] float2: the row's totals, in the form the last pass applies
#define K8_STAT (K8_TOT + 128 * 8)                 // [128 rows] float2: BWD (rstd, -mean * rstd)
#define K8_LDS (K8_STAT + 128 * 8)                 // 141 312 B

// sum over the 8 consecutive lanes that hold one row segment (DPP: no LDS crossbar)
__device__ __forceinline__ float dpp_sum8(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
  return v;
}
__device__ __forceinline__ void unpack8(const uint4& a, float (&v)[8]) {
  v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u); v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
  v[4] = __uint_as_float(a.z << 16); v[5] = __uint_as_float(a.z & 0xffff0000u); v[6] = __uint_as_float(a.w << 16); v[7] = __uint_as_float(a.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack8(const float (&v)[8]) {
  return make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
}

// -------------------------------------------------------------------- epilogue, forward: residual add + LayerNorm of the new row
// Row-segment layout of a wave's 128 x 64 patch: lane = 8 * rsub + c8 owns columns 8 c8 .. 8 c8 + 7 of rows 8 s + rsub, s = 0 .. 15.
__device__ __forceinline__ void k8_epilogue_fwd(const GemmArgs& p, const LnEpi& q, f4_t (&acc)[8][4], char* smem, int w, int lane, int t, int g, int m0) {
  constexpr int ESTR = 68;   // floats per strip row (64 + pad)
  const int c8 = lane & 7, rsub = lane >> 3, wn = w * 64, gn = wn + c8 * 8;
  const bool wact = wn < p.N;                                      // wave-uniform
  float* ew = reinterpret_cast<float*>(smem + K8_BBASE + w * K8_BWAVE);   // the wave's own B ring: its last reads were waited for in the last K step
  float2* red1 = reinterpret_cast<float2*>(smem + K8_RED1);
  float2* tot = reinterpret_cast<float2*>(smem + K8_TOT);
  const int nwa = p.N >> 6;
  uint4 xr[16];
  const brsrc_t rsY = buf_rsrc(q.y, (long long)p.M * q.ldy * 2);
  unsigned offY = (unsigned)(((long long)(m0 + rsub) * q.ldy + gn) * 2);
  const unsigned stepY = (unsigned)(8 * q.ldy * 2);
  if (wact) {
    f4_t b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
    if (p.bias) { b0 = *reinterpret_cast<const f4_t*>(p.bias + gn); b1 = *reinterpret_cast<const f4_t*>(p.bias + gn + 4); }
    const brsrc_t rsC = buf_rsrc(p.C, (long long)p.M * p.ldc * 2);
    const brsrc_t rsR = buf_rsrc(p.resid, (long long)p.M * p.ldr * 2);
    unsigned offC = (unsigned)(((long long)(m0 + rsub) * p.ldc + gn) * 2), offR = (unsigned)(((long long)(m0 + rsub) * p.ldr + gn) * 2);
    const unsigned stepC = (unsigned)(8 * p.ldc * 2), stepR = (unsigned)(8 * p.ldr * 2);
    uint4 rv[2][4];
    auto load_part = [&](uint4 (&dst)[4]) {
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) { dst[ps] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsR, offR, 0, 0)); offR += stepR; }
    };
    load_part(rv[0]);
#pragma unroll
    for (int part = 0; part < 4; ++part) {
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<f4_t*>(ew + (ii * 16 + t) * ESTR + j * 16 + 4 * g) = acc[part * 2 + ii][j];
      if (part + 1 < 4) load_part(rv[(part + 1) & 1]);   // the next part's residual rows go out before this part's stores (vmcnt is one in-order counter)
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {
        const int s = part * 4 + ps;
        const float* src = ew + (ps * 8 + rsub) * ESTR + c8 * 8;
        const f4_t v0 = *reinterpret_cast<const f4_t*>(src) + b0, v1 = *reinterpret_cast<const f4_t*>(src + 4) + b1;
        float r[8], v[8];
        unpack8(rv[part & 1][ps], r);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = v0[e] + r[e]; v[4 + e] = v1[e] + r[4 + e]; }
        xr[s] = pack8(v);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4_t, xr[s]), rsC, offC, 0, 0);
        offC += stepC;
        // this wave's share of the row's statistics, on the ROUNDED values (what the backward pass will read back): local mean, local M2
        unpack8(xr[s], v);
        float sm = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        sm = dpp_sum8(sm);
        const float ml = sm * (1.0f / 64.0f);
        float m2 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = v[e] - ml; m2 = fmaf(d, d, m2); }
        m2 = dpp_sum8(m2);
        if (c8 == 0) red1[w * 128 + s * 8 + rsub] = make_float2(ml, m2);
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < 128) {   // Chan's merge of the waves' (mean, M2) pairs, wave order fixed: mu = mean of means (equal counts), M2 = sum M2_w + 64 sum (m_w - mu)^2
    float mk[8], mu = 0.f, M2 = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) if (k < nwa) { const float2 a = red1[k * 128 + threadIdx.x]; mk[k] = a.x; mu += a.x; M2 += a.y; }
    mu /= (float)nwa;
    float dev = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) if (k < nwa) { const float d = mk[k] - mu; dev = fmaf(d, d, dev); }
    const float rs = rsqrtf((M2 + 64.f * dev) / (float)p.N + q.eps);
    tot[threadIdx.x] = make_float2(rs, -mu * rs);
    const int row = m0 + (int)threadIdx.x;
    if (row < p.M) { q.mean[row] = mu; q.rstd[row] = rs; }
  }
  __syncthreads();
  if (wact) {
    float ga[8], be[8];
    *reinterpret_cast<f4_t*>(ga) = *reinterpret_cast<const f4_t*>(q.gamma + gn); *reinterpret_cast<f4_t*>(ga + 4) = *reinterpret_cast<const f4_t*>(q.gamma + gn + 4);
    *reinterpret_cast<f4_t*>(be) = *reinterpret_cast<const f4_t*>(q.beta + gn); *reinterpret_cast<f4_t*>(be + 4) = *reinterpret_cast<const f4_t*>(q.beta + gn + 4);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const float2 a = tot[s * 8 + rsub];
      float v[8];
      unpack8(xr[s], v);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = fmaf(fmaf(v[e], a.x, a.y), ga[e], be[e]);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4_t, pack8(v)), rsY, offY, 0, 0);
      offY += stepY;
    }
  }
}

// -------------------------------------------------------------------- epilogue, backward: LayerNorm' of the product + residual gradient
// The rounded product stays in LDS between the two passes (a wave's 128 x 64 patch as bf16 rows of 136 B: 17 KiB, laid over the operand rings once
// every wave has left the loop), x and the residual gradient in registers (128): three tensors in registers (192) spilled.
#define K8_BSTRIP (128 * 136)
#define K8B_RED1 (8 * K8_BSTRIP)                    // 139 264
#define K8B_TOT (K8B_RED1 + 8 * 128 * 8)
#define K8B_STAT (K8B_TOT + 128 * 8)
#define K8B_LDS (K8B_STAT + 128 * 8)                // 149 504 B
__device__ __forceinline__ void k8_epilogue_bwd(const GemmArgs& p, const LnEpi& q, f4_t (&acc)[8][4], char* smem, int w, int lane, int t, int g, int m0, int tm,
                                                float st_mu, float st_rs) {
  constexpr int ROWB = 136;   // bytes per bf16 strip row (128 + pad): the product crosses the strip rounded, as it would cross HBM
  const int c8 = lane & 7, rsub = lane >> 3, wn = w * 64, gn = wn + c8 * 8;
  const bool wact = wn < p.N;
  char* ew = smem + w * K8_BSTRIP;
  float2* red1 = reinterpret_cast<float2*>(smem + K8B_RED1);
  float2* tot = reinterpret_cast<float2*>(smem + K8B_TOT);
  float2* stat = reinterpret_cast<float2*>(smem + K8B_STAT);
  const int nwa = p.N >> 6;
  if (threadIdx.x < 128) stat[threadIdx.x] = make_float2(st_rs, -st_mu * st_rs);
  __syncthreads();   // ... and every wave has left the main loop: the strips lie over the A slots and the B rings
  uint4 xv[16], rv[16];
  float ga[8];
  const brsrc_t rsC = buf_rsrc(p.C, (long long)p.M * p.ldc * 2);
  unsigned offC = (unsigned)(((long long)(m0 + rsub) * p.ldc + gn) * 2);
  const unsigned stepC = (unsigned)(8 * p.ldc * 2);
  const char* dsrc = ew + rsub * ROWB + c8 * 16;   // this lane's row segment of strip rows 8 s + rsub
  if (wact) {
    *reinterpret_cast<f4_t*>(ga) = *reinterpret_cast<const f4_t*>(q.gamma + gn); *reinterpret_cast<f4_t*>(ga + 4) = *reinterpret_cast<const f4_t*>(q.gamma + gn + 4);
    const brsrc_t rsX = buf_rsrc(q.x, (long long)p.M * q.ldx * 2);
    unsigned offX = (unsigned)(((long long)(m0 + rsub) * q.ldx + gn) * 2);
    const unsigned stepX = (unsigned)(8 * q.ldx * 2);
#pragma unroll
    for (int s = 0; s < 16; ++s) { xv[s] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsX, offX, 0, 0)); offX += stepX; }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f4_t v = acc[i][j];
        *reinterpret_cast<uint2*>(ew + (i * 16 + t) * ROWB + (j * 16 + 4 * g) * 2) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
      }
    asm volatile("" ::: "memory");   // (the row segments read below are other lanes' writes: keep the compiler from moving the reads above them)
    // the residual gradient does not depend on the sums: its loads go out before the exchange
    if (q.dres) {
      const brsrc_t rsD = buf_rsrc(q.dres, (long long)p.M * q.ldd * 2);
      unsigned offD = (unsigned)(((long long)(m0 + rsub) * q.ldd + gn) * 2);
      const unsigned stepD = (unsigned)(8 * q.ldd * 2);
#pragma unroll
      for (int s = 0; s < 16; ++s) { rv[s] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsD, offD, 0, 0)); offD += stepD; }
    } else {
#pragma unroll
      for (int s = 0; s < 16; ++s) rv[s] = make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const float2 a = stat[s * 8 + rsub];
      float d[8], x[8];
      unpack8(*reinterpret_cast<const uint4*>(dsrc + s * 8 * ROWB), d); unpack8(xv[s], x);
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float gg = d[e] * ga[e]; s1 += gg; s2 = fmaf(gg, fmaf(x[e], a.x, a.y), s2); }
      s1 = dpp_sum8(s1); s2 = dpp_sum8(s2);
      if (c8 == 0) red1[w * 128 + s * 8 + rsub] = make_float2(s1, s2);
      __builtin_amdgcn_sched_barrier(0);   // (one segment at a time: the unpacked operands of several segments do not fit beside the 128 packed registers)
    }
  }
  __syncthreads();
  if (threadIdx.x < 128) {   // the row's two sums over the waves, wave order fixed
    float S1 = 0.f, S2 = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) if (k < nwa) { const float2 a = red1[k * 128 + threadIdx.x]; S1 += a.x; S2 += a.y; }
    const float rs = stat[threadIdx.x].x, inv = 1.0f / (float)p.N;
    tot[threadIdx.x] = make_float2(-rs * (S1 * inv), -rs * (S2 * inv));
  }
  __syncthreads();
  if (wact) {
    float ag[8], ab[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { ag[e] = 0.f; ab[e] = 0.f; }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const float2 a = stat[s * 8 + rsub], b = tot[s * 8 + rsub];
      float d[8], x[8], r[8], o[8];
      unpack8(*reinterpret_cast<const uint4*>(dsrc + s * 8 * ROWB), d); unpack8(xv[s], x); unpack8(rv[s], r);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xh = fmaf(x[e], a.x, a.y), gg = d[e] * ga[e];
        o[e] = fmaf(xh, b.y, fmaf(gg, a.x, b.x)) + r[e];   // rstd * (g - m1 - xhat * m2) + dres
        ag[e] = fmaf(d[e], xh, ag[e]); ab[e] += d[e];
      }
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4_t, pack8(o)), rsC, offC, 0, 0);
      offC += stepC;
      __builtin_amdgcn_sched_barrier(0);
    }
    // dgamma / dbeta of the tile: a lane holds its 8 columns summed over its 16 rows; the 8 row groups are folded through the wave's strip in lane order
    asm volatile("" ::: "memory");
    float* es = reinterpret_cast<float*>(ew);
#pragma unroll
    for (int e = 0; e < 8; ++e) { es[lane * 17 + e] = ag[e]; es[64 * 17 + lane * 17 + e] = ab[e]; }
    asm volatile("" ::: "memory");
    float sg = 0.f, sb = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) { const int src = (k * 8 + (lane >> 3)) * 17 + (lane & 7); sg += es[src]; sb += es[64 * 17 + src]; }
    if (q.part) {
      float* dst = q.part + (long long)tm * 2 * p.N + wn + lane;
      dst[0] = sg; dst[p.N] = sb;
    }
  }
}

// -------------------------------------------------------------------- the tile: k2_tile's K step with eight waves on one A image
template <bool TB, int EPI>   // TB: B is K-strided (dX = dY W, W [K][N]) | K-contiguous as its K-slab mirror (y = x W^T).  EPI 0: forward LayerNorm, 1: backward
__global__ __launch_bounds__(512, 2) void gemm_bf16_k8_kernel(GemmArgs p, LnEpi q) {
  constexpr int NW = 8, FM = 8, FN = 4;
  constexpr int ASLOT = K8_ASLOT, BSLOT = K8_BSLOT, BWAVE = K8_BWAVE, BBASE = K8_BBASE;
  __shared__ __attribute__((aligned(16))) char smem[EPI == 0 ? K8_LDS : K8B_LDS];
  const int tm = xcd_remap(blockIdx.x, p.tiles_m);
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int t = lane & 15, g = lane >> 4;
  const int m0 = tm * 128, wn = w * 64;
  GTS(0);
  float st_mu = 0.f, st_rs = 0.f;
  if (EPI == 1 && threadIdx.x < 128) { const int row = min(m0 + (int)threadIdx.x, p.M - 1); st_mu = q.mean[row]; st_rs = q.rstd[row]; }   // (latency hidden by the whole loop)
  const i4_t rsA = make_rsrc(p.A, p.a_bytes), rsB = make_rsrc(p.B, p.b_bytes);
  // DMA descriptors as in k2_tile, except that wave w stages rows 16 w .. 16 w + 15 of the A image (two pieces of 8 rows x 128 B)
  const int l3 = lane >> 3;
  const unsigned avo = (unsigned)(((long long)(m0 + w * 16 + l3) * p.lda + ((lane & 7) ^ l3) * 8) * 2);
  const unsigned aqs = (unsigned)(8 * p.lda * 2);
  unsigned bvo[2], bqs, bhs;
  if (!TB) {
    const int r = lane >> 2, c = (lane & 3) ^ ((r >> 1) & 3);
    bvo[0] = bvo[1] = wn < p.N ? (unsigned)(((long long)(wn + r) * 32 + c * 8) * 2) : OOB_OFF;   // (a wave beyond N: its columns are another slab's rows, not out of range)
    bqs = 16u * 64u;
    bhs = (unsigned)(p.ldb * 64);
  } else {
    const int kr = lane >> 3, ch = lane & 7;
#pragma unroll
    for (int qo = 0; qo < 2; ++qo) {
      const int key = ((kr >> 1) & 1) | (qo << 1);
      bvo[qo] = (unsigned)(((long long)kr * p.ldb + wn + ((((ch >> 1) ^ key) << 1) | (ch & 1)) * 8) * 2);
    }
    bqs = (unsigned)(8 * p.ldb * 2);
    bhs = (unsigned)(32 * p.ldb * 2);
  }
  const unsigned lds0 = (unsigned)(size_t)LDS_PTR(char, smem);
  const unsigned ldsA = lds0 + (unsigned)(w * 2) * 1024u, ldsB = lds0 + (unsigned)(BBASE + w * BWAVE);
  auto dma_a = [&](int slot, int j, int qq) {
    lds_dma16u(rsA, avo + ((unsigned)j * 128u + (unsigned)qq * aqs), ldsA + (unsigned)(slot * ASLOT + qq * 1024));
  };
  auto dma_b = [&](int slot, int u, int qq) {
    lds_dma16u(rsB, bvo[qq & 1] + ((unsigned)u * bhs + (unsigned)qq * bqs), ldsB + (unsigned)(slot * BSLOT + qq * 1024));
  };
  f4_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int jj = 0; jj < FN; ++jj) acc[i][jj] = f4_t{0.f, 0.f, 0.f, 0.f};
  const int ra0 = t * 128 + ((g ^ (t & 7)) << 4);
  const int rb0 = !TB ? t * 64 + ((g ^ ((t >> 1) & 3)) << 4)
                      : (8 * g + (t >> 2)) * 128 + ((((t >> 3) & 1) | ((g & 1) << 1)) << 5) + (t & 3) * 8;
  constexpr int BOPS = TB ? 2 * FN : FN;
  constexpr int W0_0 = (FM - 2) + BOPS, W0_I = (FM - 1) + BOPS, W1 = FM - 2;
  static_assert(W0_I <= 15, "lgkmcnt is a 4-bit counter");
  auto read_a = [&](int slot, int h, auto ic, s8_t& fa) {
    constexpr int i = decltype(ic)::value;
    const unsigned addr = lds0 + (unsigned)(slot * ASLOT) + (unsigned)(ra0 ^ (h << 6));
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(fa) : "v"(addr), "n"(i * 2048));
  };
  auto read_b = [&](int slot, s8_t (&fb)[FN]) {
    const unsigned addr = ldsB + (unsigned)(slot * BSLOT) + (unsigned)rb0;
    if (!TB) {
      asm volatile("ds_read_b128 %0, %1" : "=v"(fb[0]) : "v"(addr));
      asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(fb[1]) : "v"(addr));
      asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(fb[2]) : "v"(addr));
      asm volatile("ds_read_b128 %0, %1 offset:3072" : "=v"(fb[3]) : "v"(addr));
    } else {
#pragma unroll
      for (int jn = 0; jn < FN; ++jn) {
        const unsigned a2 = ldsB + (unsigned)(slot * BSLOT) + (unsigned)(rb0 ^ (jn << 5));
        s4_t lo, hi;
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a2));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:512" : "=v"(hi) : "v"(a2));
        fb[jn] = join_s4(lo, hi);
      }
    }
  };
  auto wait_a = [&](auto n, s8_t& f0) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f0) : "n"(decltype(n)::value)); };
  auto wait_ab = [&](auto n, s8_t& f0, s8_t& f1, s8_t (&fb)[FN]) {
    asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(f0), "+v"(f1), "+v"(fb[0]), "+v"(fb[1]), "+v"(fb[2]), "+v"(fb[3]) : "n"(decltype(n)::value));
  };
  auto mma_row = [&](int i, const s8_t& fa, const s8_t (&fb)[FN]) {
#pragma unroll
    for (int jj = 0; jj < FN; ++jj)
      acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8_t, fb[jj]), __builtin_bit_cast(bf8_t, fa), acc[i][jj], 0, 0, 0);
  };
  // ---- prologue: A_0 (2 pieces per wave), B_0^0, B_0^1 (needed at once), A_1, B_1^0 (needed at step 0's barrier)
  const int nsteps = p.ktiles;
#pragma unroll
  for (int qq = 0; qq < 2; ++qq) dma_a(0, 0, qq);
#pragma unroll
  for (int qq = 0; qq < 4; ++qq) dma_b(0, 0, qq);
#pragma unroll
  for (int qq = 0; qq < 4; ++qq) dma_b(1, 1, qq);
  if (nsteps >= 2) {
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) dma_a(1, 1, qq);
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) dma_b(2, 2, qq);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  GTS(1);
  s8_t fa[FM], fb0[FN], fb1[FN];
#pragma unroll
  for (int i = 0; i < FM; ++i) fa[i] = s8_t{0, 0, 0, 0, 0, 0, 0, 0};
  read_a(0, 0, std::integral_constant<int, 0>{}, fa[0]); read_a(0, 0, std::integral_constant<int, 1>{}, fa[1]);
  read_b(0, fb0);
  static_for<FM - 2>([&](auto ic) { constexpr int i = decltype(ic)::value + 2; read_a(0, 0, std::integral_constant<int, i>{}, fa[i]); });
  auto nx3 = [](int s, int k) { s += k; return s >= 3 ? s - 3 : s; };
  // One K step = 16 rows of 4 MFMAs.  DMA of the step, one piece per row: B_{j+1}^1 into B_j^0's slot (rows 1, 3, 5, 7), B_{j+2}^0 into B_j^1's slot
  // (rows 8, 9, 14, 15), this wave's TWO pieces of A_{j+2} into A_j's slot behind the barrier (rows 10, 11).  vmcnt (issue order per wave and step:
  // 4 b1 | 2 b0 | 2 a | 2 b0): at the step's head B_j^1 has the 6 pieces of the previous step's second half behind it; at the barrier A_{j+1} and
  // B_{j+1}^0 have this step's 4 + 2 behind them.   MODE 0 steady | 2 second-to-last step | 3 last step
  auto step = [&](int j, int sa, int sb, auto mode_tag) {
    constexpr int MODE = decltype(mode_tag)::value;
    constexpr bool more = MODE < 3;
    const int sb1 = nx3(sb, 1), sb2 = nx3(sb, 2);
    if (MODE == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    read_b(sb1, fb1);
    static_for<FM>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      if (i == 0) wait_ab(std::integral_constant<int, W0_0>{}, fa[0], fa[1], fb0);
      else if (i >= 2) wait_a(std::integral_constant<int, W0_I>{}, fa[i]);
      mma_row(i, fa[i], fb0);
      read_a(sa, 1, ic, fa[i]);
      if (more && (i & 1)) dma_b(sb, 2 * j + 3, i >> 1);
      __builtin_amdgcn_sched_barrier(0);
    });
    wait_ab(std::integral_constant<int, W1>{}, fa[0], fa[1], fb1);
    mma_row(0, fa[0], fb1);
    if (MODE == 0) dma_b(sb1, 2 * j + 4, 0);
    __builtin_amdgcn_sched_barrier(0);
    mma_row(1, fa[1], fb1);
    if (MODE == 0) dma_b(sb1, 2 * j + 4, 1);
    __builtin_amdgcn_sched_barrier(0);
    if (MODE == 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else if (MODE == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (more) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (more) {
      read_a(sa ^ 1, 0, std::integral_constant<int, 0>{}, fa[0]); read_a(sa ^ 1, 0, std::integral_constant<int, 1>{}, fa[1]); read_b(sb2, fb0);
    }
    static_for<FM - 2>([&](auto ic) {
      constexpr int i = decltype(ic)::value + 2;
      mma_row(i, fa[i], fb1);
      if (more) read_a(sa ^ 1, 0, std::integral_constant<int, i>{}, fa[i]);
      if (MODE == 0) { if (i < 4) dma_a(sa, j + 2, i - 2); else if (i >= 6) dma_b(sb1, 2 * j + 4, i - 4); }
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  int j = 0, sa = 0, sb = 0;
  for (; j < nsteps - 2; ++j, sa ^= 1, sb = nx3(sb, 2)) step(j, sa, sb, std::integral_constant<int, 0>{});
  if (nsteps >= 2) { step(j, sa, sb, std::integral_constant<int, 2>{}); ++j; sa ^= 1; sb = nx3(sb, 2); }
  step(j, sa, sb, std::integral_constant<int, 3>{});
  GTS(2);
  if ((p.force_cfg & 16) && acc[0][0][0] != 123456.0f) return;  // tuning aid: main loop only
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (EPI == 0) k8_epilogue_fwd(p, q, acc, smem, w, lane, t, g, m0);
  else k8_epilogue_bwd(p, q, acc, smem, w, lane, t, g, m0, tm, st_mu, st_rs);
  GTS(3);
}

// -------------------------------------------------------------------- C ABI
static int k8_force_loop_only() { static const int v = csmae_debug_opt("k8_loop_only") ? 16 : 0; return v; }
static bool k8_shape_ok(long long M, long long N, long long K) { return M >= 1 && N >= 64 && N <= 512 && N % 64 == 0 && K >= 64 && K % 64 == 0; }
extern "C" int csmae_gemm_ln_supported(long long M, long long N, long long K) { return k8_shape_ok(M, N, K) ? 1 : 0; }

static void k8_fill(GemmArgs& p, long long M, long long N, long long K) {
  p.force_cfg = k8_force_loop_only(); p.split_stride = 0; p.colsum = nullptr; p.dq_a = p.dq_b = nullptr; p.a_fmt = 0; p.aux_q8 = 0; p.q_out = nullptr;
  p.aux = nullptr; p.ldaux = 0; p.M = (int)M; p.N = (int)N; p.K = (int)K; p.c_dtype = CSMAE_BF16; p.splitk = 1;
  p.ktiles = (int)(K / 64); p.ktiles_per_split = p.ktiles; p.tiles_m = cdiv(M, 128); p.tiles_n = 1;
}

// x_new = x W^T + bias + resid ;  y = LayerNorm(x_new; gamma, beta, eps) ;  mean / rstd of x_new — timm Block: `x = x + attn.proj(..)` + `norm2`, and
// `x = x + mlp.fc2(..)` + the next block's `norm1` (MAE_ViT_Baseline.py:160-188).  All tensors bf16 except bias / gamma / beta / mean / rstd (fp32).
extern "C" int csmae_gemm_ln_fwd(long long M, long long N, long long K, const void* A, long long lda, const void* Bk, long long slab_rows,
                                 const float* bias, const void* resid, long long ldr, void* X, long long ldx, const float* gamma, const float* beta,
                                 float eps, void* Y, long long ldy, float* mean, float* rstd, void* stream) {
  CSMAE_REQUIRE(k8_shape_ok(M, N, K), "csmae_gemm_ln_fwd: need 64 <= N <= 512, N %% 64 == 0, K %% 64 == 0 (M=%lld N=%lld K=%lld)", M, N, K);
  CSMAE_REQUIRE(A && Bk && resid && X && gamma && beta && Y && mean && rstd, "csmae_gemm_ln_fwd: null argument");
  CSMAE_REQUIRE(lda % 8 == 0 && lda >= K && slab_rows >= N && slab_rows % 4 == 0 && ldr % 8 == 0 && ldr >= N && ldx % 8 == 0 && ldx >= N && ldy % 8 == 0 && ldy >= N,
                "csmae_gemm_ln_fwd: leading dimensions must be multiples of 8 and cover the rows");
  CSMAE_REQUIRE((((uintptr_t)A | (uintptr_t)Bk | (uintptr_t)resid | (uintptr_t)X | (uintptr_t)Y | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)bias) & 15) == 0,
                "csmae_gemm_ln_fwd: pointers must be 16-byte aligned");
  CSMAE_REQUIRE((M + 128) * lda * 2 < 0xFFFFFFF0ll && (K / 32) * slab_rows * 64 < 0xFFFFFFF0ll && (M + 128) * ldr * 2 < 0xFFFFFFF0ll && (M + 128) * ldx * 2 < 0xFFFFFFF0ll &&
                (M + 128) * ldy * 2 < 0xFFFFFFF0ll, "csmae_gemm_ln_fwd: tensor larger than 4 GiB");
  GemmArgs p; k8_fill(p, M, N, K);
  p.A = A; p.B = Bk; p.C = X; p.bias = bias; p.resid = resid; p.lda = lda; p.ldb = slab_rows; p.ldc = ldx; p.ldr = ldr; p.epi = EPI_RESID;
  p.a_bytes = (unsigned)(M * lda * 2); p.b_bytes = (unsigned)((K / 32) * slab_rows * 64);
  LnEpi q{gamma, beta, eps, Y, ldy, mean, rstd, nullptr, 0, nullptr, 0, nullptr};
  CSMAE_LAUNCH((gemm_bf16_k8_kernel<false, 0>), dim3(p.tiles_m), dim3(512), 0, (hipStream_t)stream, p, q);
  return csmae_check_launch("csmae_gemm_ln_fwd");
}

// dx_out = LayerNorm'(dY W; x, mean, rstd, gamma) + dres_in, W [K][N] (the layer's weight as torch stores it: K = out-features), and the tile's
// dgamma / dbeta partial rows: partial_ws [ceil(M / 128)][2][N] for csmae_ln_param_reduce_rows — the backward of norm2 -> mlp.fc1 and
// norm1 -> attn.qkv of timm's Block (MAE_ViT_Baseline.py:160-188) as ONE kernel; the product itself never reaches HBM.
extern "C" int csmae_gemm_ln_bwd(long long M, long long N, long long K, const void* dY, long long lda, const void* W, long long ldb,
                                 const void* x, long long ldx, const float* mean, const float* rstd, const float* gamma, const void* dres_in, long long ldd,
                                 void* dx_out, long long ldo, float* partial_ws, long long partial_elems, void* stream) {
  CSMAE_REQUIRE(k8_shape_ok(M, N, K), "csmae_gemm_ln_bwd: need 64 <= N <= 512, N %% 64 == 0, K %% 64 == 0 (M=%lld N=%lld K=%lld)", M, N, K);
  CSMAE_REQUIRE(dY && W && x && mean && rstd && gamma && dx_out, "csmae_gemm_ln_bwd: null argument");
  CSMAE_REQUIRE(lda % 8 == 0 && lda >= K && ldb % 8 == 0 && ldb >= N && ldx % 8 == 0 && ldx >= N && ldo % 8 == 0 && ldo >= N && (!dres_in || (ldd % 8 == 0 && ldd >= N)),
                "csmae_gemm_ln_bwd: leading dimensions must be multiples of 8 and cover the rows");
  CSMAE_REQUIRE((((uintptr_t)dY | (uintptr_t)W | (uintptr_t)x | (uintptr_t)dres_in | (uintptr_t)dx_out | (uintptr_t)gamma) & 15) == 0, "csmae_gemm_ln_bwd: pointers must be 16-byte aligned");
  CSMAE_REQUIRE(!partial_ws || partial_elems >= (long long)cdiv(M, 128) * 2 * N, "csmae_gemm_ln_bwd: partial_ws needs ceil(M / 128) * 2 * N floats");
  CSMAE_REQUIRE((M + 128) * lda * 2 < 0xFFFFFFF0ll && (K + 64) * ldb * 2 < 0xFFFFFFF0ll && (M + 128) * ldx * 2 < 0xFFFFFFF0ll && (M + 128) * ldo * 2 < 0xFFFFFFF0ll &&
                (!dres_in || (M + 128) * ldd * 2 < 0xFFFFFFF0ll), "csmae_gemm_ln_bwd: tensor larger than 4 GiB");
  GemmArgs p; k8_fill(p, M, N, K);
  p.A = dY; p.B = W; p.C = dx_out; p.bias = nullptr; p.resid = nullptr; p.lda = lda; p.ldb = ldb; p.ldc = ldo; p.ldr = 0; p.epi = EPI_NONE;
  p.a_bytes = (unsigned)(M * lda * 2); p.b_bytes = (unsigned)(K * ldb * 2);
  LnEpi q{gamma, nullptr, 0.f, nullptr, 0, const_cast<float*>(mean), const_cast<float*>(rstd), x, ldx, dres_in, ldd, partial_ws};
  CSMAE_LAUNCH((gemm_bf16_k8_kernel<true, 1>), dim3(p.tiles_m), dim3(512), 0, (hipStream_t)stream, p, q);
  return csmae_check_launch("csmae_gemm_ln_bwd");
}
