// Softmax attention of the timm Block: per (sample, head) independent problems with
// tiny sequences (T = 50/65 encoder, 197/257 decoder; head_dim 64/80 and 32), so one workgroup owns one
// (sample, head) and keeps the whole K and V of that head in LDS: single-pass softmax, no online rescaling.
//
// bf16 path: everything is computed TRANSPOSED (S^T = K Q^T) so that after the MFMA each lane owns ONE query
// column: softmax row statistics are lane-local plus two cross-lane-group shuffles, and the probabilities sit
// in exactly the register layout the next MFMA wants as its B operand (no LDS round trip for P).  V^T / K^T /
// Q^T / dO^T operands come straight out of row-major LDS images through ds_read_b64_tr_b16.
// Backward = a query-row pass (dQ) and a key-column pass (dK, dV) that mirror the forward; S and dP are
// recomputed in each pass (attention is 3.7 % of the step's FLOPs; this avoids cross-wave reductions).
// fp32 path: exact fp32 (parity mode), one thread per query row / key column.
#include "common.h"
#include <cstdlib>

#define LOG2E 1.4426950408889634f
#define LN2 0.6931471805599453f

// ------------------------------------------------------------------------------------------ bf16 helpers
template <int HD> struct AttnLds { static constexpr int STRIDE = (HD + 8) * 2; };  // bytes per row (16-B multiple)

// stage rows [0,TP) of one head's matrix (column offset `col0` inside a [B*T, ld] tensor) into LDS, zero padded
template <int HD, int TP>
__device__ __forceinline__ void stage_head(char* dst, const bf16_t* src, long long row0, int ld, int col0, int T, int hd) {
  constexpr int CH = HD / 8;
  for (int e = threadIdx.x; e < TP * CH; e += blockDim.x) {
    int r = e / CH, c = e - r * CH;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (r < T && c * 8 < hd) v = *reinterpret_cast<const uint4*>(src + (row0 + r) * ld + col0 + c * 8);
    *reinterpret_cast<uint4*>(dst + r * AttnLds<HD>::STRIDE + c * 16) = v;
  }
}

// The same for several matrices at once, with every global load issued before the first LDS store.  stage_head() called three or
// four times in a row ran ~3.5 dependent load->store iterations per matrix: ~10 exposed memory latencies (~10 of the ~18 us a
// workgroup lives) before any arithmetic could start.
template <int HD, int TP, int NT, int NM>
struct HeadStager {
  static constexpr int CH = HD / 8, IT = (TP * CH + NT - 1) / NT;
  uint4 v[NM][IT];
  __device__ __forceinline__ void load(int m, const bf16_t* src, long long row0, int ld, int col0, int T, int hd) {
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int e = threadIdx.x + i * NT, r = e / CH, c = e - r * CH;
      v[m][i] = make_uint4(0, 0, 0, 0);
      if (e < TP * CH && r < T && c * 8 < hd) v[m][i] = *reinterpret_cast<const uint4*>(src + (row0 + r) * ld + col0 + c * 8);
    }
  }
  __device__ __forceinline__ void store(int m, char* dst) {
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int e = threadIdx.x + i * NT, r = e / CH, c = e - r * CH;
      if (e < TP * CH) *reinterpret_cast<uint4*>(dst + r * AttnLds<HD>::STRIDE + c * 16) = v[m][i];
    }
  }
};

// K-contiguous fragment (lane (t,g): row = row0 + t, elements d = ks*32 + 8g .. +8)
template <int HD>
__device__ __forceinline__ s8_t frag_rows(const char* img, int row0, int ks, int t, int g) {
  return *reinterpret_cast<const s8_t*>(img + (row0 + t) * AttnLds<HD>::STRIDE + (ks * 32 + 8 * g) * 2);
}
// transposed fragment via tr-read: A operand with i = column (c0 + t) and k = rows {r0+4g+j, r0+16+4g+j}
template <int HD>
__device__ __forceinline__ s8_t frag_cols_tr(const char* img, int r0, int c0, int t, int g) {
  const char* p = img + (r0 + 4 * g + (t >> 2)) * AttnLds<HD>::STRIDE + (c0 + (t & 3) * 4) * 2;
  s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s4_t, p));
  s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s4_t, p + 16 * AttnLds<HD>::STRIDE));
  return join_s4(lo, hi);
}
__device__ __forceinline__ s8_t pack_pair(f4_t a, f4_t b) {
  unsigned u0 = pack2bf(a[0], a[1]), u1 = pack2bf(a[2], a[3]), u2 = pack2bf(b[0], b[1]), u3 = pack2bf(b[2], b[3]);
  uint4 u = make_uint4(u0, u1, u2, u3);
  return __builtin_bit_cast(s8_t, u);
}
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8_t, a), __builtin_bit_cast(bf8_t, b), c, 0, 0, 0)

// ------------------------------------------------------------------------------------------ bf16 forward
// With 32-wide heads two neighbouring heads share every 128-B line of a qkv row.  Workgroups are dealt to the 8 XCDs round-robin,
// so consecutive (sample, head) ids land on different L2s and each line is fetched from HBM twice (measured: FETCH_SIZE = 2x the
// algorithmic bytes).  Inside each run of 16 ids, ids x and x + 8 (same XCD, back to back) take heads 2x' and 2x' + 1.
template <int HD>
__device__ __forceinline__ int pair_remap(int bid, int nblk) {
  if (HD * 2 >= 128 || (bid | 15) >= nblk) return bid;
  return (bid & ~15) + 2 * (bid & 7) + ((bid >> 3) & 1);
}

// EMIT (fp8 mode, csmae_attn_fwd_q / csmae_attn_bwd_q): the kernel also leaves its output as fp8 bytes for the GEMM that consumes it (attn.proj
// forward: e4m3; attn.qkv backward: e5m2), quantised from the ROUNDED bf16 values with the amax of one step earlier — exactly what the separate
// csmae_fp8_quantize pass over the bf16 tensor produced — and records the new amax.  Instantiations of their own: the bf16 step's kernels
// carry none of it.
template <int HD, int NKF, bool EMIT = false>
__global__ __launch_bounds__(256) void attn_fwd_bf16(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                     float* __restrict__ lse, int T, int H, int D, int hd, float scale, Fp8Emit em) {
  constexpr int TP = NKF * 16, KS = HD / 32, DF = HD / 16;
  // K and V of the head live in LDS (every wave reads all of them); a wave's Q fragments (its <= QB query blocks, needed by nobody
  // else) come straight from global memory into registers, fetched together with the staging loads: 36 instead of 54 KiB of LDS for
  // the decoder shape — four workgroups per CU instead of three — and no Q round trip through LDS (decoder launch 80.5 -> 76 us,
  // encoder 18.2 -> 16.3 us alone on the chip).
  constexpr int QB = (NKF + 3) / 4;
  __shared__ __attribute__((aligned(16))) char smem[2 * TP * AttnLds<HD>::STRIDE];
  char* Ks = smem;
  char* Vs = smem + TP * AttnLds<HD>::STRIDE;
  const int wid = pair_remap<HD>(blockIdx.x, gridDim.x);
  const int b = wid / H, h = wid - b * H;
  const long long row0 = (long long)b * T;
  const int ld = 3 * D;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, t = lane & 15, g = lane >> 4;
  uint4 qraw[QB][KS];
  {
    HeadStager<HD, TP, 256, 2> sg;
    sg.load(0, qkv, row0, ld, D + h * hd, T, hd); sg.load(1, qkv, row0, ld, 2 * D + h * hd, T, hd);
#pragma unroll
    for (int bq = 0; bq < QB; ++bq)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int qr = (w + 4 * bq) * 16 + t, col = ks * 32 + 8 * g;
        qraw[bq][ks] = make_uint4(0, 0, 0, 0);
        if (qr < T && col < hd) qraw[bq][ks] = *reinterpret_cast<const uint4*>(qkv + (row0 + qr) * ld + h * hd + col);
      }
    sg.store(0, Ks); sg.store(1, Vs);
  }
  __syncthreads();
  float qmax = 0.f, qseen = 0.f;
  const float qs = EMIT ? fp8_emit_scale(em, lane, qmax) : 1.f;
  const float c2 = scale * LOG2E;
  const int nqb = (T + 15) >> 4;
  constexpr int FIRST_PARTIAL = NKF <= 2 ? 0 : (NKF <= 6 ? NKF - 2 : (NKF == 14 ? 6 : 14));  // floor(T_min / 16) of the bucket dispatching to this NKF
  const int kthr = T - 4 * g;  // key f*16 + 4g + r is padding  <=>  f*16 + r >= kthr
#pragma unroll
  for (int bq = 0; bq < QB; ++bq) {
    const int qb = w + 4 * bq;
    if (qb >= nqb) break;
    const int q = qb * 16 + t;
    s8_t fq[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) fq[ks] = __builtin_bit_cast(s8_t, qraw[bq][ks]);
    f4_t s[NKF];
    float m = -INFINITY;
#pragma unroll
    for (int f = 0; f < NKF; ++f) {
      f4_t a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) a = MFMA16(frag_rows<HD>(Ks, f * 16, ks, t, g), fq[ks], a);
      if (f >= FIRST_PARTIAL) {  // only fragments that can hold padded keys for this (T bucket) are masked
#pragma unroll
        for (int r = 0; r < 4; ++r) a[r] = (f * 16 + r >= kthr) ? -INFINITY : a[r];
      }
      m = fmaxf(fmaxf(m, fmaxf(a[0], a[1])), fmaxf(a[2], a[3]));
      s[f] = a;
    }
    m = xrow_max(m);
    m *= c2;      // the scale (positive) is applied inside the exponent's fma: p = 2^(s c2 - max(s) c2), one multiply per row instead of one per score
    float l = 0.f;
#pragma unroll
    for (int f = 0; f < NKF; ++f)
#pragma unroll
      for (int r = 0; r < 4; ++r) { float p = __builtin_amdgcn_exp2f(fmaf(s[f][r], c2, -m)); s[f][r] = p; l += p; }
    l = xrow_sum(l);
    const float inv = 1.0f / l;
    s8_t fp[NKF / 2];
#pragma unroll
    for (int s2 = 0; s2 < NKF / 2; ++s2) fp[s2] = pack_pair(s[2 * s2], s[2 * s2 + 1]);
#pragma unroll
    for (int df = 0; df < DF; ++df) {
      f4_t o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s2 = 0; s2 < NKF / 2; ++s2) o = MFMA16(frag_cols_tr<HD>(Vs, 32 * s2, df * 16, t, g), fp[s2], o);
      int d = df * 16 + 4 * g;
      if (q < T && d < hd) {
        const long long at = (row0 + q) * D + h * hd + d;
        st4<bf16_t>(out + at, o * inv);
        if (EMIT) *reinterpret_cast<unsigned*>(em.q + at) = fp8_pack4(round4<bf16_t>(o * inv), qs, qmax, em.fmt, qseen);
      }
    }
    if (g == 0 && q < T) lse[((long long)b * H + h) * T + q] = (m + log2f(l)) * LN2;
  }
  if (EMIT) fp8_emit_amax(em, qseen, threadIdx.x & 63);
}

// ------------------------------------------------------------------------------------------ bf16 backward
// four gradient values of dqkv: bf16, and — EMIT — the fp8 byte copy at the same element offset (see attn_fwd_bf16)
template <bool EMIT>
__device__ __forceinline__ void st4_dqkv(bf16_t* dqkv, long long at, f4_t v, const Fp8Emit& em, float qs, float qmax, float& qseen) {
  if (!EMIT || dqkv) st4<bf16_t>(dqkv + at, v);   // (EMIT with dqkv == null: the fp8 copy is the only output — fp8 mode, where nothing reads the bf16 tensor)
  if (EMIT) *reinterpret_cast<unsigned*>(em.q + at) = fp8_pack4(round4<bf16_t>(v), qs, qmax, em.fmt, qseen);
}
template <int HD, int NKF, bool EMIT = false>
__global__ __launch_bounds__(256) void attn_bwd_bf16(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ out,
                                                     const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                     bf16_t* __restrict__ dqkv, int T, int H, int D, int hd, float scale, Fp8Emit em) {
  constexpr int TP = NKF * 16, KS = HD / 32, DF = HD / 16, IMG = TP * AttnLds<HD>::STRIDE;
  __shared__ __attribute__((aligned(16))) char smem[4 * IMG + 2 * TP * 4];
  char* Qs = smem; char* Ks = smem + IMG; char* Vs = smem + 2 * IMG; char* Gs = smem + 3 * IMG;  // Gs = dO
  float* lse2 = reinterpret_cast<float*>(smem + 4 * IMG);  // log2-domain LSE, +inf on padded rows
  float* dl = lse2 + TP;                                     // D_i = sum_d dO_i . O_i
  const int wid = pair_remap<HD>(blockIdx.x, gridDim.x);
  const int b = wid / H, h = wid - b * H;
  const long long row0 = (long long)b * T;
  const int ld = 3 * D;
  {
    HeadStager<HD, TP, 256, 4> sg;
    sg.load(0, qkv, row0, ld, h * hd, T, hd); sg.load(1, qkv, row0, ld, D + h * hd, T, hd);
    sg.load(2, qkv, row0, ld, 2 * D + h * hd, T, hd); sg.load(3, dout, row0, D, h * hd, T, hd);
    sg.store(0, Qs); sg.store(1, Ks); sg.store(2, Vs); sg.store(3, Gs);
  }
  for (int r = threadIdx.x; r < TP; r += blockDim.x) {
    float acc = 0.f, l2 = INFINITY;
    if (r < T) {
      l2 = lse[((long long)b * H + h) * T + r] * LOG2E;
      const bf16_t* o = out + (row0 + r) * D + h * hd;
      const bf16_t* gg = dout + (row0 + r) * D + h * hd;
      uint4 ov[HD / 8], gv[HD / 8];
#pragma unroll
      for (int c = 0; c < HD / 8; ++c) {
        ov[c] = c * 8 < hd ? *reinterpret_cast<const uint4*>(o + c * 8) : make_uint4(0, 0, 0, 0);
        gv[c] = c * 8 < hd ? *reinterpret_cast<const uint4*>(gg + c * 8) : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int c = 0; c < HD / 8; ++c) {
        const unsigned ow[4] = {ov[c].x, ov[c].y, ov[c].z, ov[c].w}, gw[4] = {gv[c].x, gv[c].y, gv[c].z, gv[c].w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
          acc += __uint_as_float(ow[k] << 16) * __uint_as_float(gw[k] << 16) + __uint_as_float(ow[k] & 0xffff0000u) * __uint_as_float(gw[k] & 0xffff0000u);
      }
    }
    lse2[r] = l2; dl[r] = acc;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, t = lane & 15, g = lane >> 4;
  float qmax = 0.f, qseen = 0.f;
  const float qs = EMIT ? fp8_emit_scale(em, lane, qmax) : 1.f;   // (EMIT: the fp8 copy of dqkv, see attn_fwd_bf16)
  const float c2 = scale * LOG2E;
  const int nblk = (T + 15) >> 4;
  // ---- pass 1: query rows -> dQ.  lane owns query column q; registers run over keys.
  for (int qb = w; qb < nblk; qb += 4) {
    const int q = qb * 16 + t;
    s8_t fq[KS], fg[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) { fq[ks] = frag_rows<HD>(Qs, qb * 16, ks, t, g); fg[ks] = frag_rows<HD>(Gs, qb * 16, ks, t, g); }
    const float my_l2 = lse2[q], my_dl = dl[q];
    s8_t fds[NKF / 2];
    f4_t prev = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int f = 0; f < NKF; ++f) {
      f4_t a = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        a = MFMA16(frag_rows<HD>(Ks, f * 16, ks, t, g), fq[ks], a);
        dp = MFMA16(frag_rows<HD>(Vs, f * 16, ks, t, g), fg[ks], dp);
      }
      f4_t ds;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float p = __builtin_amdgcn_exp2f(a[r] * c2 - my_l2);
        ds[r] = p * (dp[r] - my_dl) * scale;
      }
      // no key masking needed here: padded K rows are zero, so whatever dS holds for a padded key adds 0 to dQ
      if (f & 1) fds[f >> 1] = pack_pair(prev, ds); else prev = ds;
    }
#pragma unroll
    for (int df = 0; df < DF; ++df) {
      f4_t o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s2 = 0; s2 < NKF / 2; ++s2) o = MFMA16(frag_cols_tr<HD>(Ks, 32 * s2, df * 16, t, g), fds[s2], o);
      int d = df * 16 + 4 * g;
      if (q < T && d < hd) st4_dqkv<EMIT>(dqkv, (row0 + q) * ld + h * hd + d, o, em, qs, qmax, qseen);
    }
  }
  // ---- pass 2: key columns -> dK, dV.  lane owns key column; registers run over queries.
  for (int kb = w; kb < nblk; kb += 4) {
    const int key = kb * 16 + t;
    s8_t fk[KS], fv[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) { fk[ks] = frag_rows<HD>(Ks, kb * 16, ks, t, g); fv[ks] = frag_rows<HD>(Vs, kb * 16, ks, t, g); }
    s8_t fp[NKF / 2], fds[NKF / 2];
    f4_t prev_p = {0.f, 0.f, 0.f, 0.f}, prev_ds = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int f = 0; f < NKF; ++f) {
      f4_t a = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        a = MFMA16(frag_rows<HD>(Qs, f * 16, ks, t, g), fk[ks], a);
        dp = MFMA16(frag_rows<HD>(Gs, f * 16, ks, t, g), fv[ks], dp);
      }
      f4_t l4 = *reinterpret_cast<const f4_t*>(lse2 + f * 16 + 4 * g);
      f4_t d4 = *reinterpret_cast<const f4_t*>(dl + f * 16 + 4 * g);
      f4_t p, ds;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        p[r] = __builtin_amdgcn_exp2f(a[r] * c2 - l4[r]);  // padded queries have lse2 = +inf -> p = 0
        ds[r] = p[r] * (dp[r] - d4[r]) * scale;
      }
      if (f & 1) { fp[f >> 1] = pack_pair(prev_p, p); fds[f >> 1] = pack_pair(prev_ds, ds); } else { prev_p = p; prev_ds = ds; }
    }
#pragma unroll
    for (int df = 0; df < DF; ++df) {
      f4_t ok = {0.f, 0.f, 0.f, 0.f}, ov = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s2 = 0; s2 < NKF / 2; ++s2) {
        ok = MFMA16(frag_cols_tr<HD>(Qs, 32 * s2, df * 16, t, g), fds[s2], ok);
        ov = MFMA16(frag_cols_tr<HD>(Gs, 32 * s2, df * 16, t, g), fp[s2], ov);
      }
      int d = df * 16 + 4 * g;
      if (key < T && d < hd) {
        st4_dqkv<EMIT>(dqkv, (row0 + key) * ld + D + h * hd + d, ok, em, qs, qmax, qseen);
        st4_dqkv<EMIT>(dqkv, (row0 + key) * ld + 2 * D + h * hd + d, ov, em, qs, qmax, qseen);
      }
    }
  }
  if (EMIT) fp8_emit_amax(em, qseen, lane);
}

// ------------------------------------------------------------------------------------------ bf16 backward, single pass
// One workgroup (4 waves) per (sample, head).  A wave owns PAIRS of 16-key tiles (32 keys: pair w in the first sweep, pair w + 4
// in the second) and walks all query-tile pairs once per sweep:
//   S, dP  : lane <-> key, registers <-> queries  -> P and dS are directly the k-packed operands of the contractions over queries
//   dV, dK : accumulate in the wave's registers (it owns those keys)
//   dQ     : contracts over keys, i.e. needs dS with lane <-> query: each dS tile takes one 512-B round trip through a wave-private
//            LDS patch (ds_write_b64, ds_read_b64_tr_b16) instead of recomputing S, dP and the exponentials in a second pass.
//            The partial dQ of a (query pair, key pair) is added into an fp32 LDS accumulator with plain read-modify-write: in
//            step s wave w works on query pair (w + s) mod NQ, so no two waves touch the same rows between two barriers.
//            (LDS float atomics were measured at ~700 clk per ds_add_f32 wave-instruction here: 3x slower than two passes.)
// Only Q and dO are staged for the whole head; K and V fragments come straight from global memory (each key is needed by one
// wave), K^T through the wave's patch.  78 KiB of LDS and < 256 registers: two workgroups per CU, so one head's loads overlap the
// other's arithmetic — the two-pass kernel above needs 432 registers and 73 KiB and runs one workgroup per CU, latency-bound.
template <int HD, int NKF>
struct AttnBwd1p {
  static constexpr int NP = NKF / 2, TP = NKF * 16, IMG = TP * AttnLds<HD>::STRIDE, QS = HD + 4;
  static constexpr int XB = 32 * HD * 2;  // per wave: 4 dS tiles (bf16 [16 keys][16 queries]) or the wave's 32 K rows
  static constexpr int LDS = 2 * IMG + TP * QS * 4 + 4 * XB + 2 * TP * 4;
  static constexpr int LDS8 = LDS + 4 * XB;   // eight-wave form of attn_bwd1p_bf16: eight patches
};
template <int HD>  // transposed fragment from an unpadded [rows][HD] bf16 patch (same k order as frag_cols_tr)
__device__ __forceinline__ s8_t patch_cols_tr(const char* img, int c0, int t, int g) {
  const char* p = img + (4 * g + (t >> 2)) * (HD * 2) + (c0 + (t & 3) * 4) * 2;
  s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s4_t, p));
  s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s4_t, p + 16 * HD * 2));
  return join_s4(lo, hi);
}
// phase timestamps of one workgroup (tuning aid, only in -DATTN_TIMING builds: tools/attn_phase_timing.py)
#ifdef ATTN_TIMING
__device__ unsigned long long g_attn_ts[16];
extern "C" int csmae_debug_attn_ts(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_attn_ts), sizeof(g_attn_ts)); }
#define TS(i) do { if (blockIdx.x == 300 && threadIdx.x == 0) g_attn_ts[i] = __builtin_readcyclecounter(); } while (0)
#else
#define TS(i)
#endif
// NW = 8 (512 threads, one workgroup per CU): for sequences of nine key pairs and more (257 tokens: ViT-L/16 at 256^2, ViT-H/14 — BASELINE.json
// configs[3] / [4]).  Their head image (98 KiB at head_dim 32) admits ONE workgroup per CU whatever its width, and four waves walked the query
// pairs three times (pairs w, w + 4, w + 8: 27 barrier-separated steps, the third sweep with one busy wave): eight waves walk them twice
// (18 steps) with twice the waves in flight.  The rotation (wave w updates query pair (w + s) mod nq in step s) needs NW <= nq.
template <int HD, int NKF, bool EMIT = false, int NW = 4>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void attn_bwd1p_bf16(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ out,
                                                           const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                           bf16_t* __restrict__ dqkv, int T, int H, int D, int hd, float scale, Fp8Emit em) {
  using L = AttnBwd1p<HD, NKF>;
  constexpr int TP = L::TP, NP = L::NP, KS = HD / 32, DF = HD / 16, IMG = L::IMG, QS = L::QS;
  static_assert(NW == 4 || (NW == 8 && NP >= 8), "eight waves need at least eight query pairs for the dQ rotation");
  __shared__ __attribute__((aligned(16))) char smem[NW == 4 ? L::LDS : L::LDS8];
  char* Qs = smem; char* Gs = smem + IMG;                                // Gs = dO
  float* dqa = reinterpret_cast<float*>(smem + 2 * IMG);                 // dQ accumulator [TP][QS]
  char* xall = smem + 2 * IMG + TP * QS * 4;
  float* lse2 = reinterpret_cast<float*>(xall + NW * L::XB);             // log2-domain LSE, +inf on padded rows
  float* dl = lse2 + TP;                                                 // D_i = sum_d dO_i . O_i
  const int wid = pair_remap<HD>(blockIdx.x, gridDim.x);
  const int b = wid / H, h = wid - b * H;
  const long long row0 = (long long)b * T;
  const int ld = 3 * D;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, t = lane & 15, g = lane >> 4;
  float qmax = 0.f, qseen = 0.f;
  const float qs = EMIT ? fp8_emit_scale(em, lane, qmax) : 1.f;   // (EMIT: the fp8 copy of dqkv, see attn_fwd_bf16)
  TS(0);
  HeadStager<HD, TP, NW * 64, 2> sg;
  sg.load(0, qkv, row0, ld, h * hd, T, hd); sg.load(1, dout, row0, D, h * hd, T, hd);
  // K / V rows of a wave's key pair, straight from global memory.  The loads of sweep 0 go out with the staging loads above, those
  // of the next sweep right after a sweep's first barrier: their latency (4k clk each, measured) stays off the critical path.
  const int nkp = (T + 31) >> 5;            // key pairs that hold real rows
  uint4 kraw[2][KS], vraw[2][KS];
  auto load_kv = [&](int kp) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int key = 32 * kp + 16 * jj + t, col = ks * 32 + 8 * g;
        kraw[jj][ks] = make_uint4(0, 0, 0, 0); vraw[jj][ks] = make_uint4(0, 0, 0, 0);
        if (kp < nkp && key < T && col < hd) {
          kraw[jj][ks] = *reinterpret_cast<const uint4*>(qkv + (row0 + key) * ld + D + h * hd + col);
          vraw[jj][ks] = *reinterpret_cast<const uint4*>(qkv + (row0 + key) * ld + 2 * D + h * hd + col);
        }
      }
  };
  load_kv(w);
  for (int e = threadIdx.x; e < TP * QS / 4; e += blockDim.x) reinterpret_cast<f4_t*>(dqa)[e] = f4_t{0.f, 0.f, 0.f, 0.f};
  for (int r = threadIdx.x; r < TP; r += blockDim.x) {
    float acc = 0.f, l2 = INFINITY;
    if (r < T) {
      l2 = lse[((long long)b * H + h) * T + r] * LOG2E;
      const bf16_t* o = out + (row0 + r) * D + h * hd;
      const bf16_t* gg = dout + (row0 + r) * D + h * hd;
      uint4 ov[HD / 8], gv[HD / 8];
#pragma unroll
      for (int c = 0; c < HD / 8; ++c) {
        ov[c] = c * 8 < hd ? *reinterpret_cast<const uint4*>(o + c * 8) : make_uint4(0, 0, 0, 0);
        gv[c] = c * 8 < hd ? *reinterpret_cast<const uint4*>(gg + c * 8) : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int c = 0; c < HD / 8; ++c) {
        const unsigned ow[4] = {ov[c].x, ov[c].y, ov[c].z, ov[c].w}, gw[4] = {gv[c].x, gv[c].y, gv[c].z, gv[c].w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
          acc += __uint_as_float(ow[k] << 16) * __uint_as_float(gw[k] << 16) + __uint_as_float(ow[k] & 0xffff0000u) * __uint_as_float(gw[k] & 0xffff0000u);
      }
    }
    lse2[r] = l2; dl[r] = acc;
  }
  TS(1);
  sg.store(0, Qs); sg.store(1, Gs);  // (after the row loop: its global loads are in flight together with the staging loads)
  const float c2 = scale * LOG2E;
  char* xw = xall + w * L::XB;              // this wave's patch
  const int nq = (T + 31) >> 5;             // query-tile pairs that hold real rows
  for (int sweep = 0; sweep * NW < nkp; ++sweep) {
    const int kp = sweep * NW + w;
    const bool active = kp < nkp;           // (wave-uniform) the last sweep may have fewer pairs than waves
    const int k0 = 32 * kp;
    // K, V rows of the two key tiles straight from global memory; K^T (k = keys) through the patch
    s8_t fk[2][KS], fv[2][KS], kT[DF];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int col = ks * 32 + 8 * g;
        fk[jj][ks] = __builtin_bit_cast(s8_t, kraw[jj][ks]); fv[jj][ks] = __builtin_bit_cast(s8_t, vraw[jj][ks]);
        *reinterpret_cast<uint4*>(xw + (16 * jj + t) * (HD * 2) + col * 2) = kraw[jj][ks];
      }
#pragma unroll
    for (int df = 0; df < DF; ++df) kT[df] = patch_cols_tr<HD>(xw, df * 16, t, g);
    f4_t dk[2][DF], dv[2][DF];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int df = 0; df < DF; ++df) { dk[jj][df] = f4_t{0.f, 0.f, 0.f, 0.f}; dv[jj][df] = f4_t{0.f, 0.f, 0.f, 0.f}; }
    TS(2 + 4 * sweep);
    __syncthreads();                        // (first sweep: staging visible; later sweeps: previous sweep's last dQ update done)
    TS(3 + 4 * sweep);
    if ((sweep + 1) * NW < nkp) load_kv((sweep + 1) * NW + w);
    for (int s = 0; s < nq; ++s) {
      int ip = w + s; if (ip >= nq) ip -= nq;
      const int q0 = 32 * ip;
      if (active && w < nq) {
        f4_t p[2][2], ds[2][2];
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
          s8_t fq[KS], fg[KS];
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) { fq[ks] = frag_rows<HD>(Qs, q0 + 16 * ii, ks, t, g); fg[ks] = frag_rows<HD>(Gs, q0 + 16 * ii, ks, t, g); }
          const f4_t l4 = *reinterpret_cast<const f4_t*>(lse2 + q0 + 16 * ii + 4 * g);
          const f4_t d4 = *reinterpret_cast<const f4_t*>(dl + q0 + 16 * ii + 4 * g);
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            f4_t a = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) { a = MFMA16(fq[ks], fk[jj][ks], a); dp = MFMA16(fg[ks], fv[jj][ks], dp); }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              p[ii][jj][r] = __builtin_amdgcn_exp2f(a[r] * c2 - l4[r]);  // padded queries: lse2 = +inf -> p = 0
              ds[ii][jj][r] = p[ii][jj][r] * (dp[r] - d4[r]) * scale;
            }
            // dS tile -> patch [ii][jj][key t][query 4g..4g+3] (bf16)
            uint2 u = make_uint2(pack2bf(ds[ii][jj][0], ds[ii][jj][1]), pack2bf(ds[ii][jj][2], ds[ii][jj][3]));
            *reinterpret_cast<uint2*>(xw + (ii * 2 + jj) * 512 + t * 32 + g * 8) = u;
          }
        }
        // dK, dV: contraction over the 32 queries of the pair (k order {4g+e, 16+4g+e}: what pack_pair and frag_cols_tr share)
#pragma unroll
        for (int df = 0; df < DF; ++df) {
          const s8_t qT = frag_cols_tr<HD>(Qs, q0, df * 16, t, g), gT = frag_cols_tr<HD>(Gs, q0, df * 16, t, g);
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            dk[jj][df] = MFMA16(qT, pack_pair(ds[0][jj], ds[1][jj]), dk[jj][df]);
            dv[jj][df] = MFMA16(gT, pack_pair(p[0][jj], p[1][jj]), dv[jj][df]);
          }
        }
        // dQ: contraction over this wave's 32 keys; lane <-> query through the transposing read of the patch
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
          const char* xp = xw + ii * 1024 + (4 * g + (t >> 2)) * 32 + (t & 3) * 8;
          s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s4_t, xp));
          s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s4_t, xp + 512));
          const s8_t sT = join_s4(lo, hi);
          f4_t* dst = reinterpret_cast<f4_t*>(dqa + (q0 + 16 * ii + t) * QS + 4 * g);
#pragma unroll
          for (int df = 0; df < DF; ++df) dst[df * 4] = MFMA16(kT[df], sT, dst[df * 4]);
        }
      }
      __syncthreads();                      // the rows this wave updated belong to another wave in the next step
    }
    TS(4 + 4 * sweep);
    // dK, dV of the wave's keys
    if (active) {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int key = k0 + 16 * jj + t;
#pragma unroll
        for (int df = 0; df < DF; ++df) {
          const int d = df * 16 + 4 * g;
          if (key < T && d < hd) {
            st4_dqkv<EMIT>(dqkv, (row0 + key) * ld + D + h * hd + d, dk[jj][df], em, qs, qmax, qseen);
            st4_dqkv<EMIT>(dqkv, (row0 + key) * ld + 2 * D + h * hd + d, dv[jj][df], em, qs, qmax, qseen);
          }
        }
      }
    }
  }
  TS(10);
  for (int e = threadIdx.x; e < T * (HD / 4); e += blockDim.x) {
    const int q = e / (HD / 4), d = (e - q * (HD / 4)) * 4;
    if (d < hd) st4_dqkv<EMIT>(dqkv, (row0 + q) * ld + h * hd + d, *reinterpret_cast<const f4_t*>(dqa + q * QS + d), em, qs, qmax, qseen);
  }
  if (EMIT) fp8_emit_amax(em, qseen, lane);
}

// ---- the same pass with TWO key pairs per wave and a single sweep (NP <= 8 key pairs: every shape of the step).  The two-sweep form
// above walks the query pairs twice — 2 NP steps, each with its Q / dO fragment reads, one read-modify-write of the step's dQ rows in
// LDS and a workgroup barrier.  Here wave w owns key pairs w and w + 4 at once: one walk (NP steps and barriers), the Q / dO fragments
// of a step (row and transposed forms) are read once for both key pairs, and the step's dQ contribution of 64 keys is summed in
// registers before its single read-modify-write.
template <int HD, int NKF, bool EMIT = false>
__global__ __launch_bounds__(256, 2) void attn_bwd1p2_bf16(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ out,
                                                            const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                            bf16_t* __restrict__ dqkv, int T, int H, int D, int hd, float scale, Fp8Emit em) {
  using L = AttnBwd1p<HD, NKF>;
  constexpr int TP = L::TP, NP = L::NP, KS = HD / 32, DF = HD / 16, IMG = L::IMG, QS = L::QS;
  static_assert(NP <= 8, "two key pairs per wave cover at most eight");
  // NPW key pairs per wave; a pair beyond the sequence is skipped (computing it unconditionally — all-zero K / V, so that both chains
  // of a step share a basic block — measured slower: 193 vs 183 us, the idle wave's SIMD belongs to the other workgroup's waves)
  constexpr int NPW = NP > 4 ? 2 : 1;
  __shared__ __attribute__((aligned(16))) char smem[L::LDS];
  char* Qs = smem; char* Gs = smem + IMG;
  float* dqa = reinterpret_cast<float*>(smem + 2 * IMG);
  char* xall = smem + 2 * IMG + TP * QS * 4;
  float* lse2 = reinterpret_cast<float*>(xall + 4 * L::XB);
  float* dl = lse2 + TP;
  const int wid = pair_remap<HD>(blockIdx.x, gridDim.x);
  const int b = wid / H, h = wid - b * H;
  const long long row0 = (long long)b * T;
  const int ld = 3 * D;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, t = lane & 15, g = lane >> 4;
  float qmax = 0.f, qseen = 0.f;
  const float qs = EMIT ? fp8_emit_scale(em, lane, qmax) : 1.f;   // (EMIT: the fp8 copy of dqkv, see attn_fwd_bf16)
  TS(0);
  HeadStager<HD, TP, 256, 2> sg;
  sg.load(0, qkv, row0, ld, h * hd, T, hd); sg.load(1, dout, row0, D, h * hd, T, hd);
  const int nkp = (T + 31) >> 5;
  uint4 kraw[NPW][2][KS], vraw[NPW][2][KS];   // [pair][tile][k step]
#pragma unroll
  for (int pi = 0; pi < NPW; ++pi)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int kp = w + 4 * pi, key = 32 * kp + 16 * jj + t, col = ks * 32 + 8 * g;
        kraw[pi][jj][ks] = make_uint4(0, 0, 0, 0); vraw[pi][jj][ks] = make_uint4(0, 0, 0, 0);
        if (kp < nkp && key < T && col < hd) {
          kraw[pi][jj][ks] = *reinterpret_cast<const uint4*>(qkv + (row0 + key) * ld + D + h * hd + col);
          vraw[pi][jj][ks] = *reinterpret_cast<const uint4*>(qkv + (row0 + key) * ld + 2 * D + h * hd + col);
        }
      }
  for (int e = threadIdx.x; e < TP * QS / 4; e += blockDim.x) reinterpret_cast<f4_t*>(dqa)[e] = f4_t{0.f, 0.f, 0.f, 0.f};
  for (int r = threadIdx.x; r < TP; r += blockDim.x) {
    float acc = 0.f, l2 = INFINITY;
    if (r < T) {
      l2 = lse[((long long)b * H + h) * T + r] * LOG2E;
      const bf16_t* o = out + (row0 + r) * D + h * hd;
      const bf16_t* gg = dout + (row0 + r) * D + h * hd;
      uint4 ov[HD / 8], gv[HD / 8];
#pragma unroll
      for (int c = 0; c < HD / 8; ++c) {
        ov[c] = c * 8 < hd ? *reinterpret_cast<const uint4*>(o + c * 8) : make_uint4(0, 0, 0, 0);
        gv[c] = c * 8 < hd ? *reinterpret_cast<const uint4*>(gg + c * 8) : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int c = 0; c < HD / 8; ++c) {
        const unsigned ow[4] = {ov[c].x, ov[c].y, ov[c].z, ov[c].w}, gw[4] = {gv[c].x, gv[c].y, gv[c].z, gv[c].w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
          acc += __uint_as_float(ow[k] << 16) * __uint_as_float(gw[k] << 16) + __uint_as_float(ow[k] & 0xffff0000u) * __uint_as_float(gw[k] & 0xffff0000u);
      }
    }
    lse2[r] = l2; dl[r] = acc;
  }
  TS(1);
  sg.store(0, Qs); sg.store(1, Gs);
  const float c2 = scale * LOG2E;
  char* xw = xall + w * L::XB;
  const int nq = (T + 31) >> 5;
  const bool act0 = w < nkp, act1 = w + 4 < nkp;   // (wave-uniform)
  s8_t fk[NPW][2][KS], fv[NPW][2][KS], kT[NPW][DF];
  f4_t dk[NPW][2][DF], dv[NPW][2][DF];
#pragma unroll
  for (int pi = 0; pi < NPW; ++pi) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int col = ks * 32 + 8 * g;
        fk[pi][jj][ks] = __builtin_bit_cast(s8_t, kraw[pi][jj][ks]); fv[pi][jj][ks] = __builtin_bit_cast(s8_t, vraw[pi][jj][ks]);
        *reinterpret_cast<uint4*>(xw + (16 * jj + t) * (HD * 2) + col * 2) = kraw[pi][jj][ks];   // K rows -> patch -> K^T (the patch is wave-private: LDS operations of a wave stay in order)
      }
#pragma unroll
    for (int df = 0; df < DF; ++df) kT[pi][df] = patch_cols_tr<HD>(xw, df * 16, t, g);
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int df = 0; df < DF; ++df) { dk[pi][jj][df] = f4_t{0.f, 0.f, 0.f, 0.f}; dv[pi][jj][df] = f4_t{0.f, 0.f, 0.f, 0.f}; }
  }
  TS(2);
  __syncthreads();                          // staging, statistics and the cleared accumulator are visible
  TS(3);
  for (int s = 0; s < nq; ++s) {
    int ip = w + s; if (ip >= nq) ip -= nq;
    const int q0 = 32 * ip;
    if (act0) {
      s8_t fq[2][KS], fg[2][KS], qT[DF], gT[DF];
      f4_t l4[2], d4[2], dq[2][DF];
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) { fq[ii][ks] = frag_rows<HD>(Qs, q0 + 16 * ii, ks, t, g); fg[ii][ks] = frag_rows<HD>(Gs, q0 + 16 * ii, ks, t, g); }
        l4[ii] = *reinterpret_cast<const f4_t*>(lse2 + q0 + 16 * ii + 4 * g);
        d4[ii] = *reinterpret_cast<const f4_t*>(dl + q0 + 16 * ii + 4 * g);
#pragma unroll
        for (int df = 0; df < DF; ++df) dq[ii][df] = f4_t{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int df = 0; df < DF; ++df) { qT[df] = frag_cols_tr<HD>(Qs, q0, df * 16, t, g); gT[df] = frag_cols_tr<HD>(Gs, q0, df * 16, t, g); }
#pragma unroll
      for (int pi = 0; pi < NPW; ++pi) {
        if (pi == 1 && !act1) continue;
        uint2 pp[2][2], dsp[2][2];
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            f4_t a = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) { a = MFMA16(fq[ii][ks], fk[pi][jj][ks], a); dp = MFMA16(fg[ii][ks], fv[pi][jj][ks], dp); }
            f4_t p, ds;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              p[r] = __builtin_amdgcn_exp2f(a[r] * c2 - l4[ii][r]);  // padded queries: lse2 = +inf -> p = 0
              ds[r] = p[r] * (dp[r] - d4[ii][r]) * scale;
            }
            pp[ii][jj] = make_uint2(pack2bf(p[0], p[1]), pack2bf(p[2], p[3]));
            const uint2 u = make_uint2(pack2bf(ds[0], ds[1]), pack2bf(ds[2], ds[3]));
            dsp[ii][jj] = u;
            *reinterpret_cast<uint2*>(xw + (ii * 2 + jj) * 512 + t * 32 + g * 8) = u;   // dS tile -> patch [ii][jj][key t][query 4g..4g+3]
          }
#pragma unroll
        for (int df = 0; df < DF; ++df)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            dk[pi][jj][df] = MFMA16(qT[df], __builtin_bit_cast(s8_t, make_uint4(dsp[0][jj].x, dsp[0][jj].y, dsp[1][jj].x, dsp[1][jj].y)), dk[pi][jj][df]);
            dv[pi][jj][df] = MFMA16(gT[df], __builtin_bit_cast(s8_t, make_uint4(pp[0][jj].x, pp[0][jj].y, pp[1][jj].x, pp[1][jj].y)), dv[pi][jj][df]);
          }
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
          const char* xp = xw + ii * 1024 + (4 * g + (t >> 2)) * 32 + (t & 3) * 8;
          s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s4_t, xp));
          s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s4_t, xp + 512));
          const s8_t sT = join_s4(lo, hi);
#pragma unroll
          for (int df = 0; df < DF; ++df) dq[ii][df] = MFMA16(kT[pi][df], sT, dq[ii][df]);
        }
      }
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        f4_t* dst = reinterpret_cast<f4_t*>(dqa + (q0 + 16 * ii + t) * QS + 4 * g);
#pragma unroll
        for (int df = 0; df < DF; ++df) dst[df * 4] += dq[ii][df];
      }
    }
    __syncthreads();                        // the rows this wave updated belong to another wave in the next step
    if (s == 0) TS(5);
  }
  TS(4);
#pragma unroll
  for (int pi = 0; pi < NPW; ++pi) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int key = 32 * (w + 4 * pi) + 16 * jj + t;
#pragma unroll
      for (int df = 0; df < DF; ++df) {
        const int d = df * 16 + 4 * g;
        if (key < T && d < hd) {
          st4_dqkv<EMIT>(dqkv, (row0 + key) * ld + D + h * hd + d, dk[pi][jj][df], em, qs, qmax, qseen);
          st4_dqkv<EMIT>(dqkv, (row0 + key) * ld + 2 * D + h * hd + d, dv[pi][jj][df], em, qs, qmax, qseen);
        }
      }
    }
  }
  TS(10);
  for (int e = threadIdx.x; e < T * (HD / 4); e += blockDim.x) {
    const int q = e / (HD / 4), d = (e - q * (HD / 4)) * 4;
    if (d < hd) st4_dqkv<EMIT>(dqkv, (row0 + q) * ld + h * hd + d, *reinterpret_cast<const f4_t*>(dqa + q * QS + d), em, qs, qmax, qseen);
  }
  TS(11);
  if (EMIT) fp8_emit_amax(em, qseen, lane);
}

// ------------------------------------------------------------------------------------------ fp32 (parity mode)
template <int HD>
__global__ __launch_bounds__(256) void attn_fwd_f32(const float* __restrict__ qkv, float* __restrict__ out,
                                                    float* __restrict__ lse, int T, int H, int D, int hd, float scale) {
  extern __shared__ float sm[];
  const int S = hd + 1;
  float* Ks = sm; float* Vs = sm + T * S;
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const long long row0 = (long long)b * T;
  const int ld = 3 * D;
  for (int e = threadIdx.x; e < T * hd; e += blockDim.x) {
    int r = e / hd, d = e - r * hd;
    Ks[r * S + d] = qkv[(row0 + r) * ld + D + h * hd + d];
    Vs[r * S + d] = qkv[(row0 + r) * ld + 2 * D + h * hd + d];
  }
  __syncthreads();
  for (int q = threadIdx.x; q < T; q += blockDim.x) {
    float qr[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) qr[d] = d < hd ? qkv[(row0 + q) * ld + h * hd + d] * scale : 0.f;
    float m = -INFINITY;
    for (int j = 0; j < T; ++j) {
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < HD; ++d) if (d < hd) s = fmaf(qr[d], Ks[j * S + d], s);
      m = fmaxf(m, s);
    }
    float l = 0.f, o[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) o[d] = 0.f;
    for (int j = 0; j < T; ++j) {
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < HD; ++d) if (d < hd) s = fmaf(qr[d], Ks[j * S + d], s);
      float p = expf(s - m);
      l += p;
#pragma unroll
      for (int d = 0; d < HD; ++d) if (d < hd) o[d] = fmaf(p, Vs[j * S + d], o[d]);
    }
    float inv = 1.0f / l;
#pragma unroll
    for (int d = 0; d < HD; ++d) if (d < hd) out[(row0 + q) * D + h * hd + d] = o[d] * inv;
    lse[((long long)b * H + h) * T + q] = m + logf(l);
  }
}

template <int HD>
__global__ __launch_bounds__(256) void attn_bwd_f32(const float* __restrict__ qkv, const float* __restrict__ out,
                                                    const float* __restrict__ dout, const float* __restrict__ lse,
                                                    float* __restrict__ dqkv, int T, int H, int D, int hd, float scale) {
  extern __shared__ float sm[];
  const int S = hd + 1;
  float* Qs = sm; float* Ks = Qs + T * S; float* Vs = Ks + T * S; float* Gs = Vs + T * S;
  float* ls = Gs + T * S; float* dl = ls + T;
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const long long row0 = (long long)b * T;
  const int ld = 3 * D;
  for (int e = threadIdx.x; e < T * hd; e += blockDim.x) {
    int r = e / hd, d = e - r * hd;
    Qs[r * S + d] = qkv[(row0 + r) * ld + h * hd + d];
    Ks[r * S + d] = qkv[(row0 + r) * ld + D + h * hd + d];
    Vs[r * S + d] = qkv[(row0 + r) * ld + 2 * D + h * hd + d];
    Gs[r * S + d] = dout[(row0 + r) * D + h * hd + d];
  }
  for (int r = threadIdx.x; r < T; r += blockDim.x) {
    float acc = 0.f;
    for (int d = 0; d < hd; ++d) acc = fmaf(out[(row0 + r) * D + h * hd + d], dout[(row0 + r) * D + h * hd + d], acc);
    dl[r] = acc; ls[r] = lse[((long long)b * H + h) * T + r];
  }
  __syncthreads();
  for (int q = threadIdx.x; q < T; q += blockDim.x) {  // dQ
    float dq[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) dq[d] = 0.f;
    const float lq = ls[q], dlq = dl[q];
    for (int j = 0; j < T; ++j) {
      float s = 0.f, dp = 0.f;
      for (int d = 0; d < hd; ++d) { s = fmaf(Qs[q * S + d], Ks[j * S + d], s); dp = fmaf(Gs[q * S + d], Vs[j * S + d], dp); }
      float ds = expf(s * scale - lq) * (dp - dlq) * scale;
#pragma unroll
      for (int d = 0; d < HD; ++d) if (d < hd) dq[d] = fmaf(ds, Ks[j * S + d], dq[d]);
    }
#pragma unroll
    for (int d = 0; d < HD; ++d) if (d < hd) dqkv[(row0 + q) * ld + h * hd + d] = dq[d];
  }
  for (int j = threadIdx.x; j < T; j += blockDim.x) {  // dK, dV
    float dk[HD], dv[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) { dk[d] = 0.f; dv[d] = 0.f; }
    for (int i = 0; i < T; ++i) {
      float s = 0.f, dp = 0.f;
      for (int d = 0; d < hd; ++d) { s = fmaf(Qs[i * S + d], Ks[j * S + d], s); dp = fmaf(Gs[i * S + d], Vs[j * S + d], dp); }
      float p = expf(s * scale - ls[i]);
      float ds = p * (dp - dl[i]) * scale;
#pragma unroll
      for (int d = 0; d < HD; ++d) if (d < hd) { dk[d] = fmaf(ds, Qs[i * S + d], dk[d]); dv[d] = fmaf(p, Gs[i * S + d], dv[d]); }
    }
#pragma unroll
    for (int d = 0; d < HD; ++d) if (d < hd) {
      dqkv[(row0 + j) * ld + D + h * hd + d] = dk[d];
      dqkv[(row0 + j) * ld + 2 * D + h * hd + d] = dv[d];
    }
  }
}

// ------------------------------------------------------------------------------------------ any length (fallback)
// Sequences that do not fit the LDS-resident kernels above (`--input_size 320` and up: T > 288; the reference accepts any
// --input_size / --patch_size, main_pretrain.py:70-86; fp32 parity mode from T (hd + 1) 16 B > 160 KiB).  Same arithmetic as the fp32
// kernels — one thread per query row / key column, fp32 accumulation, exact softmax — with K, V, Q and dO rows read from global memory
// (every thread of a wave reads the same row: one transaction per load) and only the per-row statistics in LDS.  Outside the surveyed
// geometries (T <= 257), so correctness — not speed — is what it is for: ~10-30x slower than the MFMA kernels per FLOP.
template <typename T, int HD>
__global__ __launch_bounds__(256) void attn_fwd_any(const T* __restrict__ qkv, T* __restrict__ out, float* __restrict__ lse, int Tn, int H, int D, int hd, float scale) {
  const int b = blockIdx.y / H, h = blockIdx.y - b * H;
  const long long row0 = (long long)b * Tn;
  const int ld = 3 * D;
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= Tn) return;
  float qr[HD], o[HD];
#pragma unroll
  for (int d = 0; d < HD; d += 4) {
    const f4_t v = d < hd ? ld4<T>(qkv + (row0 + q) * ld + h * hd + d) * scale : f4_t{0.f, 0.f, 0.f, 0.f};
    qr[d] = v[0]; qr[d + 1] = v[1]; qr[d + 2] = v[2]; qr[d + 3] = v[3];
    o[d] = o[d + 1] = o[d + 2] = o[d + 3] = 0.f;
  }
  float m = -INFINITY, l = 0.f;
  for (int j = 0; j < Tn; ++j) {
    const T* kr = qkv + (row0 + j) * ld + D + h * hd;
    float sc = 0.f;
#pragma unroll
    for (int d = 0; d < HD; d += 4) if (d < hd) { const f4_t k4 = ld4<T>(kr + d); sc += (qr[d] * k4[0] + qr[d + 1] * k4[1]) + (qr[d + 2] * k4[2] + qr[d + 3] * k4[3]); }
    const float mn = fmaxf(m, sc), c = expf(m - mn), pj = expf(sc - mn);   // online softmax (expf(-inf) = 0 on the first key)
    l = l * c + pj; m = mn;
    const T* vr = kr + D;
#pragma unroll
    for (int d = 0; d < HD; d += 4) if (d < hd) { const f4_t v4 = ld4<T>(vr + d); o[d] = o[d] * c + pj * v4[0]; o[d + 1] = o[d + 1] * c + pj * v4[1]; o[d + 2] = o[d + 2] * c + pj * v4[2]; o[d + 3] = o[d + 3] * c + pj * v4[3]; }
  }
  const float inv = 1.0f / l;
#pragma unroll
  for (int d = 0; d < HD; d += 4) if (d < hd) st4<T>(out + (row0 + q) * D + h * hd + d, f4_t{o[d] * inv, o[d + 1] * inv, o[d + 2] * inv, o[d + 3] * inv});
  lse[((long long)b * H + h) * Tn + q] = m + logf(l);
}

template <typename T, int HD>
__global__ __launch_bounds__(256) void attn_bwd_any(const T* __restrict__ qkv, const T* __restrict__ out, const T* __restrict__ dout, const float* __restrict__ lse,
                                                    T* __restrict__ dqkv, int Tn, int H, int D, int hd, float scale) {
  extern __shared__ float sm[];
  float* ls = sm; float* dl = sm + Tn;
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const long long row0 = (long long)b * Tn;
  const int ld = 3 * D;
  for (int r = threadIdx.x; r < Tn; r += blockDim.x) {
    float acc = 0.f;
    for (int d = 0; d < hd; d += 4) { const f4_t a = ld4<T>(out + (row0 + r) * D + h * hd + d), g4 = ld4<T>(dout + (row0 + r) * D + h * hd + d); acc += (a[0] * g4[0] + a[1] * g4[1]) + (a[2] * g4[2] + a[3] * g4[3]); }
    dl[r] = acc; ls[r] = lse[((long long)b * H + h) * Tn + r];
  }
  __syncthreads();
  auto dot = [&](const float (&a)[HD], const T* row) {
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < HD; d += 4) if (d < hd) { const f4_t v = ld4<T>(row + d); s += (a[d] * v[0] + a[d + 1] * v[1]) + (a[d + 2] * v[2] + a[d + 3] * v[3]); }
    return s;
  };
  auto load_row = [&](float (&a)[HD], const T* row) {
#pragma unroll
    for (int d = 0; d < HD; d += 4) { const f4_t v = d < hd ? ld4<T>(row + d) : f4_t{0.f, 0.f, 0.f, 0.f}; a[d] = v[0]; a[d + 1] = v[1]; a[d + 2] = v[2]; a[d + 3] = v[3]; }
  };
  auto axpy = [&](float (&a)[HD], float w, const T* row) {
#pragma unroll
    for (int d = 0; d < HD; d += 4) if (d < hd) { const f4_t v = ld4<T>(row + d); a[d] = fmaf(w, v[0], a[d]); a[d + 1] = fmaf(w, v[1], a[d + 1]); a[d + 2] = fmaf(w, v[2], a[d + 2]); a[d + 3] = fmaf(w, v[3], a[d + 3]); }
  };
  auto store_row = [&](const float (&a)[HD], T* row) {
#pragma unroll
    for (int d = 0; d < HD; d += 4) if (d < hd) st4<T>(row + d, f4_t{a[d], a[d + 1], a[d + 2], a[d + 3]});
  };
  for (int q = threadIdx.x; q < Tn; q += blockDim.x) {  // dQ: thread = query row
    float qi[HD], gi[HD], dq[HD];
    load_row(qi, qkv + (row0 + q) * ld + h * hd); load_row(gi, dout + (row0 + q) * D + h * hd);
#pragma unroll
    for (int d = 0; d < HD; ++d) dq[d] = 0.f;
    const float lq = ls[q], dlq = dl[q];
    for (int j = 0; j < Tn; ++j) {
      const T* kr = qkv + (row0 + j) * ld + D + h * hd;
      const float ds = expf(dot(qi, kr) * scale - lq) * (dot(gi, kr + D) - dlq) * scale;
      axpy(dq, ds, kr);
    }
    store_row(dq, dqkv + (row0 + q) * ld + h * hd);
  }
  for (int j = threadIdx.x; j < Tn; j += blockDim.x) {  // dK, dV: thread = key column
    float kj[HD], vj[HD], dk[HD], dv[HD];
    load_row(kj, qkv + (row0 + j) * ld + D + h * hd); load_row(vj, qkv + (row0 + j) * ld + 2 * D + h * hd);
#pragma unroll
    for (int d = 0; d < HD; ++d) { dk[d] = 0.f; dv[d] = 0.f; }
    for (int i = 0; i < Tn; ++i) {
      const T* qr = qkv + (row0 + i) * ld + h * hd;
      const T* gr = dout + (row0 + i) * D + h * hd;
      const float pij = expf(dot(kj, qr) * scale - ls[i]);
      const float ds = pij * (dot(vj, gr) - dl[i]) * scale;
      axpy(dk, ds, qr); axpy(dv, pij, gr);
    }
    store_row(dk, dqkv + (row0 + j) * ld + D + h * hd);
    store_row(dv, dqkv + (row0 + j) * ld + 2 * D + h * hd);
  }
}
template <typename T>
static void launch_any(bool bwd, long long B, int Tn, int H, int D, int hd, float scale, const void* qkv, const void* out, const void* dout, float* lse_w, const float* lse_r,
                       void* o, hipStream_t st) {
  const int BH = (int)(B * H);
#define ANY(HDV)                                                                                                                                     \
  if (!bwd) hipLaunchKernelGGL((attn_fwd_any<T, HDV>), dim3(cdiv(Tn, 256), BH), dim3(256), 0, st, (const T*)qkv, (T*)o, lse_w, Tn, H, D, hd, scale);   \
  else hipLaunchKernelGGL((attn_bwd_any<T, HDV>), dim3(BH), dim3(256), (size_t)2 * Tn * sizeof(float), st, (const T*)qkv, (const T*)out, (const T*)dout, lse_r, (T*)o, Tn, H, D, hd, scale)
  if (hd <= 32) { ANY(32); } else if (hd <= 64) { ANY(64); } else if (hd <= 96) { ANY(96); } else { ANY(128); }
#undef ANY
}

// ------------------------------------------------------------------------------------------ dispatch
template <int HD, int NKF>
static void launch_fwd_bf16(int BH, const void* qkv, void* out, float* lse, int T, int H, int D, int hd, float scale, hipStream_t st, const Fp8Emit* em) {
  if (em) hipLaunchKernelGGL((attn_fwd_bf16<HD, NKF, true>), dim3(BH), dim3(256), 0, st, (const bf16_t*)qkv, (bf16_t*)out, lse, T, H, D, hd, scale, *em);
  else hipLaunchKernelGGL((attn_fwd_bf16<HD, NKF, false>), dim3(BH), dim3(256), 0, st, (const bf16_t*)qkv, (bf16_t*)out, lse, T, H, D, hd, scale, Fp8Emit{});
}
template <int HD, int NKF>
static void launch_bwd_bf16(int BH, const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int T, int H, int D, int hd, float scale, hipStream_t st,
                            const Fp8Emit* em) {
  static const bool two_pass = csmae_debug_opt("attn_bwd_2pass") != nullptr;  // tuning aid: the older two-pass kernel
  static const bool two_sweeps = csmae_debug_opt("attn_bwd_2sweep") != nullptr;  // tuning aid: one key pair per wave and sweep (the first single-pass version)
  constexpr bool KP2_OK = HD <= 32 && NKF >= 6 && NKF <= 16 && AttnBwd1p<HD, NKF>::LDS <= 160 * 1024;
  constexpr int NK1 = AttnBwd1p<HD, NKF>::LDS <= 160 * 1024 ? NKF : 2;
  constexpr bool W8_OK = NKF >= 18 && AttnBwd1p<HD, NKF>::LDS8 <= 160 * 1024;
  static const bool four_waves = csmae_debug_opt("attn_bwd_4waves") != nullptr;   // tuning aid: the four-wave form for long sequences as well
#define ATTN_BWD_ARGS (const bf16_t*)qkv, (const bf16_t*)out, (const bf16_t*)dout, lse, (bf16_t*)dqkv, T, H, D, hd, scale
  if (KP2_OK && !two_pass && !two_sweeps) {
    if (em) CSMAE_LAUNCH((attn_bwd1p2_bf16<HD, (KP2_OK ? NKF : 2), true>), dim3(BH), dim3(256), 0, st, ATTN_BWD_ARGS, *em);
    else CSMAE_LAUNCH((attn_bwd1p2_bf16<HD, (KP2_OK ? NKF : 2), false>), dim3(BH), dim3(256), 0, st, ATTN_BWD_ARGS, Fp8Emit{});
  } else if (W8_OK && !two_pass && !four_waves) {   // nine key pairs and more (257 tokens): eight waves, two sweeps
    if constexpr (W8_OK) {
      if (em) CSMAE_LAUNCH((attn_bwd1p_bf16<HD, NKF, true, 8>), dim3(BH), dim3(512), 0, st, ATTN_BWD_ARGS, *em);
      else CSMAE_LAUNCH((attn_bwd1p_bf16<HD, NKF, false, 8>), dim3(BH), dim3(512), 0, st, ATTN_BWD_ARGS, Fp8Emit{});
    }
  } else if (AttnBwd1p<HD, NKF>::LDS <= 160 * 1024 && NKF >= 6 && !two_pass) {  // (<= 64 tokens: fewer key pairs than waves, the two-pass split is faster)
    if (em) CSMAE_LAUNCH((attn_bwd1p_bf16<HD, NK1, true>), dim3(BH), dim3(256), 0, st, ATTN_BWD_ARGS, *em);
    else CSMAE_LAUNCH((attn_bwd1p_bf16<HD, NK1, false>), dim3(BH), dim3(256), 0, st, ATTN_BWD_ARGS, Fp8Emit{});
  } else {
    if (em) CSMAE_LAUNCH((attn_bwd_bf16<HD, NKF, true>), dim3(BH), dim3(256), 0, st, ATTN_BWD_ARGS, *em);
    else CSMAE_LAUNCH((attn_bwd_bf16<HD, NKF, false>), dim3(BH), dim3(256), 0, st, ATTN_BWD_ARGS, Fp8Emit{});
  }
#undef ATTN_BWD_ARGS
}

// (head_dim bucket, key-fragment count) combinations whose LDS images fit 160 KiB in the backward kernel
#define NKF_SMALL(HDV, CALL)                                 \
  if (T <= 32) { CALL(HDV, 2); }                             \
  else if (T <= 64) { CALL(HDV, 4); }                        \
  else if (T <= 96) { CALL(HDV, 6); }
#define DISPATCH_BF16(CALL)                                                            \
  if (hd <= 32) { NKF_SMALL(32, CALL) else if (T <= 224) { CALL(32, 14); } else { CALL(32, 18); } } \
  else if (hd <= 64) { NKF_SMALL(64, CALL) else { CALL(64, 14); } }                    \
  else { NKF_SMALL(96, CALL) }

// (T, head_dim) pairs the LDS-resident MFMA kernels cover; anything else with head_dim <= 128, T <= 8192 runs the any-length kernels
static bool bf16_resident(int T, int hd) { return hd % 8 == 0 && T <= 288 && (hd <= 32 || (hd <= 64 && T <= 224) || (hd <= 96 && T <= 96)); }
static int check_common(const char* who, long long B, int T, int H, int D, int hd) {
  CSMAE_REQUIRE(B > 0 && T > 0 && H > 0 && hd > 0 && D == H * hd, "%s: bad geometry B=%lld T=%d H=%d D=%d hd=%d", who, B, T, H, D, hd);
  CSMAE_REQUIRE(hd <= 128 && hd % 4 == 0 && T <= 8192 && B * H <= 0x7fffffffll, "%s: head_dim %d (multiple of 4, <= 128) / sequence length %d (<= 8192) unsupported", who, hd, T);
  return CSMAE_OK;
}

static int attn_fwd_impl(int dtype, long long B, int T, int H, int hd, const void* qkv, void* out, float* lse, void* stream, const Fp8Emit* em) {
  const int D = H * hd;
  int rc = check_common("csmae_attn_fwd", B, T, H, D, hd);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const float scale = 1.0f / sqrtf((float)hd);
  const int BH = (int)(B * H);
  if (dtype == CSMAE_BF16) {
    if (!bf16_resident(T, hd)) { CSMAE_REQUIRE(!em, "csmae_attn_fwd_q: the fp8 copy is emitted by the LDS-resident kernels only (csmae_attn_resident)"); launch_any<bf16_t>(false, B, T, H, D, hd, scale, qkv, nullptr, nullptr, lse, nullptr, out, st); return csmae_check_launch("csmae_attn_fwd"); }
#define CALLF(HDV, NK) launch_fwd_bf16<HDV, NK>(BH, qkv, out, lse, T, H, D, hd, scale, st, em)
    DISPATCH_BF16(CALLF)
#undef CALLF
  } else if (dtype == CSMAE_F32) {
    size_t sh = (size_t)2 * T * (hd + 1) * sizeof(float);
    if (sh > 160 * 1024 || hd > 96) { launch_any<float>(false, B, T, H, D, hd, scale, qkv, nullptr, nullptr, lse, nullptr, out, st); return csmae_check_launch("csmae_attn_fwd"); }
    if (hd <= 32) hipLaunchKernelGGL((attn_fwd_f32<32>), dim3(BH), dim3(256), sh, st, (const float*)qkv, (float*)out, lse, T, H, D, hd, scale);
    else if (hd <= 64) hipLaunchKernelGGL((attn_fwd_f32<64>), dim3(BH), dim3(256), sh, st, (const float*)qkv, (float*)out, lse, T, H, D, hd, scale);
    else hipLaunchKernelGGL((attn_fwd_f32<96>), dim3(BH), dim3(256), sh, st, (const float*)qkv, (float*)out, lse, T, H, D, hd, scale);
  } else { csmae_set_error("csmae_attn_fwd: unsupported dtype %d", dtype); return CSMAE_ERR_UNSUPPORTED; }
  return csmae_check_launch("csmae_attn_fwd");
}

extern "C" int csmae_attn_fwd(int dtype, long long B, int T, int H, int hd, const void* qkv, void* out, float* lse, void* stream) {
  return attn_fwd_impl(dtype, B, T, H, hd, qkv, out, lse, stream, nullptr);
}
// 1 when the (dtype, T, head_dim) shape runs the LDS-resident MFMA kernels, which can emit the fp8 copy of their output (csmae_attn_*_q)
extern "C" int csmae_attn_resident(int dtype, int T, int hd) { return dtype == CSMAE_BF16 && bf16_resident(T, hd) ? 1 : 0; }
static int attn_emit_check(const char* who, int dtype, int T, int hd, void* q, int fmt, const float* prev, float* next, float* dq) {
  CSMAE_REQUIRE(dtype == CSMAE_BF16 && bf16_resident(T, hd), "%s: bf16 shapes of the LDS-resident kernels only (csmae_attn_resident)", who);
  CSMAE_REQUIRE(q && prev && next && dq && (fmt == 0 || fmt == 1) && ((uintptr_t)q & 3) == 0, "%s: the fp8 copy needs q_out (4-byte aligned), amax_prev, amax_next, dq and fmt 0 / 1", who);
  return CSMAE_OK;
}
// softmax(QK^T / sqrt d) V as csmae_attn_fwd, plus `q_out` [B*T, H*hd] = the output as OCP fp8 bytes (q_fmt 0: e4m3) scaled by FMAX / max(amax_prev[64]);
// amax_next[64] receives partial maxima of |out|, dq[0] the de-quantisation factor (delayed scaling: csmae_gemm_fp8's conventions)
extern "C" int csmae_attn_fwd_q(int dtype, long long B, int T, int H, int hd, const void* qkv, void* out, float* lse, void* q_out, int q_fmt,
                                const float* q_amax_prev, float* q_amax_next, float* q_dq, void* stream) {
  if (int rc = attn_emit_check("csmae_attn_fwd_q", dtype, T, hd, q_out, q_fmt, q_amax_prev, q_amax_next, q_dq)) return rc;
  const Fp8Emit em{(unsigned char*)q_out, q_amax_prev, q_amax_next, q_dq, q_fmt};
  return attn_fwd_impl(dtype, B, T, H, hd, qkv, out, lse, stream, &em);
}

static int attn_bwd_impl(int dtype, long long B, int T, int H, int hd, const void* qkv, const void* out, const void* dout,
                         const float* lse, void* dqkv, void* stream, const Fp8Emit* em) {
  const int D = H * hd;
  int rc = check_common("csmae_attn_bwd", B, T, H, D, hd);
  if (rc) return rc;
  CSMAE_REQUIRE(dqkv || em, "csmae_attn_bwd: dqkv is null (allowed in csmae_attn_bwd_q only: the fp8 copy is then the only output)");
  hipStream_t st = (hipStream_t)stream;
  const float scale = 1.0f / sqrtf((float)hd);
  const int BH = (int)(B * H);
  if (dtype == CSMAE_BF16) {
    if (!bf16_resident(T, hd)) { CSMAE_REQUIRE(!em, "csmae_attn_bwd_q: the fp8 copy is emitted by the LDS-resident kernels only (csmae_attn_resident)"); launch_any<bf16_t>(true, B, T, H, D, hd, scale, qkv, out, dout, nullptr, lse, dqkv, st); return csmae_check_launch("csmae_attn_bwd"); }
#define CALLB(HDV, NK) launch_bwd_bf16<HDV, NK>(BH, qkv, out, dout, lse, dqkv, T, H, D, hd, scale, st, em)
    DISPATCH_BF16(CALLB)
#undef CALLB
  } else if (dtype == CSMAE_F32) {
    size_t sh = ((size_t)4 * T * (hd + 1) + 2 * T) * sizeof(float);
    if (sh > 160 * 1024 || hd > 96) { launch_any<float>(true, B, T, H, D, hd, scale, qkv, out, dout, nullptr, lse, dqkv, st); return csmae_check_launch("csmae_attn_bwd"); }
    if (hd <= 32) hipLaunchKernelGGL((attn_bwd_f32<32>), dim3(BH), dim3(256), sh, st, (const float*)qkv, (const float*)out, (const float*)dout, lse, (float*)dqkv, T, H, D, hd, scale);
    else if (hd <= 64) hipLaunchKernelGGL((attn_bwd_f32<64>), dim3(BH), dim3(256), sh, st, (const float*)qkv, (const float*)out, (const float*)dout, lse, (float*)dqkv, T, H, D, hd, scale);
    else hipLaunchKernelGGL((attn_bwd_f32<96>), dim3(BH), dim3(256), sh, st, (const float*)qkv, (const float*)out, (const float*)dout, lse, (float*)dqkv, T, H, D, hd, scale);
  } else { csmae_set_error("csmae_attn_bwd: unsupported dtype %d", dtype); return CSMAE_ERR_UNSUPPORTED; }
  return csmae_check_launch("csmae_attn_bwd");
}

extern "C" int csmae_attn_bwd(int dtype, long long B, int T, int H, int hd, const void* qkv, const void* out, const void* dout,
                              const float* lse, void* dqkv, void* stream) {
  return attn_bwd_impl(dtype, B, T, H, hd, qkv, out, dout, lse, dqkv, stream, nullptr);
}
// ... and its backward, plus `q_out` [B*T, 3*H*hd] = dqkv as fp8 bytes (q_fmt 1: e5m2), see csmae_attn_fwd_q
extern "C" int csmae_attn_bwd_q(int dtype, long long B, int T, int H, int hd, const void* qkv, const void* out, const void* dout, const float* lse,
                                void* dqkv, void* q_out, int q_fmt, const float* q_amax_prev, float* q_amax_next, float* q_dq, void* stream) {
  if (int rc = attn_emit_check("csmae_attn_bwd_q", dtype, T, hd, q_out, q_fmt, q_amax_prev, q_amax_next, q_dq)) return rc;
  const Fp8Emit em{(unsigned char*)q_out, q_amax_prev, q_amax_next, q_dq, q_fmt};
  return attn_bwd_impl(dtype, B, T, H, hd, qkv, out, dout, lse, dqkv, stream, &em);
}
