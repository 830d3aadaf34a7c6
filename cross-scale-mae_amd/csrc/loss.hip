// Loss heads of the Cross-Scale MAE step (fp32 reductions, HBM-bound):
//   recon_loss   MAE_ViT_Shared.py:97-163,269-290  patchify-on-the-fly target (+ optional norm_pix), per-patch mse/l2/mae/l1/bce
//   pair_loss    un-masked `.mean()` form used for the cross-decoder (MAE_ViT_MsLdCeCd.py:56-59) and latent (MsLdLe.py:44) terms
//   ntxent       util/contrast_loss.py:44-101 with cos_sim=True, tau=0.5 (mean-pool + normalise fused; masks are analytic)
//   finalize     deterministic single-workgroup reduction of all partial buffers into the scalar terms
#include "common.h"

#define LOSS_MSE 0
#define LOSS_L2 1
#define LOSS_MAE 2
#define LOSS_L1 3
#define LOSS_BCE 4
#define LOSS_NONE 5  // no per-patch term (pure ssim / ms_ssim): the gradient is the ssim family's `extra` alone

__device__ __forceinline__ float elem_loss(int kind, float pred, float t) {
  if (kind == LOSS_MSE || kind == LOSS_L2) { float d = pred - t; return d * d; }
  if (kind == LOSS_MAE || kind == LOSS_L1) return fabsf(pred - t);
  // bce with logits, torch's stable form: max(x,0) - x*t + log1p(exp(-|x|))
  return fmaxf(pred, 0.f) - pred * t + log1pf(expf(-fabsf(pred)));
}
__device__ __forceinline__ float elem_grad(int kind, float pred, float t) {  // d elem_loss / d pred
  if (kind == LOSS_MSE || kind == LOSS_L2) return 2.f * (pred - t);
  if (kind == LOSS_MAE || kind == LOSS_L1) { float d = pred - t; return d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); }
  if (kind == LOSS_NONE) return 0.f;
  return 1.f / (1.f + expf(-pred)) - t;
}
__device__ __forceinline__ bool mean_over_last(int kind) { return kind == LOSS_MSE || kind == LOSS_MAE || kind == LOSS_BCE; }

// one wave per patch: target values of patch (n2, l); element e = (ph*p + pw)*C + c   (MAE_ViT_Shared.py:36-38 "nhwpqc")
struct PatchGeom { int N, C, S, p, L, G, P; unsigned mC, mp; };   // mC / mp: ceil(2^24 / C), ceil(2^24 / p) — n / d = (n * m) >> 24 for n * d < 2^24 (0: divide)
__device__ __forceinline__ const float* patch_img(const PatchGeom& g, const float* img0, const float* img1, long long n2) {
  int view = (int)(n2 / g.N);
  return (view ? img1 : img0) + (n2 - (long long)view * g.N) * g.C * g.S * g.S;
}
__device__ __forceinline__ float patch_elem(const PatchGeom& g, const float* img, int l, int e) {
  int gh = l / g.G, gw = l - gh * g.G;
  // (three integer divisions per element were most of this loop: exact multiply-shift instead when the operands allow it)
  const int r = g.mC ? (int)(((unsigned long long)(unsigned)e * g.mC) >> 24) : e / g.C, c = e - r * g.C;
  const int ph = g.mp ? (int)(((unsigned long long)(unsigned)r * g.mp) >> 24) : r / g.p, pw = r - ph * g.p;
  return img[((long long)c * g.S + gh * g.p + ph) * g.S + gw * g.p + pw];
}
// per-patch normalisation statistics (norm_pix_loss: unbiased variance, eps 1e-6 — MAE_ViT_Shared.py:106-109)
__device__ __forceinline__ void patch_stats(const PatchGeom& g, const float* img, int l, int lane, float& mu, float& rs) {
  float s = 0.f;
  for (int e = lane; e < g.P; e += 64) s += patch_elem(g, img, l, e);
  mu = wave_sum(s) / g.P;
  float q = 0.f;
  for (int e = lane; e < g.P; e += 64) { float d = patch_elem(g, img, l, e) - mu; q += d * d; }
  rs = rsqrtf(wave_sum(q) / (g.P - 1) + 1.0e-6f);
}

// min/max of the processed target per patch (bce's scale_01 works on the whole tensor — MAE_ViT_Shared.py:94-95)
__global__ __launch_bounds__(256) void target_minmax_kernel(PatchGeom g, int norm_pix, long long patches, const float* __restrict__ img0,
                                                            const float* __restrict__ img1, float* __restrict__ mm /*[patches][2]*/) {
  const int lane = threadIdx.x & 63;
  const long long pt = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pt >= patches) return;
  const long long n2 = pt / g.L; const int l = (int)(pt - n2 * g.L);
  const float* img = patch_img(g, img0, img1, n2);
  float mu = 0.f, rs = 1.f;
  if (norm_pix) patch_stats(g, img, l, lane, mu, rs);
  float lo = INFINITY, hi = -INFINITY;
  for (int e = lane; e < g.P; e += 64) { float t = (patch_elem(g, img, l, e) - mu) * rs; lo = fminf(lo, t); hi = fmaxf(hi, t); }
  lo = -wave_max(-lo); hi = wave_max(hi);
  if (lane == 0) { mm[pt * 2] = lo; mm[pt * 2 + 1] = hi; }
}
__global__ __launch_bounds__(1024) void minmax_reduce_kernel(long long per_view, int views, const float* __restrict__ mm, float* __restrict__ out /*[views][2]*/) {
  __shared__ float slo[16], shi[16];
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int v = 0; v < views; ++v) {
    float lo = INFINITY, hi = -INFINITY;
    for (long long i = threadIdx.x; i < per_view; i += blockDim.x) { lo = fminf(lo, mm[(v * per_view + i) * 2]); hi = fmaxf(hi, mm[(v * per_view + i) * 2 + 1]); }
    lo = -wave_max(-lo); hi = wave_max(hi);
    if ((threadIdx.x & 63) == 0) { slo[w] = lo; shi[w] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) { for (int i = 1; i < nw; ++i) { lo = fminf(lo, slo[i]); hi = fmaxf(hi, shi[i]); } out[v * 2] = lo; out[v * 2 + 1] = hi; }
    __syncthreads();
  }
}

// rowloss[n2*L + l] = mean/sum_e f(pred[n2, 1+l, e], target).  `mask` (nullable): patches with mask 0 are skipped (rowloss 0) — the
// loss only weighs masked patches (MAE_ViT_Shared.py:113-120), three quarters of them at mask_ratio 0.75.
template <typename TP>
__global__ __launch_bounds__(256) void recon_fwd_kernel(PatchGeom g, int kind, int norm_pix, long long patches, const float* __restrict__ img0,
                                                        const float* __restrict__ img1, const TP* __restrict__ pred, long long ldp,
                                                        const float* __restrict__ minmax, const float* __restrict__ mask, float* __restrict__ rowloss) {
  const int lane = threadIdx.x & 63;
  const long long pt = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pt >= patches) return;
  if (mask && mask[pt] == 0.f) { if (lane == 0) rowloss[pt] = 0.f; return; }
  const long long n2 = pt / g.L; const int l = (int)(pt - n2 * g.L);
  const float* img = patch_img(g, img0, img1, n2);
  float mu = 0.f, rs = 1.f;
  if (norm_pix) patch_stats(g, img, l, lane, mu, rs);
  float lo = 0.f, sc = 1.f;
  if (kind == LOSS_BCE) { int v = (int)(n2 / g.N); lo = minmax[v * 2]; sc = 1.f / (minmax[v * 2 + 1] - lo + 1.0e-6f); }
  const TP* pr = pred + (n2 * (g.L + 1) + 1 + l) * ldp;
  float s = 0.f;
  for (int e = lane; e < g.P; e += 64) {
    float t = (patch_elem(g, img, l, e) - mu) * rs;
    if (kind == LOSS_BCE) t = (t - lo) * sc;
    s += elem_loss(kind, ld_as_f32<TP>(pr + e), t);
  }
  s = wave_sum(s);
  if (lane == 0) rowloss[pt] = mean_over_last(kind) ? s / g.P : s;
}
// dpred[n2, 1+l, e] = gout * vscale * mask / masksum(view) * f'(pred, t) / (P or 1) (+ extra[n2, l, e], the ssim family's share,
// already scaled, which also reaches visible patches through scale_01's min / max) ; cls rows and pad columns are zeroed
template <typename T, typename TP>
__global__ __launch_bounds__(256) void recon_bwd_kernel(PatchGeom g, int kind, int norm_pix, long long rows, const float* __restrict__ img0,
                                                        const float* __restrict__ img1, const TP* __restrict__ pred, long long ldp,
                                                        const float* __restrict__ minmax, const float* __restrict__ mask,
                                                        const float* __restrict__ losses, const float* __restrict__ gout, float vscale,
                                                        const float* __restrict__ extra, T* __restrict__ dpred, long long ldd) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);  // row of [B2*(L+1)]
  if (row >= rows) return;
  const long long n2 = row / (g.L + 1); const int j = (int)(row - n2 * (g.L + 1));
  T* dp = dpred + row * ldd;
  float m = j > 0 ? mask[n2 * g.L + j - 1] : 0.f;
  const float* ex = (extra && j > 0) ? extra + (n2 * g.L + j - 1) * g.P : nullptr;
  if (m == 0.f || kind == LOSS_NONE) { for (int e = lane; e < ldd; e += 64) st_from_f32<T>(dp + e, (ex && e < g.P) ? ex[e] : 0.f); return; }
  const int l = j - 1, v = (int)(n2 / g.N);
  const float* img = patch_img(g, img0, img1, n2);
  float mu = 0.f, rs = 1.f;
  if (norm_pix) patch_stats(g, img, l, lane, mu, rs);
  float lo = 0.f, sc = 1.f;
  if (kind == LOSS_BCE) { lo = minmax[v * 2]; sc = 1.f / (minmax[v * 2 + 1] - lo + 1.0e-6f); }
  const float coef = gout[0] * vscale * m / losses[6 + v] / (mean_over_last(kind) ? (float)g.P : 1.f);
  const TP* pr = pred + row * ldp;
  for (int e = lane; e < ldd; e += 64) {
    float o = 0.f;
    if (e < g.P) {
      float t = (patch_elem(g, img, l, e) - mu) * rs;
      if (kind == LOSS_BCE) t = (t - lo) * sc;
      o = coef * elem_grad(kind, ld_as_f32<TP>(pr + e), t);
      if (ex) o += ex[e];
    }
    st_from_f32<T>(dp + e, o);
  }
}

// ---- throughput forms of the two kernels above for the geometry the step spends its time in: bf16 predictions and bf16 dpred, C * p * p a
// multiple of 128 (ViT-*/16 RGB: P = 768; 4-band: 1024), the mse / l2 / mae / l1 kinds.  A wave owns a patch; a lane owns the element PAIRS
// e = 2 lane + 128 it: the prediction row is read as one 4-byte load per pair (whole 256-byte lines per wave instruction), the 2 NP image
// values of a lane are independent gathers that are all in flight before the first use (the generic form walks the patch with a runtime
// trip count: one dependent load pair per iteration), and the image is read ONCE also under norm_pix_loss (the values stay in registers
// for the statistics).  Same per-element arithmetic as the generic kernels (the partial sums are taken in another order).
template <int C, int P_, int NP>
__device__ __forceinline__ void patch_load_pairs(const PatchGeom& g, const float* __restrict__ img, int l, int lane, float (&t)[2 * NP]) {
  const int gh = l / g.G, gw = l - gh * g.G;
  const float* base = img + ((long long)gh * P_) * g.S + gw * P_;
#pragma unroll
  for (int it = 0; it < NP; ++it)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int e = 2 * lane + 128 * it + k, r = e / C, c = e - r * C, ph = r / P_, pw = r - ph * P_;
      t[2 * it + k] = base[((long long)c * g.S + ph) * g.S + pw];
    }
}
template <int NV>
__device__ __forceinline__ void patch_normalise(int P, float (&t)[NV], float& mu, float& rs) {   // norm_pix_loss: unbiased variance, eps 1e-6
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k) s += t[k];
  mu = wave_sum(s) / P;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k) { const float d = t[k] - mu; q += d * d; }
  rs = rsqrtf(wave_sum(q) / (P - 1) + 1.0e-6f);
#pragma unroll
  for (int k = 0; k < NV; ++k) t[k] = (t[k] - mu) * rs;
}
template <int C, int P_, int NP>
__global__ __launch_bounds__(256) void recon_fwd_fast_kernel(PatchGeom g, int kind, int norm_pix, long long patches, const float* __restrict__ img0,
                                                             const float* __restrict__ img1, const bf16_t* __restrict__ pred, long long ldp,
                                                             const float* __restrict__ mask, float* __restrict__ rowloss) {
  const int lane = threadIdx.x & 63;
  const long long pt = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pt >= patches) return;
  if (mask && mask[pt] == 0.f) { if (lane == 0) rowloss[pt] = 0.f; return; }
  const long long n2 = pt / g.L; const int l = (int)(pt - n2 * g.L);
  const unsigned* pr = reinterpret_cast<const unsigned*>(pred + (n2 * (g.L + 1) + 1 + l) * ldp) + lane;
  unsigned pv[NP];
#pragma unroll
  for (int it = 0; it < NP; ++it) pv[it] = pr[64 * it];
  float t[2 * NP];
  patch_load_pairs<C, P_, NP>(g, patch_img(g, img0, img1, n2), l, lane, t);
  float mu, rs;
  if (norm_pix) patch_normalise<2 * NP>(g.P, t, mu, rs);
  float s = 0.f;
#pragma unroll
  for (int it = 0; it < NP; ++it) {
    s += elem_loss(kind, __uint_as_float(pv[it] << 16), t[2 * it]);
    s += elem_loss(kind, __uint_as_float(pv[it] & 0xffff0000u), t[2 * it + 1]);
  }
  s = wave_sum(s);
  if (lane == 0) rowloss[pt] = mean_over_last(kind) ? s / g.P : s;
}
template <int C, int P_, int NP>
__global__ __launch_bounds__(256) void recon_bwd_fast_kernel(PatchGeom g, int kind, int norm_pix, long long rows, const float* __restrict__ img0,
                                                             const float* __restrict__ img1, const bf16_t* __restrict__ pred, long long ldp,
                                                             const float* __restrict__ mask, const float* __restrict__ losses,
                                                             const float* __restrict__ gout, float vscale, bf16_t* __restrict__ dpred, long long ldd) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);  // row of [B2*(L+1)]
  if (row >= rows) return;
  const long long n2 = row / (g.L + 1); const int j = (int)(row - n2 * (g.L + 1));
  unsigned* dp = reinterpret_cast<unsigned*>(dpred + row * ldd) + lane;   // (ldd == P here: no pad columns)
  const float m = j > 0 ? mask[n2 * g.L + j - 1] : 0.f;
  if (m == 0.f) {
#pragma unroll
    for (int it = 0; it < NP; ++it) dp[64 * it] = 0u;
    return;
  }
  const int l = j - 1, v = (int)(n2 / g.N);
  const unsigned* pr = reinterpret_cast<const unsigned*>(pred + row * ldp) + lane;
  unsigned pv[NP];
#pragma unroll
  for (int it = 0; it < NP; ++it) pv[it] = pr[64 * it];
  float t[2 * NP];
  patch_load_pairs<C, P_, NP>(g, patch_img(g, img0, img1, n2), l, lane, t);
  float mu, rs;
  if (norm_pix) patch_normalise<2 * NP>(g.P, t, mu, rs);
  const float coef = gout[0] * vscale * m / losses[6 + v] / (mean_over_last(kind) ? (float)g.P : 1.f);
#pragma unroll
  for (int it = 0; it < NP; ++it)
    dp[64 * it] = pack2bf(coef * elem_grad(kind, __uint_as_float(pv[it] << 16), t[2 * it]), coef * elem_grad(kind, __uint_as_float(pv[it] & 0xffff0000u), t[2 * it + 1]));
}
static bool recon_fast_geometry(int kind, int C, int p, long long ldp, long long ldd) {
  return kind >= LOSS_MSE && kind <= LOSS_L1 && p == 16 && (C == 3 || C == 4) && ldp % 2 == 0 && ldd == (long long)C * p * p;
}

static PatchGeom make_geom(int N, int C, int S, int p) {
  PatchGeom g; g.N = N; g.C = C; g.S = S; g.p = p; g.G = S / p; g.L = g.G * g.G; g.P = p * p * C;
  const bool fast = (long long)g.P * (C > p ? C : p) < (1ll << 24);
  g.mC = fast ? (unsigned)(((1ull << 24) + C - 1) / C) : 0u; g.mp = fast ? (unsigned)(((1ull << 24) + p - 1) / p) : 0u;
  return g;
}

extern "C" int csmae_target_minmax(int norm_pix, long long B2, int N, int C, int S, int p, const float* img0, const float* img1,
                                   float* scratch /*[B2*L*2]*/, float* out /*[views*2]*/, void* stream) {
  PatchGeom g = make_geom(N, C, S, p);
  long long patches = B2 * g.L;
  hipLaunchKernelGGL(target_minmax_kernel, dim3(cdiv(patches, 4)), dim3(256), 0, (hipStream_t)stream, g, norm_pix, patches, img0, img1, scratch);
  hipLaunchKernelGGL(minmax_reduce_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (long long)N * g.L, (int)(B2 / N), scratch, out);
  return csmae_check_launch("csmae_target_minmax");
}
extern "C" int csmae_recon_loss_fwd(int kind, int norm_pix, int pred_dtype, long long B2, int N, int C, int S, int p, const float* img0, const float* img1,
                                    const void* pred, long long ldp, const float* minmax, const float* mask, float* rowloss, void* stream) {
  CSMAE_REQUIRE(kind >= LOSS_MSE && kind <= LOSS_BCE, "csmae_recon_loss_fwd: loss kind %d is outside the hot-path scope (ssim family: SURVEY §2 row 2)", kind);
  CSMAE_REQUIRE(B2 > 0 && N > 0 && B2 % N == 0 && S % p == 0 && (kind != LOSS_BCE || minmax), "csmae_recon_loss_fwd: bad args");
  CSMAE_REQUIRE(pred_dtype == CSMAE_F32 || pred_dtype == CSMAE_BF16, "csmae_recon_loss_fwd: bad prediction dtype %d", pred_dtype);
  PatchGeom g = make_geom(N, C, S, p);
  long long patches = B2 * g.L;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(cdiv(patches, 4)), block(256);
  if (pred_dtype == CSMAE_BF16 && recon_fast_geometry(kind, C, p, ldp, g.P)) {
    if (C == 3) hipLaunchKernelGGL((recon_fwd_fast_kernel<3, 16, 6>), grid, block, 0, st, g, kind, norm_pix, patches, img0, img1, (const bf16_t*)pred, ldp, mask, rowloss);
    else hipLaunchKernelGGL((recon_fwd_fast_kernel<4, 16, 8>), grid, block, 0, st, g, kind, norm_pix, patches, img0, img1, (const bf16_t*)pred, ldp, mask, rowloss);
  } else if (pred_dtype == CSMAE_BF16) hipLaunchKernelGGL((recon_fwd_kernel<bf16_t>), grid, block, 0, st, g, kind, norm_pix, patches, img0, img1, (const bf16_t*)pred, ldp, minmax, mask, rowloss);
  else hipLaunchKernelGGL((recon_fwd_kernel<float>), grid, block, 0, st, g, kind, norm_pix, patches, img0, img1, (const float*)pred, ldp, minmax, mask, rowloss);
  return csmae_check_launch("csmae_recon_loss_fwd");
}
extern "C" int csmae_recon_loss_bwd(int kind, int norm_pix, int out_dtype, int pred_dtype, long long B2, int N, int C, int S, int p, const float* img0,
                                    const float* img1, const void* pred, long long ldp, const float* minmax, const float* mask,
                                    const float* losses, const float* gout, float vscale, const float* extra, void* dpred, long long ldd,
                                    void* stream) {
  CSMAE_REQUIRE(kind >= LOSS_MSE && kind <= LOSS_NONE && (kind != LOSS_NONE || extra), "csmae_recon_loss_bwd: bad loss kind %d", kind);
  CSMAE_REQUIRE((out_dtype == CSMAE_F32 || out_dtype == CSMAE_BF16) && (pred_dtype == CSMAE_F32 || pred_dtype == CSMAE_BF16), "csmae_recon_loss_bwd: bad dtype %d / %d", out_dtype, pred_dtype);
  PatchGeom g = make_geom(N, C, S, p);
  long long rows = B2 * (g.L + 1);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(cdiv(rows, 4)), block(256);
#define RECON_BWD(T, TP) hipLaunchKernelGGL((recon_bwd_kernel<T, TP>), grid, block, 0, st, g, kind, norm_pix, rows, img0, img1, (const TP*)pred, ldp, minmax, mask, losses, gout, vscale, extra, (T*)dpred, ldd)
  if (out_dtype == CSMAE_BF16 && pred_dtype == CSMAE_BF16 && !extra && recon_fast_geometry(kind, C, p, ldp, ldd)) {
    if (C == 3) hipLaunchKernelGGL((recon_bwd_fast_kernel<3, 16, 6>), grid, block, 0, st, g, kind, norm_pix, rows, img0, img1, (const bf16_t*)pred, ldp, mask, losses, gout, vscale, (bf16_t*)dpred, ldd);
    else hipLaunchKernelGGL((recon_bwd_fast_kernel<4, 16, 8>), grid, block, 0, st, g, kind, norm_pix, rows, img0, img1, (const bf16_t*)pred, ldp, mask, losses, gout, vscale, (bf16_t*)dpred, ldd);
  } else if (out_dtype == CSMAE_BF16 && pred_dtype == CSMAE_BF16) RECON_BWD(bf16_t, bf16_t);
  else if (out_dtype == CSMAE_BF16) RECON_BWD(bf16_t, float);
  else if (pred_dtype == CSMAE_BF16) RECON_BWD(float, bf16_t);
  else RECON_BWD(float, float);
#undef RECON_BWD
  return csmae_check_launch("csmae_recon_loss_bwd");
}

// ------------------------------------------------------------------------------------------ un-masked pair loss
// view row r -> storage row (r / group) * gstride + off + r % group  (same convention as rows_gather)
struct RowView { long long group, gstride, off; };
__device__ __forceinline__ long long vrow(const RowView& v, long long r) { return (r / v.group) * v.gstride + v.off + r % v.group; }
#define PAIR_BLOCKS 512
__global__ __launch_bounds__(256) void pair_fwd_kernel(int kind, long long rows, int D, const float* __restrict__ a, RowView va,
                                                       const float* __restrict__ t, RowView vt, float* __restrict__ partial) {
  __shared__ float red[32];
  const int dv = D >> 2;
  float s = 0.f;
  // items = (row, 16-byte column) pairs, grid-strided: every thread has work whatever the row length (a 512-wide row is 128 items: half of a
  // workgroup idled when a workgroup walked one row at a time), two items per trip so that four loads are in flight
  const long long items = rows * dv, step = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < items; i += 2 * step) {
    const long long r0 = i / dv, i1 = i + step;
    const int c0 = (int)(i - r0 * dv);
    f4_t x0 = *reinterpret_cast<const f4_t*>(a + vrow(va, r0) * D + c0 * 4), y0 = *reinterpret_cast<const f4_t*>(t + vrow(vt, r0) * D + c0 * 4);
    if (i1 < items) {
      const long long r1 = i1 / dv; const int c1 = (int)(i1 - r1 * dv);
      f4_t x1 = *reinterpret_cast<const f4_t*>(a + vrow(va, r1) * D + c1 * 4), y1 = *reinterpret_cast<const f4_t*>(t + vrow(vt, r1) * D + c1 * 4);
      for (int k = 0; k < 4; ++k) s += elem_loss(kind, x1[k], y1[k]);
    }
    for (int k = 0; k < 4; ++k) s += elem_loss(kind, x0[k], y0[k]);
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
// d/da = gout * coef * f'(a - t); d/dt = -that.  da_lp: low-precision copy (GEMM operand); *_acc: fp32 += into residual-grad buffers
template <typename T>
__global__ __launch_bounds__(256) void pair_bwd_kernel(int kind, long long rows, int D, const float* __restrict__ a, RowView va,
                                                       const float* __restrict__ t, RowView vt, const float* __restrict__ gout, float coef,
                                                       T* __restrict__ da_lp, float* __restrict__ da_acc, float* __restrict__ dt_acc) {
  const int dv = D >> 2;
  const float cf = gout[0] * coef;
  for (long long r = blockIdx.x; r < rows; r += gridDim.x) {
    const long long ra = vrow(va, r), rt = vrow(vt, r);
    for (int c = threadIdx.x; c < dv; c += blockDim.x) {
      f4_t x = *reinterpret_cast<const f4_t*>(a + ra * D + c * 4), y = *reinterpret_cast<const f4_t*>(t + rt * D + c * 4), g;
      for (int k = 0; k < 4; ++k) g[k] = cf * elem_grad(kind, x[k], y[k]);
      if (da_lp) st4<T>(da_lp + r * D + c * 4, g);
      if (da_acc) { f4_t o = *reinterpret_cast<f4_t*>(da_acc + ra * D + c * 4) + g; *reinterpret_cast<f4_t*>(da_acc + ra * D + c * 4) = o; }
      if (dt_acc) { f4_t o = *reinterpret_cast<f4_t*>(dt_acc + rt * D + c * 4) - g; *reinterpret_cast<f4_t*>(dt_acc + rt * D + c * 4) = o; }
    }
  }
}
extern "C" int csmae_pair_loss_fwd(int kind, long long rows, int D, const float* a, long long a_group, long long a_gstride, long long a_off,
                                   const float* t, long long t_group, long long t_gstride, long long t_off, float* partial /*[512]*/, void* stream) {
  CSMAE_REQUIRE(kind >= LOSS_MSE && kind <= LOSS_L1, "csmae_pair_loss: kind %d unsupported for un-masked pair losses (mse/l2/mae/l1 only)", kind);
  CSMAE_REQUIRE(rows > 0 && D % 4 == 0, "csmae_pair_loss_fwd: bad geometry");
  RowView va{a_group, a_gstride, a_off}, vt{t_group, t_gstride, t_off};
  hipLaunchKernelGGL(pair_fwd_kernel, dim3(PAIR_BLOCKS), dim3(256), 0, (hipStream_t)stream, kind, rows, D, a, va, t, vt, partial);
  return csmae_check_launch("csmae_pair_loss_fwd");
}
extern "C" int csmae_pair_loss_bwd(int kind, int lp_dtype, long long rows, int D, const float* a, long long a_group, long long a_gstride, long long a_off,
                                   const float* t, long long t_group, long long t_gstride, long long t_off, const float* gout, float coef,
                                   void* da_lp, float* da_acc, float* dt_acc, void* stream) {
  CSMAE_REQUIRE(kind >= LOSS_MSE && kind <= LOSS_L1, "csmae_pair_loss: kind %d unsupported for un-masked pair losses (mse/l2/mae/l1 only)", kind);
  RowView va{a_group, a_gstride, a_off}, vt{t_group, t_gstride, t_off};
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)(rows < 8192 ? rows : 8192));   // (2 048 workgroups walked 12 rows each one after the other: 1.5-2.5 TB/s; every CU's share resident at once)
  if (lp_dtype == CSMAE_BF16) hipLaunchKernelGGL((pair_bwd_kernel<bf16_t>), grid, dim3(128), 0, st, kind, rows, D, a, va, t, vt, gout, coef, (bf16_t*)da_lp, da_acc, dt_acc);
  else hipLaunchKernelGGL((pair_bwd_kernel<float>), grid, dim3(128), 0, st, kind, rows, D, a, va, t, vt, gout, coef, (float*)da_lp, da_acc, dt_acc);
  return csmae_check_launch("csmae_pair_loss_bwd");
}

// ------------------------------------------------------------------------------------------ NT-Xent (cosine, per-GPU negatives)
// z_i = normalize(mean_t latent[i, 1+t, :]);  e_ij = exp(z_i.z_j / tau);  pos_i = e_{i, i+-N};  neg_i = sum_{j != i, j != partner} e_ij
// rowloss_i = -log(pos_i / (neg_i + eps))        (positives are NOT in the denominator — contrast_loss.py:28,94-99)
__global__ __launch_bounds__(256) void ntx_pool_kernel(int Te, int keep, int D, const float* __restrict__ latent, float* __restrict__ z, float* __restrict__ inv_norm) {
  __shared__ float red[32];
  const long long i = blockIdx.x;
  float q = 0.f;
  for (int d = threadIdx.x; d < D; d += blockDim.x) {
    float s = 0.f;
    for (int t = 0; t < keep; ++t) s += latent[(i * Te + 1 + t) * D + d];
    s /= keep;
    z[i * D + d] = s; q += s * s;
  }
  q = block_sum(q, red);
  const float inv = 1.f / fmaxf(sqrtf(q), 1e-12f);
  for (int d = threadIdx.x; d < D; d += blockDim.x) z[i * D + d] *= inv;
  if (threadIdx.x == 0) inv_norm[i] = inv;
}
// The same for D % 4 == 0, D <= 4096 (every geometry of the step): the sample's keep x D values are walked by ALL 1024 threads — G = 1024 / (D / 4)
// groups of rows, a thread owns one 16-byte column of every G-th row, its loads independent of each other — and folded through LDS in group
// order (deterministic).  The first form gave a thread a whole column: `keep` dependent 4-byte loads, 256 threads per sample (0.65 TB/s).
__global__ __launch_bounds__(1024) void ntx_pool4_kernel(int Te, int keep, int D, const float* __restrict__ latent, float* __restrict__ z, float* __restrict__ inv_norm) {
  __shared__ float red[32];
  __shared__ f4_t part[1024];
  const long long i = blockIdx.x;
  const int dv = D >> 2, G = 1024 / dv, g = threadIdx.x / dv, c = threadIdx.x - g * dv;
  f4_t s = {0.f, 0.f, 0.f, 0.f};
  if (g < G) {
    const float* base = latent + (i * Te + 1) * D + c * 4;
    int t = g;
    for (; t + 3 * G < keep; t += 4 * G) {
      const f4_t a0 = *reinterpret_cast<const f4_t*>(base + (long long)t * D), a1 = *reinterpret_cast<const f4_t*>(base + (long long)(t + G) * D);
      const f4_t a2 = *reinterpret_cast<const f4_t*>(base + (long long)(t + 2 * G) * D), a3 = *reinterpret_cast<const f4_t*>(base + (long long)(t + 3 * G) * D);
      s += (a0 + a1) + (a2 + a3);
    }
    for (; t < keep; t += G) s += *reinterpret_cast<const f4_t*>(base + (long long)t * D);
    part[threadIdx.x] = s;
  }
  __syncthreads();
  float q = 0.f;
  if (g == 0) {
    for (int k = 1; k < G; ++k) s += part[k * dv + c];
    s = s / (float)keep;
    q = s[0] * s[0] + s[1] * s[1] + s[2] * s[2] + s[3] * s[3];
  }
  q = block_sum(q, red);
  const float inv = 1.f / fmaxf(sqrtf(q), 1e-12f);
  if (g == 0) *reinterpret_cast<f4_t*>(z + i * D + c * 4) = s * inv;
  if (threadIdx.x == 0) inv_norm[i] = inv;
}
__global__ __launch_bounds__(1024) void ntx_sim_kernel(int N, int D, const float* __restrict__ z, float tau, float eps, float* __restrict__ E,
                                                      float* __restrict__ neg, float* __restrict__ rowloss) {
  __shared__ float red[32];
  extern __shared__ float zi[];
  const int i = blockIdx.x, B2 = 2 * N, partner = (i + N) % B2;
  for (int d = threadIdx.x; d < D; d += blockDim.x) zi[d] = z[(long long)i * D + d];
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const bool vec = (D & 3) == 0;
  float nsum = 0.f;
  for (int j0 = 4 * w; j0 < B2; j0 += 4 * (blockDim.x >> 6)) {  // four rows per wave iteration: independent load chains, one pass over z_i
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    if (vec) {   // 16-byte loads: D / 256 trips with four rows' loads in flight (scalar loads made this a 12-trip dependent walk at D = 768)
      for (int d = lane * 4; d < D; d += 256) {
        const f4_t a = *reinterpret_cast<const f4_t*>(zi + d);
#pragma unroll
        for (int u = 0; u < 4; ++u) if (j0 + u < B2) {
          const f4_t b = *reinterpret_cast<const f4_t*>(z + (long long)(j0 + u) * D + d);
          s[u] += (a[0] * b[0] + a[1] * b[1]) + (a[2] * b[2] + a[3] * b[3]);
        }
      }
    } else {
      for (int d = lane; d < D; d += 64) {
        const float a = zi[d];
#pragma unroll
        for (int u = 0; u < 4; ++u) if (j0 + u < B2) s[u] += a * z[(long long)(j0 + u) * D + d];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = j0 + u;
      const float e = expf(wave_sum(s[u]) / tau);
      if (lane == 0 && j < B2) { E[(long long)i * B2 + j] = e; if (j != i && j != partner) nsum += e; }
    }
  }
  nsum = block_sum(nsum, red);
  if (threadIdx.x == 0) neg[i] = nsum;
  __syncthreads();
  if (threadIdx.x == 0) rowloss[i] = -logf(E[(long long)i * B2 + partner] / (nsum + eps));
}
// dL/dc_ij = w * ( [j neg] e_ij / (tau (neg_i+eps))  -  [j == partner] / tau ),  w = gout / 2N ;  dz_i = sum_j (G_ij + G_ji) z_j
// then through F.normalize: df = (dz - z (z.dz)) * inv_norm ; dpool = df (the 1/keep of the mean is applied by latent_grad_finish)
__global__ __launch_bounds__(1024) void ntx_bwd_kernel(int N, int D, const float* __restrict__ z, const float* __restrict__ inv_norm,
                                                      const float* __restrict__ E, const float* __restrict__ neg, float tau, float eps,
                                                      const float* __restrict__ gout, float* __restrict__ dpool) {
  __shared__ float red[32];
  extern __shared__ float coef[];  // [2N]
  const int i = blockIdx.x, B2 = 2 * N, partner = (i + N) % B2;
  const float w = gout[0] / B2;
  for (int j = threadIdx.x; j < B2; j += blockDim.x) {
    float c = 0.f;
    if (j == partner) c = -2.f * w / tau;  // G_ip + G_pi
    else if (j != i) c = w / tau * (E[(long long)i * B2 + j] / (neg[i] + eps) + E[(long long)j * B2 + i] / (neg[j] + eps));
    coef[j] = c;
  }
  __syncthreads();
  float dot = 0.f;
  for (int d = threadIdx.x; d < D; d += blockDim.x) {
    float s = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;   // four independent chains: the 2N-long walk is latency-, not bandwidth-bound
    int j = 0;
    for (; j + 3 < B2; j += 4) {
      s += coef[j] * z[(long long)j * D + d]; s1 += coef[j + 1] * z[(long long)(j + 1) * D + d];
      s2 += coef[j + 2] * z[(long long)(j + 2) * D + d]; s3 += coef[j + 3] * z[(long long)(j + 3) * D + d];
    }
    for (; j < B2; ++j) s += coef[j] * z[(long long)j * D + d];
    s = (s + s1) + (s2 + s3);
    dpool[(long long)i * D + d] = s;
    dot += s * z[(long long)i * D + d];
  }
  dot = block_sum(dot, red);
  const float inv = inv_norm[i];
  for (int d = threadIdx.x; d < D; d += blockDim.x) dpool[(long long)i * D + d] = (dpool[(long long)i * D + d] - z[(long long)i * D + d] * dot) * inv;
}
// The same for D % 4 == 0, D <= 4096: 16-byte columns, the 2N-long walk split over G = 1024 / (D / 4) groups of threads with four loads in flight
// each, folded through LDS in group order (the first form: one 4-byte column per thread, 2N dependent-latency trips; it sits on the main
// chain between the decoder's and the encoder's backward).
__global__ __launch_bounds__(1024) void ntx_bwd4_kernel(int N, int D, const float* __restrict__ z, const float* __restrict__ inv_norm,
                                                       const float* __restrict__ E, const float* __restrict__ neg, float tau, float eps,
                                                       const float* __restrict__ gout, float* __restrict__ dpool) {
  __shared__ float red[32];
  __shared__ f4_t part[1024];
  extern __shared__ float coef[];  // [2N]
  const int i = blockIdx.x, B2 = 2 * N, partner = (i + N) % B2;
  const float w = gout[0] / B2;
  for (int j = threadIdx.x; j < B2; j += blockDim.x) {
    float c = 0.f;
    if (j == partner) c = -2.f * w / tau;  // G_ip + G_pi
    else if (j != i) c = w / tau * (E[(long long)i * B2 + j] / (neg[i] + eps) + E[(long long)j * B2 + i] / (neg[j] + eps));
    coef[j] = c;
  }
  __syncthreads();
  const int dv = D >> 2, G = 1024 / dv, g = threadIdx.x / dv, c = threadIdx.x - g * dv;
  f4_t s = {0.f, 0.f, 0.f, 0.f};
  if (g < G) {
    const float* base = z + c * 4;
    int j = g;
    for (; j + 3 * G < B2; j += 4 * G) {
      const f4_t a0 = *reinterpret_cast<const f4_t*>(base + (long long)j * D), a1 = *reinterpret_cast<const f4_t*>(base + (long long)(j + G) * D);
      const f4_t a2 = *reinterpret_cast<const f4_t*>(base + (long long)(j + 2 * G) * D), a3 = *reinterpret_cast<const f4_t*>(base + (long long)(j + 3 * G) * D);
      s += (a0 * coef[j] + a1 * coef[j + G]) + (a2 * coef[j + 2 * G] + a3 * coef[j + 3 * G]);
    }
    for (; j < B2; j += G) s += *reinterpret_cast<const f4_t*>(base + (long long)j * D) * coef[j];
    part[threadIdx.x] = s;
  }
  __syncthreads();
  float dot = 0.f;
  f4_t zi = {0.f, 0.f, 0.f, 0.f};
  if (g == 0) {
    for (int k = 1; k < G; ++k) s += part[k * dv + c];
    zi = *reinterpret_cast<const f4_t*>(z + (long long)i * D + c * 4);
    dot = (s[0] * zi[0] + s[1] * zi[1]) + (s[2] * zi[2] + s[3] * zi[3]);
  }
  dot = block_sum(dot, red);
  if (g == 0) *reinterpret_cast<f4_t*>(dpool + (long long)i * D + c * 4) = (s - zi * dot) * inv_norm[i];
}
extern "C" int csmae_ntxent_fwd(int N, int Te, int keep, int D, const float* latent, float tau, float eps, float* z, float* inv_norm,
                                float* E, float* neg, float* rowloss, void* stream) {
  CSMAE_REQUIRE(N > 0 && keep > 0 && keep < Te && D > 0 && D * 4 <= 64 * 1024, "csmae_ntxent_fwd: bad geometry N=%d Te=%d keep=%d D=%d", N, Te, keep, D);
  hipStream_t st = (hipStream_t)stream;
  if (D % 4 == 0 && D <= 4096) hipLaunchKernelGGL(ntx_pool4_kernel, dim3(2 * N), dim3(1024), 0, st, Te, keep, D, latent, z, inv_norm);
  else hipLaunchKernelGGL(ntx_pool_kernel, dim3(2 * N), dim3(256), 0, st, Te, keep, D, latent, z, inv_norm);
  hipLaunchKernelGGL(ntx_sim_kernel, dim3(2 * N), dim3(1024), D * sizeof(float), st, N, D, z, tau, eps, E, neg, rowloss);
  return csmae_check_launch("csmae_ntxent_fwd");
}
extern "C" int csmae_ntxent_bwd(int N, int D, const float* z, const float* inv_norm, const float* E, const float* neg, float tau, float eps,
                                const float* gout, float* dpool, void* stream) {
  CSMAE_REQUIRE(N > 0 && D > 0 && 2 * N * 4 <= 64 * 1024, "csmae_ntxent_bwd: bad geometry");
  if (D % 4 == 0 && D <= 4096) hipLaunchKernelGGL(ntx_bwd4_kernel, dim3(2 * N), dim3(1024), 2 * N * sizeof(float), (hipStream_t)stream, N, D, z, inv_norm, E, neg, tau, eps, gout, dpool);
  else hipLaunchKernelGGL(ntx_bwd_kernel, dim3(2 * N), dim3(D >= 1024 ? 1024 : ((D + 63) / 64) * 64), 2 * N * sizeof(float), (hipStream_t)stream, N, D, z, inv_norm, E, neg, tau, eps, gout, dpool);
  return csmae_check_launch("csmae_ntxent_bwd");
}
// dlat[n, t>=1, :] += dpool[n, :] * inv_keep ; then emit the low-precision copy that the encoder backward GEMMs consume
template <typename T>
__global__ __launch_bounds__(256) void latent_grad_finish_kernel(long long rows, int Te, int D, float* __restrict__ dlat, const float* __restrict__ dpool,
                                                                 float inv_keep, T* __restrict__ dlat_lp) {
  const int dv = D >> 2;
  for (long long r = blockIdx.x; r < rows; r += gridDim.x) {
    const long long n = r / Te; const int t = (int)(r - n * Te);
    for (int c = threadIdx.x; c < dv; c += blockDim.x) {
      f4_t g = *reinterpret_cast<f4_t*>(dlat + r * D + c * 4);
      if (dpool && t > 0) { g += *reinterpret_cast<const f4_t*>(dpool + n * D + c * 4) * inv_keep; *reinterpret_cast<f4_t*>(dlat + r * D + c * 4) = g; }
      if (dlat_lp) st4<T>(dlat_lp + r * D + c * 4, g);
    }
  }
}
extern "C" int csmae_latent_grad_finish(int lp_dtype, long long B2, int Te, int D, float* dlat, const float* dpool, float inv_keep, void* dlat_lp, void* stream) {
  CSMAE_REQUIRE(B2 > 0 && Te > 0 && D % 4 == 0, "csmae_latent_grad_finish: bad geometry");
  long long rows = B2 * Te;
  dim3 grid((unsigned)fmin((double)rows, 4096.0)), block(D >= 1024 ? 256 : 128);
  hipStream_t st = (hipStream_t)stream;
  if (lp_dtype == CSMAE_BF16) hipLaunchKernelGGL((latent_grad_finish_kernel<bf16_t>), grid, block, 0, st, rows, Te, D, dlat, dpool, inv_keep, (bf16_t*)dlat_lp);
  else hipLaunchKernelGGL((latent_grad_finish_kernel<float>), grid, block, 0, st, rows, Te, D, dlat, dpool, inv_keep, (float*)dlat_lp);
  return csmae_check_launch("csmae_latent_grad_finish");
}

// ------------------------------------------------------------------------------------------ ssim family
// MAE_ViT_Shared.py:165-267 around pytorch-msssim 0.2.1 (`env.yml:118`): both operands are min-max scaled over the whole per-view
// tensor (scale_01 :94-95), un-patchified, multiplied by the patch mask, then compared with ssim(data_range 1, nonnegative) or the
// five-scale ms_ssim.  HBM-bound stencils: planes [B2*C][H][W] fp32 per level, an 11-tap separable gaussian ("valid" windows, the
// H axis first as the package does), 32x32 tiles staged through LDS.  All reductions are two-stage and deterministic.
//   workspace (floats): see SsimLayout.  X = prediction planes, Y = target planes, D = gradient w.r.t. X.
#define SSIM_WIN 11
#define SSIM_R (SSIM_WIN - 1)
#define SSIM_TILE 32
#define SSIM_MAX_LEVELS 5
struct SsimWin { float w[SSIM_WIN]; };
struct SsimLayout {
  int levels, H[SSIM_MAX_LEVELS], Ho[SSIM_MAX_LEVELS], tiles[SSIM_MAX_LEVELS], pad[SSIM_MAX_LEVELS];
  long long planes, X[SSIM_MAX_LEVELS], Y[SSIM_MAX_LEVELS], D[SSIM_MAX_LEVELS], part[SSIM_MAX_LEVELS], coef, val, mm, stat, total;
  // stat: [views][8] = pred lo, hi, target lo, hi, tie-term A, tie-term B, (int) count lo, (int) count hi
};
static SsimLayout ssim_layout(long long B2, int C, int S, int p, int levels) {
  SsimLayout L;
  L.levels = levels; L.planes = B2 * C;
  long long off = 0;
  auto take = [&](long long n) { long long o = off; off += (n + 3) & ~3ll; return o; };
  int h = S;
  for (int l = 0; l < SSIM_MAX_LEVELS; ++l) {
    L.H[l] = h; L.Ho[l] = h - SSIM_R; L.pad[l] = h & 1;
    const int t = cdiv(h - SSIM_R > 0 ? h - SSIM_R : 1, SSIM_TILE);
    L.tiles[l] = t * t;
    if (l < levels) {
      L.X[l] = take(L.planes * h * h); L.Y[l] = take(L.planes * h * h); L.D[l] = take(L.planes * h * h);
      L.part[l] = take(L.planes * L.tiles[l] * 2);
    } else L.X[l] = L.Y[l] = L.D[l] = L.part[l] = 0;
    h = (h + 2 * (h & 1) - 2) / 2 + 1;  // avg_pool2d(kernel 2, stride 2, padding = h % 2)
  }
  L.coef = take(L.planes * SSIM_MAX_LEVELS * 2);
  L.val = take(L.planes);
  const long long patches = B2 * (long long)(S / p) * (S / p);
  L.mm = take(patches * 2);
  L.stat = take(2 * 8);
  L.total = off;
  return L;
}
static SsimWin ssim_window() {  // pytorch-msssim `_fspecial_gauss_1d(11, 1.5)` in fp32
  SsimWin w; float s = 0.f;
  for (int i = 0; i < SSIM_WIN; ++i) { float c = (float)(i - SSIM_WIN / 2); w.w[i] = expf(-(c * c) / (2.f * 1.5f * 1.5f)); s += w.w[i]; }
  for (int i = 0; i < SSIM_WIN; ++i) w.w[i] /= s;
  return w;
}

// per-patch min / max of the prediction rows (cls row excluded, pad columns excluded)
__global__ __launch_bounds__(256) void pred_minmax_kernel(PatchGeom g, long long patches, const float* __restrict__ pred, long long ldp, float* __restrict__ mm) {
  const int lane = threadIdx.x & 63;
  const long long pt = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pt >= patches) return;
  const long long n2 = pt / g.L; const int l = (int)(pt - n2 * g.L);
  const float* pr = pred + (n2 * (g.L + 1) + 1 + l) * ldp;
  float lo = INFINITY, hi = -INFINITY;
  for (int e = lane; e < g.P; e += 64) { float v = pr[e]; lo = fminf(lo, v); hi = fmaxf(hi, v); }
  lo = -wave_max(-lo); hi = wave_max(hi);
  if (lane == 0) { mm[pt * 2] = lo; mm[pt * 2 + 1] = hi; }
}
__global__ __launch_bounds__(256) void ssim_stat_store_kernel(int views, int which, const float* __restrict__ mmout, float* __restrict__ stat) {
  if (threadIdx.x < views * 2) { int v = threadIdx.x >> 1, k = threadIdx.x & 1; stat[v * 8 + which * 2 + k] = mmout[v * 2 + k]; }
  if (which == 0 && threadIdx.x < views * 2) reinterpret_cast<int*>(stat)[(threadIdx.x >> 1) * 8 + 6 + (threadIdx.x & 1)] = 0;
}
// level-0 planes: X = mask * scale_01(pred), Y = mask * scale_01(target); counts the elements that attain the prediction's min / max
// (the backward of x.min() / x.max() spreads its gradient evenly over ties)
__global__ __launch_bounds__(256) void ssim_prepare_kernel(PatchGeom g, int norm_pix, long long patches, const float* __restrict__ img0,
                                                           const float* __restrict__ img1, const float* __restrict__ pred, long long ldp,
                                                           const float* __restrict__ mask, float* __restrict__ stat, float* __restrict__ X,
                                                           float* __restrict__ Y, int raw) {
  const int lane = threadIdx.x & 63;
  const long long pt = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pt >= patches) return;
  const long long n2 = pt / g.L; const int l = (int)(pt - n2 * g.L), v = (int)(n2 / g.N);
  const float* img = patch_img(g, img0, img1, n2);
  float mu = 0.f, rs = 1.f;
  if (norm_pix) patch_stats(g, img, l, lane, mu, rs);
  // raw: the operands are compared as they are (util/metrics.py: images already in [0, 1]); otherwise scale_01 of each (loss family)
  const float plo = raw ? 0.f : stat[v * 8], phi = stat[v * 8 + 1], tlo = raw ? 0.f : stat[v * 8 + 2], thi = stat[v * 8 + 3];
  const float psc = raw ? 1.f : 1.f / (phi - plo + 1.0e-6f), tsc = raw ? 1.f : 1.f / (thi - tlo + 1.0e-6f);
  const float m = mask ? mask[pt] : 1.f;
  const float* pr = pred + (n2 * (g.L + 1) + 1 + l) * ldp;
  const int gh = l / g.G, gw = l - gh * g.G;
  int nlo = 0, nhi = 0;
  const int pp = g.p * g.p;
  for (int c = 0; c < g.C; ++c)            // channel-major walk: a wave's stores are whole p-pixel row segments of ONE plane (the element
    for (int r = lane; r < pp; r += 64) {  // order of a patch row interleaves the channels; its 12-byte-strided reads hit the cache)
      const int ph = r / g.p, pw = r - ph * g.p, e = r * g.C + c;
      const long long o = ((n2 * g.C + c) * g.S + gh * g.p + ph) * g.S + gw * g.p + pw;
      const float pv = pr[e];
      nlo += pv == plo; nhi += pv == phi;
      X[o] = (pv - plo) * psc * m;
      Y[o] = ((img[((long long)c * g.S + gh * g.p + ph) * g.S + gw * g.p + pw] - mu) * rs - tlo) * tsc * m;
    }
  if (nlo) atomicAdd(reinterpret_cast<int*>(stat) + v * 8 + 6, nlo);
  if (nhi) atomicAdd(reinterpret_cast<int*>(stat) + v * 8 + 7, nhi);
}
// 2x2 average pooling with `pad` rows / columns of zeros in front (count_include_pad): both operands of one level
__global__ __launch_bounds__(256) void ssim_pool_kernel(long long planes, int H, int pad, int Hn, const float* __restrict__ X, const float* __restrict__ Y,
                                                        float* __restrict__ Xn, float* __restrict__ Yn) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= planes * Hn * Hn) return;
  const long long pl = i / ((long long)Hn * Hn); const int r = (int)(i - pl * Hn * Hn), y = r / Hn, x = r - y * Hn;
  const float* px = X + pl * H * H; const float* py = Y + pl * H * H;
  float sx = 0.f, sy = 0.f;
#pragma unroll
  for (int dy = 0; dy < 2; ++dy)
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      const int yy = 2 * y - pad + dy, xx = 2 * x - pad + dx;
      if (yy >= 0 && yy < H && xx >= 0 && xx < H) { sx += px[yy * H + xx]; sy += py[yy * H + xx]; }
    }
  Xn[i] = 0.25f * sx; Yn[i] = 0.25f * sy;
}

#define SSIM_C1 1.0e-4f   // (0.01 * data_range)^2, data_range = 1
#define SSIM_C2 9.0e-4f   // (0.03 * data_range)^2
// 11-tap FIR over a register window: four consecutive outputs from fourteen consecutive inputs (each LDS value is read once per
// four outputs instead of once per tap).  Every element is a PAIR of independent signals (two adjacent columns in the passes along
// H, two adjacent rows in the passes along W), so that the multiply-adds are gfx950's packed fp32 instructions: these kernels are
// bound by VALU issue, not by HBM or LDS.
__device__ __forceinline__ void fir4(const SsimWin& w, const f2_t (&in)[4 + SSIM_R], f2_t (&out)[4]) {
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    f2_t a = in[o] * w.w[0];
#pragma unroll
    for (int k = 1; k < SSIM_WIN; ++k) a += in[o + k] * w.w[k];
    out[o] = a;
  }
}
__device__ __forceinline__ f2_t rcp2(f2_t v) { return f2_t{__builtin_amdgcn_rcpf(v[0]), __builtin_amdgcn_rcpf(v[1])}; }
// stage a (rows x 2*pairs) window of two planes into LDS (zero outside the plane); 8-byte loads when the plane rows allow it
template <int ROWS, int PAIRS, int STRIDE>
__device__ __forceinline__ void ssim_stage(const float* __restrict__ px, const float* __restrict__ py, int H, int gy0, int gx0, float* sx, float* sy) {
  const bool vec = (H & 1) == 0;   // gx0 is even: a pair is 8-byte aligned and lies inside or outside the plane as a whole
  for (int i = threadIdx.x; i < ROWS * PAIRS; i += 256) {
    const int r = i / PAIRS, cp = i - r * PAIRS, gy = gy0 + r, gx = gx0 + 2 * cp;
    f2_t x = {0.f, 0.f}, y = {0.f, 0.f};
    if (gy >= 0 && gy < H) {
      if (vec) { if (gx >= 0 && gx < H) { x = *reinterpret_cast<const f2_t*>(px + (long long)gy * H + gx); y = *reinterpret_cast<const f2_t*>(py + (long long)gy * H + gx); } }
      else {
        if (gx >= 0 && gx < H) { x[0] = px[(long long)gy * H + gx]; y[0] = py[(long long)gy * H + gx]; }
        if (gx + 1 >= 0 && gx + 1 < H) { x[1] = px[(long long)gy * H + gx + 1]; y[1] = py[(long long)gy * H + gx + 1]; }
      }
    }
    *reinterpret_cast<f2_t*>(sx + r * STRIDE + 2 * cp) = x;
    *reinterpret_cast<f2_t*>(sy + r * STRIDE + 2 * cp) = y;
  }
}
// pass along H over staged planes: for NRG groups of four rows and NCP column pairs, the five filtered quantities
// (x, y, x^2, y^2, xy) -> V[m][row][col]
template <int NRG, int NCP, int SIN, int SOUT, int VROWS>
__device__ __forceinline__ void ssim_pass_h(const SsimWin& win, const float* sx, const float* sy, float* V) {
  for (int i = threadIdx.x; i < NRG * NCP; i += 256) {
    const int rg = i / NCP, cp = i - rg * NCP;
    f2_t x[4 + SSIM_R], y[4 + SSIM_R], t[4 + SSIM_R], o[4];
#pragma unroll
    for (int k = 0; k < 4 + SSIM_R; ++k) {
      x[k] = *reinterpret_cast<const f2_t*>(sx + (rg * 4 + k) * SIN + 2 * cp);
      y[k] = *reinterpret_cast<const f2_t*>(sy + (rg * 4 + k) * SIN + 2 * cp);
    }
    auto put = [&](int m) {
#pragma unroll
      for (int j = 0; j < 4; ++j) *reinterpret_cast<f2_t*>(V + (m * VROWS + rg * 4 + j) * SOUT + 2 * cp) = o[j];
    };
    fir4(win, x, o); put(0);
    fir4(win, y, o); put(1);
#pragma unroll
    for (int k = 0; k < 4 + SSIM_R; ++k) t[k] = x[k] * x[k];
    fir4(win, t, o); put(2);
#pragma unroll
    for (int k = 0; k < 4 + SSIM_R; ++k) t[k] = y[k] * y[k];
    fir4(win, t, o); put(3);
#pragma unroll
    for (int k = 0; k < 4 + SSIM_R; ++k) t[k] = x[k] * y[k];
    fir4(win, t, o); put(4);
  }
}
// pass along W for one (row pair rp, column group c0): NM maps of V -> f[m][4] (element = the two rows)
template <int NM, int SV, int VROWS>
__device__ __forceinline__ void ssim_pass_w(const SsimWin& win, const float* V, int rp, int c0, f2_t (&f)[NM][4]) {
#pragma unroll
  for (int m = 0; m < NM; ++m) {
    f2_t in[4 + SSIM_R];
    const float* v0 = V + (m * VROWS + 2 * rp) * SV + c0;
#pragma unroll
    for (int k = 0; k < 4 + SSIM_R; ++k) in[k] = f2_t{v0[k], v0[SV + k]};
    fir4(win, in, f[m]);
  }
}
// One 32x32 tile of the SSIM / contrast-structure maps of one plane -> part[plane][tile] = (sum ssim_map, sum cs_map)
__global__ __launch_bounds__(256) void ssim_level_fwd_kernel(SsimWin win, int H, int Ho, int tiles_x, const float* __restrict__ X, const float* __restrict__ Y,
                                                             float* __restrict__ part) {
  constexpr int T = SSIM_TILE, E = T + SSIM_R, ES = 44;  // 42 staged rows / columns, row stride 44 floats
  __shared__ __attribute__((aligned(16))) float sx[E * ES], sy[E * ES], V[5 * T * ES];
  __shared__ float red[32];
  const long long pl = blockIdx.y;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x, y0 = ty * T, x0 = tx * T;
  ssim_stage<E, E / 2, ES>(X + pl * H * H, Y + pl * H * H, H, y0, x0, sx, sy);
  __syncthreads();
  ssim_pass_h<T / 4, E / 2, ES, ES, T>(win, sx, sy, V);
  __syncthreads();
  float ss = 0.f, sc = 0.f;
  if (threadIdx.x < (T / 2) * (T / 4)) {  // 16 row pairs x 8 column groups
    const int rp = threadIdx.x >> 3, c0 = (threadIdx.x & 7) * 4;
    f2_t f[5][4];
    ssim_pass_w<5, ES, T>(win, V, rp, c0, f);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f2_t mu1 = f[0][j], mu2 = f[1][j], m11 = mu1 * mu1, m22 = mu2 * mu2, m12 = mu1 * mu2;
      const f2_t s1 = f[2][j] - m11, s2 = f[3][j] - m22, s12 = f[4][j] - m12;
      const f2_t cs = (s12 * 2.f + SSIM_C2) * rcp2(s1 + s2 + SSIM_C2);
      const f2_t sm = (m12 * 2.f + SSIM_C1) * rcp2(m11 + m22 + SSIM_C1) * cs;
      if (x0 + c0 + j < Ho) {
        if (y0 + 2 * rp < Ho) { ss += sm[0]; sc += cs[0]; }
        if (y0 + 2 * rp + 1 < Ho) { ss += sm[1]; sc += cs[1]; }
      }
    }
  }
  ss = block_sum(ss, red); sc = block_sum(sc, red);
  if (threadIdx.x == 0) { part[(pl * gridDim.x + blockIdx.x) * 2] = ss; part[(pl * gridDim.x + blockIdx.x) * 2 + 1] = sc; }
}
// Per-plane means -> per-plane score -> per-view loss term, and the coefficients the backward needs:
//   d(term_v) / d(mean ssim_map of level l, plane) = coef[plane][l][0],  d / d(mean cs_map) = coef[plane][l][1]   (already / Ho^2)
// ssim: relu(mean) per plane (nonnegative_ssim), averaged.  ms_ssim: prod_l relu(.)^w_l with cs for l < 4 and ssim for l = 4; a
// clamped factor zeroes the product and (threshold backward selects 0) every gradient of that plane.
struct SsimStatArgs { int levels, tiles[SSIM_MAX_LEVELS], Ho[SSIM_MAX_LEVELS]; const float* part[SSIM_MAX_LEVELS]; };
__global__ __launch_bounds__(256) void ssim_stats_kernel(SsimStatArgs a, long long planes, int views, float* __restrict__ coef, float* __restrict__ val,
                                                         int signed_ssim) {
  const float wts[5] = {0.0448f, 0.2856f, 0.3001f, 0.2363f, 0.1333f};
  const long long per_view = planes / views;
  // one wave per plane: the tile partials of a level are summed across the lanes (fixed order: deterministic)
  const long long pl = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (pl >= planes) return;
  float ms[SSIM_MAX_LEVELS], mc[SSIM_MAX_LEVELS];
#pragma unroll
  for (int l = 0; l < SSIM_MAX_LEVELS; ++l) {
    ms[l] = mc[l] = 0.f;
    if (l < a.levels) {
      float s = 0.f, c = 0.f;
      for (int t = lane; t < a.tiles[l]; t += 64) { s += a.part[l][(pl * a.tiles[l] + t) * 2]; c += a.part[l][(pl * a.tiles[l] + t) * 2 + 1]; }
      const float inv = 1.f / ((float)a.Ho[l] * a.Ho[l]);
      ms[l] = wave_sum(s) * inv; mc[l] = wave_sum(c) * inv;
    }
  }
  if (lane != 0) return;
  float cf[SSIM_MAX_LEVELS * 2];
#pragma unroll
  for (int l = 0; l < SSIM_MAX_LEVELS * 2; ++l) cf[l] = 0.f;
  const float base = -1.f / (float)per_view;
  float v;
  if (a.levels == 1) {
    v = signed_ssim ? ms[0] : fmaxf(ms[0], 0.f);   // nonnegative_ssim=True in the loss (MAE_ViT_Shared.py:204-206), False in util/metrics.py
    cf[0] = (signed_ssim || ms[0] > 0.f) ? base / ((float)a.Ho[0] * a.Ho[0]) : 0.f;
  } else {
    float t[SSIM_MAX_LEVELS]; bool pos = true;
    v = 1.f;
#pragma unroll
    for (int l = 0; l < SSIM_MAX_LEVELS; ++l) { t[l] = fmaxf(l == SSIM_MAX_LEVELS - 1 ? ms[l] : mc[l], 0.f); pos &= t[l] > 0.f; v *= powf(t[l], wts[l]); }
    if (!pos) v = 0.f;
#pragma unroll
    for (int l = 0; l < SSIM_MAX_LEVELS; ++l)
      cf[l * 2 + (l == SSIM_MAX_LEVELS - 1 ? 0 : 1)] = pos ? base * wts[l] * v / t[l] / ((float)a.Ho[l] * a.Ho[l]) : 0.f;
  }
#pragma unroll
  for (int l = 0; l < SSIM_MAX_LEVELS * 2; ++l) coef[pl * SSIM_MAX_LEVELS * 2 + l] = cf[l];
  val[pl] = v;
}
__global__ __launch_bounds__(1024) void ssim_terms_kernel(long long per_view, int views, const float* __restrict__ val, float* __restrict__ terms) {
  __shared__ float red[32];
  for (int vw = 0; vw < views; ++vw) {
    float s = 0.f;
    for (long long i = threadIdx.x; i < per_view; i += blockDim.x) s += val[vw * per_view + i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) terms[vw] = 1.f - s / (float)per_view;
  }
}
// Gradient of one level w.r.t. its X plane, for one 32x32 tile of pixels:
//   dX(p) = sum_q w(p - q) [G0(q) + 2 X(p) G1(q) + Y(p) G2(q)]  (+ 1/4 of the next level's gradient at the pooled position)
// with, at every window position q (F = cs * (a * lum + b) is what the plane's score depends on):
//   c = a lum + b,  G1 = dF/dE[xx] = -c cs / B2,  G2 = dF/dE[xy] = 2 c / B2,  G0 = dF/dmu1 = a cs (2 mu2 - 2 lum mu1) / B1 - 2 mu1 G1 - mu2 G2
__global__ __launch_bounds__(256) void ssim_level_bwd_kernel(SsimWin win, int H, int Ho, int tiles_x, int lvl, const float* __restrict__ X,
                                                             const float* __restrict__ Y, const float* __restrict__ coef,
                                                             const float* __restrict__ Dn, int Hn, int pad, float* __restrict__ D) {
  // window positions q of this tile: 42 x 42 (q = p - 10 .. p), padded to 44 so that every pass works on groups of four
  constexpr int T = SSIM_TILE, E1P = 44, E2 = E1P + SSIM_R, S2 = 56, S1 = 44;
  __shared__ __attribute__((aligned(16))) float sxy[2 * E2 * S2];  // X | Y over the 54 x 54 halo region, later G[3][44][44]
  __shared__ __attribute__((aligned(16))) float V[5 * E1P * S2];   // H-filtered quantities [5][44][56], later Tt[3][32][44]
  static_assert(3 * E1P * S1 <= 2 * E2 * S2 && 3 * T * S1 <= 5 * E1P * S2, "aliases fit");
  float* sx = sxy; float* sy = sxy + E2 * S2;
  const long long pl = blockIdx.y;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x, y0 = ty * T, x0 = tx * T;
  const float* px = X + pl * H * H; const float* py = Y + pl * H * H;
  const float ca = coef[pl * SSIM_MAX_LEVELS * 2 + lvl * 2], cb = coef[pl * SSIM_MAX_LEVELS * 2 + lvl * 2 + 1];
  ssim_stage<E2, E2 / 2, S2>(px, py, H, y0 - SSIM_R, x0 - SSIM_R, sx, sy);
  __syncthreads();
  ssim_pass_h<E1P / 4, E2 / 2, S2, S2, E1P>(win, sx, sy, V);   // window rows q = y0 - 10 + r
  __syncthreads();
  float* G = sxy;
  if (threadIdx.x < (E1P / 2) * (E1P / 4)) {  // 22 row pairs x 11 column groups: along W, then the three coefficient maps
    const int rp = threadIdx.x / (E1P / 4), c0 = (threadIdx.x - rp * (E1P / 4)) * 4, qy = y0 - SSIM_R + 2 * rp;
    f2_t f[5][4];
    ssim_pass_w<5, S2, E1P>(win, V, rp, c0, f);
    const bool oky0 = qy >= 0 && qy < Ho, oky1 = qy + 1 >= 0 && qy + 1 < Ho;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int qx = x0 - SSIM_R + c0 + j;
      const f2_t mu1 = f[0][j], mu2 = f[1][j], m11 = mu1 * mu1, m22 = mu2 * mu2, m12 = mu1 * mu2;
      const f2_t s1 = f[2][j] - m11, s2 = f[3][j] - m22, s12 = f[4][j] - m12;
      const f2_t rB1 = rcp2(m11 + m22 + SSIM_C1), rB2 = rcp2(s1 + s2 + SSIM_C2);
      const f2_t lum = (m12 * 2.f + SSIM_C1) * rB1, cs = (s12 * 2.f + SSIM_C2) * rB2;
      const f2_t cc = lum * ca + cb;
      f2_t g1 = -cc * cs * rB2;
      f2_t g2 = cc * rB2 * 2.f;
      f2_t g0 = cs * (mu2 - lum * mu1) * rB1 * (2.f * ca) - mu1 * g1 * 2.f - mu2 * g2;
      const bool okx = qx >= 0 && qx < Ho;
      if (!(okx && oky0)) { g0[0] = 0.f; g1[0] = 0.f; g2[0] = 0.f; }
      if (!(okx && oky1)) { g0[1] = 0.f; g1[1] = 0.f; g2[1] = 0.f; }
      float* g = G + (2 * rp) * S1 + c0 + j;
      g[0] = g0[0]; g[S1] = g0[1];
      g[E1P * S1] = g1[0]; g[E1P * S1 + S1] = g1[1];
      g[2 * E1P * S1] = g2[0]; g[2 * E1P * S1 + S1] = g2[1];
    }
  }
  __syncthreads();
  float* Tt = V;
  for (int i = threadIdx.x; i < 3 * (T / 4) * (E1P / 2); i += 256) {  // transposed filter along H: pixel rows p = y0 + r take windows q = p - k
    const int m = i / ((T / 4) * (E1P / 2)), j = i - m * (T / 4) * (E1P / 2), rg = j / (E1P / 2), cp = j - rg * (E1P / 2);
    f2_t in[4 + SSIM_R], o[4];
#pragma unroll
    for (int k = 0; k < 4 + SSIM_R; ++k) in[k] = *reinterpret_cast<const f2_t*>(G + (m * E1P + rg * 4 + k) * S1 + 2 * cp);   // (symmetric window: G rows r .. r + 10 of pixel row r)
    fir4(win, in, o);
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) *reinterpret_cast<f2_t*>(Tt + (m * T + rg * 4 + jj) * S1 + 2 * cp) = o[jj];
  }
  __syncthreads();
  if (threadIdx.x < (T / 2) * (T / 4)) {
    float* pd = D + pl * H * H;
    const float* pn = Dn ? Dn + pl * Hn * Hn : nullptr;
    const int rp = threadIdx.x >> 3, c0 = (threadIdx.x & 7) * 4;
    f2_t o[3][4];
    ssim_pass_w<3, S1, T>(win, Tt, rp, c0, o);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int gy = y0 + 2 * rp + h;
      if (gy >= H) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int gx = x0 + c0 + j;
        if (gx >= H) continue;
        float d = o[0][j][h] + 2.f * px[(long long)gy * H + gx] * o[1][j][h] + py[(long long)gy * H + gx] * o[2][j][h];
        if (pn) { const int yy = (gy + pad) >> 1, xx = (gx + pad) >> 1; if (yy < Hn && xx < Hn) d += 0.25f * pn[yy * Hn + xx]; }
        pd[(long long)gy * H + gx] = d;
      }
    }
  }
}
// extra[pt][e] = gout * scale * dX0 * mask / range  (d loss / d pred through the scaled value), per-patch partial sums of the two
// terms that flow into the tensor's min and max:  A = sum dxs (xs - 1) / range,  B = -sum dxs xs / range
__global__ __launch_bounds__(256) void ssim_pred_bwd_kernel(PatchGeom g, long long patches, const float* __restrict__ pred, long long ldp,
                                                            const float* __restrict__ mask, const float* __restrict__ stat, const float* __restrict__ D0,
                                                            const float* __restrict__ gout, float scale, float* __restrict__ extra, float* __restrict__ ab) {
  const int lane = threadIdx.x & 63;
  const long long pt = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pt >= patches) return;
  const long long n2 = pt / g.L; const int l = (int)(pt - n2 * g.L), v = (int)(n2 / g.N);
  const float m = mask ? mask[pt] : 1.f;
  float* ex = extra + pt * g.P;
  float A = 0.f, B = 0.f;
  if (m == 0.f) { for (int e = lane; e < g.P; e += 64) ex[e] = 0.f; }
  else {
    const float plo = stat[v * 8], psc = 1.f / (stat[v * 8 + 1] - plo + 1.0e-6f), gs = gout[0] * scale * m;
    const float* pr = pred + (n2 * (g.L + 1) + 1 + l) * ldp;
    const int gh = l / g.G, gw = l - gh * g.G;
    for (int e = lane; e < g.P; e += 64) {
      const int c = e % g.C, r = e / g.C, ph = r / g.p, pw = r - ph * g.p;
      const float dxs = D0[((n2 * g.C + c) * g.S + gh * g.p + ph) * g.S + gw * g.p + pw] * gs;
      const float xs = (pr[e] - plo) * psc;
      ex[e] = dxs * psc;
      A += dxs * (xs - 1.f) * psc; B -= dxs * xs * psc;
    }
  }
  A = wave_sum(A); B = wave_sum(B);
  if (lane == 0) { ab[pt * 2] = A; ab[pt * 2 + 1] = B; }
}
__global__ __launch_bounds__(1024) void ssim_tie_reduce_kernel(long long per_view, int views, const float* __restrict__ ab, float* __restrict__ stat) {
  __shared__ float red[32];
  for (int v = 0; v < views; ++v) {
    float A = 0.f, B = 0.f;
    for (long long i = threadIdx.x; i < per_view; i += blockDim.x) { A += ab[(v * per_view + i) * 2]; B += ab[(v * per_view + i) * 2 + 1]; }
    A = block_sum(A, red); B = block_sum(B, red);
    if (threadIdx.x == 0) {
      const int* cnt = reinterpret_cast<const int*>(stat) + v * 8 + 6;
      stat[v * 8 + 4] = A / (float)max(cnt[0], 1); stat[v * 8 + 5] = B / (float)max(cnt[1], 1);
    }
  }
}
__global__ __launch_bounds__(256) void ssim_tie_apply_kernel(PatchGeom g, long long patches, const float* __restrict__ pred, long long ldp,
                                                             const float* __restrict__ stat, float* __restrict__ extra) {
  const int lane = threadIdx.x & 63;
  const long long pt = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pt >= patches) return;
  const long long n2 = pt / g.L; const int l = (int)(pt - n2 * g.L), v = (int)(n2 / g.N);
  const float plo = stat[v * 8], phi = stat[v * 8 + 1], A = stat[v * 8 + 4], B = stat[v * 8 + 5];
  const float* pr = pred + (n2 * (g.L + 1) + 1 + l) * ldp;
  for (int e = lane; e < g.P; e += 64) {
    const float pv = pr[e];
    if (pv == plo || pv == phi) extra[pt * g.P + e] += (pv == plo ? A : 0.f) + (pv == phi ? B : 0.f);
  }
}
// losses[] patch-up after csmae_loss_finalize: the ssim term of each view joins (weight 0.1, the mse_* kinds) or replaces the
// masked per-patch term
__global__ void ssim_apply_kernel(int pure, int views, float weight, float recon_scale, const float* __restrict__ terms, float* __restrict__ losses) {
  if (threadIdx.x != 0) return;
  float add = 0.f, old = 0.f;
  for (int v = 0; v < views; ++v) {
    const float t = weight * terms[v];
    if (pure) { old += losses[1 + v]; losses[1 + v] = t; } else losses[1 + v] += t;
    add += t;
  }
  losses[0] = (pure ? losses[3] + losses[4] + losses[5] : losses[0]) + recon_scale * add;
  (void)old;
}

extern "C" int csmae_ssim_workspace_floats(long long B2, int C, int S, int p, int levels, long long* floats) {
  CSMAE_REQUIRE(B2 > 0 && C > 0 && S > 0 && p > 0 && S % p == 0 && (levels == 1 || levels == SSIM_MAX_LEVELS) && floats, "csmae_ssim_workspace_floats: bad args");
  *floats = ssim_layout(B2, C, S, p, levels).total;
  return CSMAE_OK;
}
extern "C" int csmae_ssim_fwd(int levels, int flags, int norm_pix, long long B2, int N, int C, int S, int p, const float* img0, const float* img1,
                              const float* pred, long long ldp, const float* mask, float* ws, float* terms, void* stream) {
  CSMAE_REQUIRE(flags >= 0 && flags <= 3, "csmae_ssim_fwd: bad flags %d", flags);
  CSMAE_REQUIRE(levels == 1 || levels == SSIM_MAX_LEVELS, "csmae_ssim_fwd: levels must be 1 (ssim) or 5 (ms_ssim)");
  CSMAE_REQUIRE(B2 > 0 && N > 0 && B2 % N == 0 && B2 / N <= 2 && S % p == 0 && ws && terms && pred && img0, "csmae_ssim_fwd: bad args");
  CSMAE_REQUIRE(S >= SSIM_WIN, "csmae_ssim_fwd: images smaller than the 11-tap window are not supported (S = %d)", S);
  CSMAE_REQUIRE(levels == 1 || S > SSIM_R * 16, "csmae_ssim_fwd: Image size should be larger than 160 due to the 4 downsamplings in ms-ssim (S = %d)", S);
  hipStream_t st = (hipStream_t)stream;
  const PatchGeom g = make_geom(N, C, S, p);
  const SsimLayout L = ssim_layout(B2, C, S, p, levels);
  const SsimWin win = ssim_window();
  const long long patches = B2 * g.L;
  const int views = (int)(B2 / N);
  float* stat = ws + L.stat; float* mm = ws + L.mm; float* mmout = ws + L.val;  // (val is free until the stats kernel)
  hipLaunchKernelGGL(pred_minmax_kernel, dim3(cdiv(patches, 4)), dim3(256), 0, st, g, patches, pred, ldp, mm);
  hipLaunchKernelGGL(minmax_reduce_kernel, dim3(1), dim3(1024), 0, st, (long long)N * g.L, views, mm, mmout);
  hipLaunchKernelGGL(ssim_stat_store_kernel, dim3(1), dim3(64), 0, st, views, 0, mmout, stat);
  hipLaunchKernelGGL(target_minmax_kernel, dim3(cdiv(patches, 4)), dim3(256), 0, st, g, norm_pix, patches, img0, img1, mm);
  hipLaunchKernelGGL(minmax_reduce_kernel, dim3(1), dim3(1024), 0, st, (long long)N * g.L, views, mm, mmout);
  hipLaunchKernelGGL(ssim_stat_store_kernel, dim3(1), dim3(64), 0, st, views, 1, mmout, stat);
  hipLaunchKernelGGL(ssim_prepare_kernel, dim3(cdiv(patches, 4)), dim3(256), 0, st, g, norm_pix, patches, img0, img1, pred, ldp, mask, stat, ws + L.X[0], ws + L.Y[0], flags & 1);
  SsimStatArgs sa; sa.levels = levels;
  for (int l = 0; l < SSIM_MAX_LEVELS; ++l) { sa.tiles[l] = L.tiles[l]; sa.Ho[l] = L.Ho[l]; sa.part[l] = ws + L.part[l]; }
  for (int l = 0; l < levels; ++l) {
    const int tx = cdiv(L.Ho[l], SSIM_TILE);
    hipLaunchKernelGGL(ssim_level_fwd_kernel, dim3(tx * tx, (unsigned)L.planes), dim3(256), 0, st, win, L.H[l], L.Ho[l], tx, ws + L.X[l], ws + L.Y[l], ws + L.part[l]);
    if (l + 1 < levels)
      hipLaunchKernelGGL(ssim_pool_kernel, dim3(cdiv(L.planes * L.H[l + 1] * L.H[l + 1], 256)), dim3(256), 0, st, L.planes, L.H[l], L.pad[l], L.H[l + 1],
                         ws + L.X[l], ws + L.Y[l], ws + L.X[l + 1], ws + L.Y[l + 1]);
  }
  hipLaunchKernelGGL(ssim_stats_kernel, dim3(cdiv(L.planes, 4)), dim3(256), 0, st, sa, L.planes, views, ws + L.coef, ws + L.val, (flags >> 1) & 1);
  hipLaunchKernelGGL(ssim_terms_kernel, dim3(1), dim3(1024), 0, st, L.planes / views, views, ws + L.val, terms);
  return csmae_check_launch("csmae_ssim_fwd");
}
extern "C" int csmae_ssim_apply(int pure, int views, float weight, float recon_scale, const float* terms, float* losses, void* stream) {
  CSMAE_REQUIRE((views == 1 || views == 2) && terms && losses, "csmae_ssim_apply: bad args");
  hipLaunchKernelGGL(ssim_apply_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, pure, views, weight, recon_scale, terms, losses);
  return csmae_check_launch("csmae_ssim_apply");
}
extern "C" int csmae_ssim_bwd(int levels, long long B2, int N, int C, int S, int p, const float* pred, long long ldp, const float* mask,
                              const float* gout, float scale, float* ws, float* extra, void* stream) {
  CSMAE_REQUIRE(levels == 1 || levels == SSIM_MAX_LEVELS, "csmae_ssim_bwd: levels must be 1 (ssim) or 5 (ms_ssim)");
  CSMAE_REQUIRE(B2 > 0 && N > 0 && B2 % N == 0 && B2 / N <= 2 && S % p == 0 && ws && extra && pred && gout, "csmae_ssim_bwd: bad args");
  hipStream_t st = (hipStream_t)stream;
  const PatchGeom g = make_geom(N, C, S, p);
  const SsimLayout L = ssim_layout(B2, C, S, p, levels);
  const SsimWin win = ssim_window();
  const long long patches = B2 * g.L;
  const int views = (int)(B2 / N);
  for (int l = levels - 1; l >= 0; --l) {
    const int tx = cdiv(L.H[l], SSIM_TILE);
    const bool nxt = l + 1 < levels;
    hipLaunchKernelGGL(ssim_level_bwd_kernel, dim3(tx * tx, (unsigned)L.planes), dim3(256), 0, st, win, L.H[l], L.Ho[l], tx, l, ws + L.X[l], ws + L.Y[l],
                       ws + L.coef, nxt ? ws + L.D[l + 1] : nullptr, nxt ? L.H[l + 1] : 0, L.pad[l], ws + L.D[l]);
  }
  hipLaunchKernelGGL(ssim_pred_bwd_kernel, dim3(cdiv(patches, 4)), dim3(256), 0, st, g, patches, pred, ldp, mask, ws + L.stat, ws + L.D[0], gout, scale, extra, ws + L.mm);
  hipLaunchKernelGGL(ssim_tie_reduce_kernel, dim3(1), dim3(1024), 0, st, (long long)N * g.L, views, ws + L.mm, ws + L.stat);
  hipLaunchKernelGGL(ssim_tie_apply_kernel, dim3(cdiv(patches, 4)), dim3(256), 0, st, g, patches, pred, ldp, ws + L.stat, extra);
  return csmae_check_launch("csmae_ssim_bwd");
}

// ------------------------------------------------------------------------------------------ scalar assembly
// losses[0]=total [1]=recon orig [2]=recon crop [3]=cross-decoder [4]=contrastive [5]=latent [6]=sum(mask) orig [7]=sum(mask) crop
// One pass over everything with independent accumulators and 16-byte loads, ONE seven-value block reduction (the first form walked the two
// views one after the other with a dependent scalar chain each and ran seven block reductions of three barriers: 29 us on the critical path
// between forward and backward).  Fixed thread-to-element assignment and fold order: deterministic.
__global__ __launch_bounds__(1024) void finalize_kernel(long long per_view, int views, const float* __restrict__ rowloss, const float* __restrict__ mask,
                                                       float recon_scale, const float* __restrict__ cd_partial, float cd_scale,
                                                       const float* __restrict__ e_partial, float e_scale, const float* __restrict__ ce_rowloss,
                                                       int ce_rows, float* __restrict__ losses) {
  __shared__ float red[16][8];
  float v[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // num0, den0, num1, den1, cd, e, ce
  const long long total = per_view * views;
  if ((per_view & 3) == 0) {
    for (long long i = (long long)threadIdx.x * 4; i < total; i += 4096) {
      const f4_t r = *reinterpret_cast<const f4_t*>(rowloss + i), m = *reinterpret_cast<const f4_t*>(mask + i);
      const float num = (r[0] * m[0] + r[1] * m[1]) + (r[2] * m[2] + r[3] * m[3]), den = (m[0] + m[1]) + (m[2] + m[3]);
      if (i < per_view) { v[0] += num; v[1] += den; } else { v[2] += num; v[3] += den; }
    }
  } else {
    for (long long i = threadIdx.x; i < total; i += 1024) {
      const float m = mask[i], num = rowloss[i] * m;
      if (i < per_view) { v[0] += num; v[1] += m; } else { v[2] += num; v[3] += m; }
    }
  }
  if (cd_partial) for (int i = threadIdx.x; i < PAIR_BLOCKS; i += 1024) v[4] += cd_partial[i];
  if (e_partial) for (int i = threadIdx.x; i < PAIR_BLOCKS; i += 1024) v[5] += e_partial[i];
  if (ce_rowloss) for (int i = threadIdx.x; i < ce_rows; i += 1024) v[6] += ce_rowloss[i];
#pragma unroll
  for (int k = 0; k < 7; ++k) v[k] = wave_sum(v[k]);
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int k = 0; k < 7; ++k) red[threadIdx.x >> 6][k] = v[k];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float t[7];
    for (int k = 0; k < 7; ++k) { float a = 0.f; for (int w = 0; w < 16; ++w) a += red[w][k]; t[k] = a; }
    const float l0 = t[0] / t[1];                              // mask_ratio = 0 -> 0/0 = NaN, as in the reference (MAE_ViT_Shared.py:119)
    const float l1 = views > 1 ? t[2] / t[3] : 0.f;
    const float cd = cd_partial ? t[4] * cd_scale : 0.f, e = e_partial ? t[5] * e_scale : 0.f, ce = ce_rowloss ? t[6] / ce_rows : 0.f;
    losses[1] = l0; losses[2] = l1; losses[6] = t[1]; losses[7] = views > 1 ? t[3] : 0.f;
    losses[3] = cd; losses[4] = ce; losses[5] = e;
    losses[0] = (l0 * recon_scale + l1 * recon_scale) + cd + ce + e;
  }
}
extern "C" int csmae_loss_finalize(long long per_view, int views, const float* rowloss, const float* mask, float recon_scale,
                                   const float* cd_partial, float cd_scale, const float* e_partial, float e_scale,
                                   const float* ce_rowloss, int ce_rows, float* losses, void* stream) {
  CSMAE_REQUIRE(per_view > 0 && (views == 1 || views == 2) && losses, "csmae_loss_finalize: bad args");
  hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, per_view, views, rowloss, mask, recon_scale, cd_partial, cd_scale, e_partial, e_scale, ce_rowloss, ce_rows, losses);
  return csmae_check_launch("csmae_loss_finalize");
}
