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

__device__ __forceinline__ float elem_loss(int kind, float pred, float t) {
  if (kind == LOSS_MSE || kind == LOSS_L2) { float d = pred - t; return d * d; }
  if (kind == LOSS_MAE || kind == LOSS_L1) return fabsf(pred - t);
  // bce with logits, torch's stable form: max(x,0) - x*t + log1p(exp(-|x|))
  return fmaxf(pred, 0.f) - pred * t + log1pf(expf(-fabsf(pred)));
}
__device__ __forceinline__ float elem_grad(int kind, float pred, float t) {  // d elem_loss / d pred
  if (kind == LOSS_MSE || kind == LOSS_L2) return 2.f * (pred - t);
  if (kind == LOSS_MAE || kind == LOSS_L1) { float d = pred - t; return d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); }
  return 1.f / (1.f + expf(-pred)) - t;
}
__device__ __forceinline__ bool mean_over_last(int kind) { return kind == LOSS_MSE || kind == LOSS_MAE || kind == LOSS_BCE; }

// one wave per patch: target values of patch (n2, l); element e = (ph*p + pw)*C + c   (MAE_ViT_Shared.py:36-38 "nhwpqc")
struct PatchGeom { int N, C, S, p, L, G, P; };
__device__ __forceinline__ const float* patch_img(const PatchGeom& g, const float* img0, const float* img1, long long n2) {
  int view = (int)(n2 / g.N);
  return (view ? img1 : img0) + (n2 - (long long)view * g.N) * g.C * g.S * g.S;
}
__device__ __forceinline__ float patch_elem(const PatchGeom& g, const float* img, int l, int e) {
  int gh = l / g.G, gw = l - gh * g.G;
  int c = e % g.C, r = e / g.C, ph = r / g.p, pw = r - ph * g.p;
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
__global__ __launch_bounds__(256) void minmax_reduce_kernel(long long per_view, int views, const float* __restrict__ mm, float* __restrict__ out /*[views][2]*/) {
  __shared__ float slo[256], shi[256];
  for (int v = 0; v < views; ++v) {
    float lo = INFINITY, hi = -INFINITY;
    for (long long i = threadIdx.x; i < per_view; i += blockDim.x) { lo = fminf(lo, mm[(v * per_view + i) * 2]); hi = fmaxf(hi, mm[(v * per_view + i) * 2 + 1]); }
    slo[threadIdx.x] = lo; shi[threadIdx.x] = hi;
    __syncthreads();
    if (threadIdx.x == 0) { for (int i = 1; i < 256; ++i) { lo = fminf(lo, slo[i]); hi = fmaxf(hi, shi[i]); } out[v * 2] = lo; out[v * 2 + 1] = hi; }
    __syncthreads();
  }
}

// rowloss[n2*L + l] = mean/sum_e f(pred[n2, 1+l, e], target)
__global__ __launch_bounds__(256) void recon_fwd_kernel(PatchGeom g, int kind, int norm_pix, long long patches, const float* __restrict__ img0,
                                                        const float* __restrict__ img1, const float* __restrict__ pred, long long ldp,
                                                        const float* __restrict__ minmax, float* __restrict__ rowloss) {
  const int lane = threadIdx.x & 63;
  const long long pt = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pt >= patches) return;
  const long long n2 = pt / g.L; const int l = (int)(pt - n2 * g.L);
  const float* img = patch_img(g, img0, img1, n2);
  float mu = 0.f, rs = 1.f;
  if (norm_pix) patch_stats(g, img, l, lane, mu, rs);
  float lo = 0.f, sc = 1.f;
  if (kind == LOSS_BCE) { int v = (int)(n2 / g.N); lo = minmax[v * 2]; sc = 1.f / (minmax[v * 2 + 1] - lo + 1.0e-6f); }
  const float* pr = pred + (n2 * (g.L + 1) + 1 + l) * ldp;
  float s = 0.f;
  for (int e = lane; e < g.P; e += 64) {
    float t = (patch_elem(g, img, l, e) - mu) * rs;
    if (kind == LOSS_BCE) t = (t - lo) * sc;
    s += elem_loss(kind, pr[e], t);
  }
  s = wave_sum(s);
  if (lane == 0) rowloss[pt] = mean_over_last(kind) ? s / g.P : s;
}
// dpred[n2, 1+l, e] = gout * vscale * mask / masksum(view) * f'(pred, t) / (P or 1) ; cls rows and pad columns are zeroed
template <typename T>
__global__ __launch_bounds__(256) void recon_bwd_kernel(PatchGeom g, int kind, int norm_pix, long long rows, const float* __restrict__ img0,
                                                        const float* __restrict__ img1, const float* __restrict__ pred, long long ldp,
                                                        const float* __restrict__ minmax, const float* __restrict__ mask,
                                                        const float* __restrict__ losses, const float* __restrict__ gout, float vscale,
                                                        T* __restrict__ dpred, long long ldd) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);  // row of [B2*(L+1)]
  if (row >= rows) return;
  const long long n2 = row / (g.L + 1); const int j = (int)(row - n2 * (g.L + 1));
  T* dp = dpred + row * ldd;
  float m = j > 0 ? mask[n2 * g.L + j - 1] : 0.f;
  if (m == 0.f) { for (int e = lane; e < ldd; e += 64) st_from_f32<T>(dp + e, 0.f); return; }
  const int l = j - 1, v = (int)(n2 / g.N);
  const float* img = patch_img(g, img0, img1, n2);
  float mu = 0.f, rs = 1.f;
  if (norm_pix) patch_stats(g, img, l, lane, mu, rs);
  float lo = 0.f, sc = 1.f;
  if (kind == LOSS_BCE) { lo = minmax[v * 2]; sc = 1.f / (minmax[v * 2 + 1] - lo + 1.0e-6f); }
  const float coef = gout[0] * vscale * m / losses[6 + v] / (mean_over_last(kind) ? (float)g.P : 1.f);
  const float* pr = pred + row * ldp;
  for (int e = lane; e < ldd; e += 64) {
    float o = 0.f;
    if (e < g.P) {
      float t = (patch_elem(g, img, l, e) - mu) * rs;
      if (kind == LOSS_BCE) t = (t - lo) * sc;
      o = coef * elem_grad(kind, pr[e], t);
    }
    st_from_f32<T>(dp + e, o);
  }
}

static PatchGeom make_geom(int N, int C, int S, int p) { PatchGeom g; g.N = N; g.C = C; g.S = S; g.p = p; g.G = S / p; g.L = g.G * g.G; g.P = p * p * C; return g; }

extern "C" int csmae_target_minmax(int norm_pix, long long B2, int N, int C, int S, int p, const float* img0, const float* img1,
                                   float* scratch /*[B2*L*2]*/, float* out /*[views*2]*/, void* stream) {
  PatchGeom g = make_geom(N, C, S, p);
  long long patches = B2 * g.L;
  hipLaunchKernelGGL(target_minmax_kernel, dim3(cdiv(patches, 4)), dim3(256), 0, (hipStream_t)stream, g, norm_pix, patches, img0, img1, scratch);
  hipLaunchKernelGGL(minmax_reduce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (long long)N * g.L, (int)(B2 / N), scratch, out);
  return csmae_check_launch("csmae_target_minmax");
}
extern "C" int csmae_recon_loss_fwd(int kind, int norm_pix, long long B2, int N, int C, int S, int p, const float* img0, const float* img1,
                                    const float* pred, long long ldp, const float* minmax, float* rowloss, void* stream) {
  CSMAE_REQUIRE(kind >= LOSS_MSE && kind <= LOSS_BCE, "csmae_recon_loss_fwd: loss kind %d is outside the hot-path scope (ssim family: SURVEY §2 row 2)", kind);
  CSMAE_REQUIRE(B2 > 0 && N > 0 && B2 % N == 0 && S % p == 0 && (kind != LOSS_BCE || minmax), "csmae_recon_loss_fwd: bad args");
  PatchGeom g = make_geom(N, C, S, p);
  long long patches = B2 * g.L;
  hipLaunchKernelGGL(recon_fwd_kernel, dim3(cdiv(patches, 4)), dim3(256), 0, (hipStream_t)stream, g, kind, norm_pix, patches, img0, img1, pred, ldp, minmax, rowloss);
  return csmae_check_launch("csmae_recon_loss_fwd");
}
extern "C" int csmae_recon_loss_bwd(int kind, int norm_pix, int out_dtype, long long B2, int N, int C, int S, int p, const float* img0,
                                    const float* img1, const float* pred, long long ldp, const float* minmax, const float* mask,
                                    const float* losses, const float* gout, float vscale, void* dpred, long long ldd, void* stream) {
  CSMAE_REQUIRE(kind >= LOSS_MSE && kind <= LOSS_BCE, "csmae_recon_loss_bwd: bad loss kind %d", kind);
  PatchGeom g = make_geom(N, C, S, p);
  long long rows = B2 * (g.L + 1);
  hipStream_t st = (hipStream_t)stream;
  if (out_dtype == CSMAE_BF16) hipLaunchKernelGGL((recon_bwd_kernel<bf16_t>), dim3(cdiv(rows, 4)), dim3(256), 0, st, g, kind, norm_pix, rows, img0, img1, pred, ldp, minmax, mask, losses, gout, vscale, (bf16_t*)dpred, ldd);
  else if (out_dtype == CSMAE_F32) hipLaunchKernelGGL((recon_bwd_kernel<float>), dim3(cdiv(rows, 4)), dim3(256), 0, st, g, kind, norm_pix, rows, img0, img1, pred, ldp, minmax, mask, losses, gout, vscale, (float*)dpred, ldd);
  else { csmae_set_error("csmae_recon_loss_bwd: bad dtype %d", out_dtype); return CSMAE_ERR_UNSUPPORTED; }
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
  for (long long r = blockIdx.x; r < rows; r += gridDim.x) {
    const float* pa = a + vrow(va, r) * D; const float* pt = t + vrow(vt, r) * D;
    for (int c = threadIdx.x; c < dv; c += blockDim.x) {
      f4_t x = *reinterpret_cast<const f4_t*>(pa + c * 4), y = *reinterpret_cast<const f4_t*>(pt + c * 4);
      for (int k = 0; k < 4; ++k) s += elem_loss(kind, x[k], y[k]);
    }
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
  if (lp_dtype == CSMAE_BF16) hipLaunchKernelGGL((pair_bwd_kernel<bf16_t>), dim3(2048), dim3(128), 0, st, kind, rows, D, a, va, t, vt, gout, coef, (bf16_t*)da_lp, da_acc, dt_acc);
  else hipLaunchKernelGGL((pair_bwd_kernel<float>), dim3(2048), dim3(128), 0, st, kind, rows, D, a, va, t, vt, gout, coef, (float*)da_lp, da_acc, dt_acc);
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
__global__ __launch_bounds__(1024) void ntx_sim_kernel(int N, int D, const float* __restrict__ z, float tau, float eps, float* __restrict__ E,
                                                      float* __restrict__ neg, float* __restrict__ rowloss) {
  __shared__ float red[32];
  extern __shared__ float zi[];
  const int i = blockIdx.x, B2 = 2 * N, partner = (i + N) % B2;
  for (int d = threadIdx.x; d < D; d += blockDim.x) zi[d] = z[(long long)i * D + d];
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float nsum = 0.f;
  for (int j0 = 4 * w; j0 < B2; j0 += 4 * (blockDim.x >> 6)) {  // four rows per wave iteration: independent load chains, one pass over z_i
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int d = lane; d < D; d += 64) {
      const float a = zi[d];
#pragma unroll
      for (int u = 0; u < 4; ++u) if (j0 + u < B2) s[u] += a * z[(long long)(j0 + u) * D + d];
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
__global__ __launch_bounds__(256) void ntx_bwd_kernel(int N, int D, const float* __restrict__ z, const float* __restrict__ inv_norm,
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
    float s = 0.f;
    for (int j = 0; j < B2; ++j) s += coef[j] * z[(long long)j * D + d];
    dpool[(long long)i * D + d] = s;
    dot += s * z[(long long)i * D + d];
  }
  dot = block_sum(dot, red);
  const float inv = inv_norm[i];
  for (int d = threadIdx.x; d < D; d += blockDim.x) dpool[(long long)i * D + d] = (dpool[(long long)i * D + d] - z[(long long)i * D + d] * dot) * inv;
}
extern "C" int csmae_ntxent_fwd(int N, int Te, int keep, int D, const float* latent, float tau, float eps, float* z, float* inv_norm,
                                float* E, float* neg, float* rowloss, void* stream) {
  CSMAE_REQUIRE(N > 0 && keep > 0 && keep < Te && D > 0 && D * 4 <= 64 * 1024, "csmae_ntxent_fwd: bad geometry N=%d Te=%d keep=%d D=%d", N, Te, keep, D);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(ntx_pool_kernel, dim3(2 * N), dim3(256), 0, st, Te, keep, D, latent, z, inv_norm);
  hipLaunchKernelGGL(ntx_sim_kernel, dim3(2 * N), dim3(1024), D * sizeof(float), st, N, D, z, tau, eps, E, neg, rowloss);
  return csmae_check_launch("csmae_ntxent_fwd");
}
extern "C" int csmae_ntxent_bwd(int N, int D, const float* z, const float* inv_norm, const float* E, const float* neg, float tau, float eps,
                                const float* gout, float* dpool, void* stream) {
  CSMAE_REQUIRE(N > 0 && D > 0 && 2 * N * 4 <= 64 * 1024, "csmae_ntxent_bwd: bad geometry");
  hipLaunchKernelGGL(ntx_bwd_kernel, dim3(2 * N), dim3(256), 2 * N * sizeof(float), (hipStream_t)stream, N, D, z, inv_norm, E, neg, tau, eps, gout, dpool);
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

// ------------------------------------------------------------------------------------------ scalar assembly
// losses[0]=total [1]=recon orig [2]=recon crop [3]=cross-decoder [4]=contrastive [5]=latent [6]=sum(mask) orig [7]=sum(mask) crop
__global__ __launch_bounds__(1024) void finalize_kernel(long long per_view, int views, const float* __restrict__ rowloss, const float* __restrict__ mask,
                                                       float recon_scale, const float* __restrict__ cd_partial, float cd_scale,
                                                       const float* __restrict__ e_partial, float e_scale, const float* __restrict__ ce_rowloss,
                                                       int ce_rows, float* __restrict__ losses) {
  __shared__ float red[32];
  float total = 0.f;
  for (int v = 0; v < 2; ++v) {
    float num = 0.f, den = 0.f;
    if (v < views) for (long long i = threadIdx.x; i < per_view; i += blockDim.x) { float m = mask[v * per_view + i]; num += rowloss[v * per_view + i] * m; den += m; }
    num = block_sum(num, red); den = block_sum(den, red);
    float lv = v < views ? num / den : 0.f;  // mask_ratio = 0 -> 0/0 = NaN, as in the reference (MAE_ViT_Shared.py:119)
    if (threadIdx.x == 0) { losses[1 + v] = lv; losses[6 + v] = den; }
    total += lv * recon_scale;
  }
  float cd = 0.f, e = 0.f, ce = 0.f;
  if (cd_partial) { float s = 0.f; for (int i = threadIdx.x; i < PAIR_BLOCKS; i += blockDim.x) s += cd_partial[i]; cd = block_sum(s, red) * cd_scale; }
  if (e_partial) { float s = 0.f; for (int i = threadIdx.x; i < PAIR_BLOCKS; i += blockDim.x) s += e_partial[i]; e = block_sum(s, red) * e_scale; }
  if (ce_rowloss) { float s = 0.f; for (int i = threadIdx.x; i < ce_rows; i += blockDim.x) s += ce_rowloss[i]; ce = block_sum(s, red) / ce_rows; }
  if (threadIdx.x == 0) { losses[3] = cd; losses[4] = ce; losses[5] = e; losses[0] = total + cd + ce + e; }
}
extern "C" int csmae_loss_finalize(long long per_view, int views, const float* rowloss, const float* mask, float recon_scale,
                                   const float* cd_partial, float cd_scale, const float* e_partial, float e_scale,
                                   const float* ce_rowloss, int ce_rows, float* losses, void* stream) {
  CSMAE_REQUIRE(per_view > 0 && (views == 1 || views == 2) && losses, "csmae_loss_finalize: bad args");
  hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, per_view, views, rowloss, mask, recon_scale, cd_partial, cd_scale, e_partial, e_scale, ce_rowloss, ce_rows, losses);
  return csmae_check_launch("csmae_loss_finalize");
}
