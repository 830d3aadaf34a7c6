// Token plumbing around the transformer stacks (all HBM/latency bound, integer + copy work):
//   crop_resize      MAE_ViT_MsLd.py:29-35,52  (one RandomResizedCrop box per batch, bilinear anti-aliased resize)
//   mask_sort        MAE_ViT_Shared.py:57-84   (random_masking: argsort x2, mask)       -- bit-exact contract
//   patch_gather     timm PatchEmbed on KEPT patches only (MAE_ViT_Baseline.py:245,251)
//   embed_assemble   + encoder_pos_embed, cls prepend (MAE_ViT_Baseline.py:248,253-256)
//   unshuffle        forward_decoder's mask-token fill + gather(ids_restore) + decoder_pos_embed (:273-283)
//   rows_gather/scatter  "[:, 1:, :]" views feeding the predictor (MAE_ViT_MsLdCeCd.py:57-58)
#include "common.h"

// ------------------------------------------------------------------------------------------ crop + AA bilinear resize
// Follows ATen's separable anti-aliased bilinear filter (triangle filter, support = max(scale, 1), weights
// renormalised), horizontal pass first, in float.  `box` = device int[4] {top i, left j, height h, width w}.
// A workgroup = CROP_ROWS output rows of one plane, a thread = one output column of them, no LDS, no barrier.  The crop of the step always
// up-samples (the box is a part of the image: support 1, at most 3 taps per axis): a thread computes its column's taps ONCE for the strip's
// rows, the rows' taps are computed by the first CROP_ROWS lanes of every wave (one row each) and handed to all lanes as scalars
// (v_readlane), and the <= 9 source pixels of an output come straight from global memory, where neighbouring outputs share them (L1 / L2).
// History: tap tables in LDS per workgroup + three barrier-separated LDS phases (two dependent global round trips per workgroup at four
// workgroups per CU): 92-110 us for 154 MB; one thread per pixel with both axes' taps per thread: 129 us (~400 instructions per pixel, the
// divisions of the weight normalisation).  Boxes larger than the output (down-sampling: up to 8 taps per axis, as ATen's kernel) take
// the per-pixel loop form: same sums in the same order.
#define CROP_ROWS 8
struct AxisRange { float center, inv; int lo, n; float tot; };
__device__ __forceinline__ float aa_w(const AxisRange& r, int k) { const float a = fabsf((k + r.lo - r.center + 0.5f) * r.inv); return a < 1.f ? 1.f - a : 0.f; }
__device__ __forceinline__ AxisRange aa_range(int o, int in_size, int out_size) {
  AxisRange r;
  const float scale = (float)in_size / (float)out_size;
  const float support = scale >= 1.f ? scale : 1.f;
  r.inv = scale >= 1.f ? 1.f / scale : 1.f;
  r.center = scale * (o + 0.5f);
  int lo = (int)(r.center - support + 0.5f); lo = lo < 0 ? 0 : lo;
  int hi = (int)(r.center + support + 0.5f); hi = hi > in_size ? in_size : hi;
  int n = hi - lo; n = n > 8 ? 8 : n;
  r.lo = lo; r.n = n;
  float tot = 0.f;
  for (int k = 0; k < n; ++k) tot += aa_w(r, k);
  r.tot = tot;
  return r;
}
__global__ __launch_bounds__(256) void crop_resize_kernel(int S, const float* __restrict__ src, float* __restrict__ dst, const int* __restrict__ box) {
  const int bi = box[0], bj = box[1], bh = box[2], bw = box[3];
  const int oy0 = blockIdx.x * CROP_ROWS, lane = threadIdx.x & 63;
  const long long pl = blockIdx.y;
  const float* plane = src + pl * S * S;
  float* out = dst + pl * S * S;
  if (bw > S || bh > S) {   // down-sampling (not the step's case): per-pixel loops
    for (int i = threadIdx.x; i < CROP_ROWS * S; i += 256) {
      const int r = i / S, ox = i - r * S, oy = oy0 + r;
      if (oy >= S) break;
      const AxisRange rx = aa_range(ox, bw, S), ry = aa_range(oy, bh, S);
      const float* base = plane + (long long)(bi + ry.lo) * S + bj + rx.lo;
      float acc = 0.f;
      for (int a = 0; a < ry.n; ++a) {
        float hsum = 0.f;
        for (int b = 0; b < rx.n; ++b) hsum += (aa_w(rx, b) / rx.tot) * base[a * S + b];
        acc += (aa_w(ry, a) / ry.tot) * hsum;
      }
      out[(long long)oy * S + ox] = acc;
    }
    return;
  }
  // rows' taps: lane r < CROP_ROWS of every wave computes those of row oy0 + r
  const AxisRange ry = aa_range(min(oy0 + min(lane, CROP_ROWS - 1), S - 1), bh, S);
  float wyl[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) wyl[k] = k < ry.n ? aa_w(ry, k) / ry.tot : 0.f;
  for (int ox = threadIdx.x; ox < S; ox += 256) {   // (one trip for S <= 256)
    const AxisRange rx = aa_range(ox, bw, S);
    float wx[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) wx[k] = k < rx.n ? aa_w(rx, k) / rx.tot : 0.f;
    const float* col = plane + (long long)bi * S + bj + rx.lo;
#pragma unroll
    for (int r = 0; r < CROP_ROWS; ++r) {
      const int oy = oy0 + r;
      if (oy >= S) break;
      const int ylo = __builtin_amdgcn_readlane(ry.lo, r), yn = __builtin_amdgcn_readlane(ry.n, r);
      const float wy0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wyl[0]), r));
      const float wy1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wyl[1]), r));
      const float wy2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wyl[2]), r));
      const float wy[3] = {wy0, wy1, wy2};
      const float* base = col + (long long)ylo * S;
      float v[3][3];
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) v[a][b] = (a < yn && b < rx.n) ? base[a * S + b] : 0.f;   // (all loads in flight before the first use)
      float acc = 0.f;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        float hsum = 0.f;
#pragma unroll
        for (int b = 0; b < 3; ++b) if (b < rx.n) hsum += wx[b] * v[a][b];
        if (a < yn) acc += wy[a] * hsum;
      }
      out[(long long)oy * S + ox] = acc;
    }
  }
}
extern "C" int csmae_crop_resize(long long planes, int S, const float* src, float* dst, const int* box, void* stream) {
  CSMAE_REQUIRE(planes > 0 && planes < 65536 && S > 0 && S <= 1024 && src && dst && box, "csmae_crop_resize: bad args");
  dim3 grid((S + CROP_ROWS - 1) / CROP_ROWS, (unsigned)planes), block(256);
  hipLaunchKernelGGL(crop_resize_kernel, grid, block, 0, (hipStream_t)stream, S, src, dst, box);
  return csmae_check_launch("csmae_crop_resize");
}

// ------------------------------------------------------------------------------------------ random_masking indices
// rank sort (stable ascending): rank(i) = #{j : noise[j] < noise[i] or (noise[j] == noise[i] and j < i)}.
// ids_shuffle[rank] = i ; ids_restore[i] = rank ; mask[i] = rank >= keep.  (ids_restore = argsort(ids_shuffle) exactly,
// because ids_shuffle is a permutation.)
__global__ __launch_bounds__(256) void mask_sort_kernel(int L, int keep, const float* __restrict__ noise, long long* __restrict__ ids_restore,
                                                        float* __restrict__ mask, int* __restrict__ ids_keep, int* __restrict__ ids_shuffle) {
  extern __shared__ float nz[];
  const long long row = blockIdx.x;
  for (int i = threadIdx.x; i < L; i += blockDim.x) nz[i] = noise[row * L + i];
  __syncthreads();
  for (int i = threadIdx.x; i < L; i += blockDim.x) {
    const float v = nz[i];
    int rank = 0;
    for (int j = 0; j < L; ++j) { float u = nz[j]; rank += (u < v) || (u == v && j < i); }
    ids_restore[row * L + i] = rank;
    mask[row * L + i] = rank >= keep ? 1.f : 0.f;
    if (rank < keep) ids_keep[row * keep + rank] = i;
    if (ids_shuffle) ids_shuffle[row * L + rank] = i;
  }
}
extern "C" int csmae_mask_sort(long long rows, int L, int keep, const float* noise, long long* ids_restore, float* mask, int* ids_keep,
                               int* ids_shuffle, void* stream) {
  CSMAE_REQUIRE(rows > 0 && L > 0 && keep >= 0 && keep <= L && L <= 16384, "csmae_mask_sort: bad geometry rows=%lld L=%d keep=%d", rows, L, keep);
  hipLaunchKernelGGL(mask_sort_kernel, dim3((unsigned)rows), dim3(256), L * sizeof(float), (hipStream_t)stream, L, keep, noise, ids_restore, mask, ids_keep, ids_shuffle);
  return csmae_check_launch("csmae_mask_sort");
}

// ------------------------------------------------------------------------------------------ kept-patch im2col
// out[row = n2*keep + t][c*p*p + ph*p + pw] = img_view(n2)[n][c][gh*p + ph][gw*p + pw], token l = ids_keep[row] = gh*G + gw.
// Column order matches Conv2d weight [D, C, p, p] flattened (MAE_ViT_Baseline.py:222-224).  Pad columns [P, ld) are zeroed.
template <typename T>
__global__ __launch_bounds__(256) void patch_gather_kernel(int keep, int N, int C, int S, int p, const float* __restrict__ img0,
                                                           const float* __restrict__ img1, const int* __restrict__ ids_keep,
                                                           T* __restrict__ out, long long ld) {
  const long long row = blockIdx.x;
  const long long n2 = row / keep;
  const int view = (int)(n2 / N), n = (int)(n2 - (long long)view * N);
  const float* img = (view ? img1 : img0) + (long long)n * C * S * S;
  const int G = S / p, l = ids_keep[row], gh = l / G, gw = l - gh * G;
  const int P = C * p * p;
  for (int e = threadIdx.x; e < ld; e += blockDim.x) {
    float v = 0.f;
    if (e < P) { int c = e / (p * p), r = e - c * p * p, ph = r / p, pw = r - ph * p; v = img[((long long)c * S + gh * p + ph) * S + gw * p + pw]; }
    st_from_f32<T>(out + row * ld + e, v);
  }
}
extern "C" int csmae_patch_gather(int dtype, long long rows, int keep, int N, int C, int S, int p, const float* img0, const float* img1,
                                  const int* ids_keep, void* out, long long ld, void* stream) {
  CSMAE_REQUIRE(rows > 0 && keep > 0 && N > 0 && S % p == 0 && ld >= (long long)C * p * p, "csmae_patch_gather: bad geometry");
  CSMAE_REQUIRE(rows <= (long long)N * keep || img1, "csmae_patch_gather: second view requested but img1 is null");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CSMAE_BF16) hipLaunchKernelGGL((patch_gather_kernel<bf16_t>), dim3((unsigned)rows), dim3(256), 0, st, keep, N, C, S, p, img0, img1, ids_keep, (bf16_t*)out, ld);
  else if (dtype == CSMAE_F32) hipLaunchKernelGGL((patch_gather_kernel<float>), dim3((unsigned)rows), dim3(256), 0, st, keep, N, C, S, p, img0, img1, ids_keep, (float*)out, ld);
  else { csmae_set_error("csmae_patch_gather: bad dtype %d", dtype); return CSMAE_ERR_UNSUPPORTED; }
  return csmae_check_launch("csmae_patch_gather");
}

// ------------------------------------------------------------------------------------------ pos-embed add + cls prepend
template <typename TX>   // TX: type of the residual stream (fp32; bf16 in throughput mode)
__global__ __launch_bounds__(256) void embed_assemble_kernel(int keep, int D, const float* __restrict__ tok, const float* __restrict__ pos,
                                                             const float* __restrict__ cls, const int* __restrict__ ids_keep, TX* __restrict__ x) {
  const long long n2 = blockIdx.x;
  const int t = blockIdx.y;  // 0 = cls
  const int dv = D >> 2;
  TX* dst = x + (n2 * (keep + 1) + t) * D;
  const float* a = cls;
  const float* pp = pos;
  if (t != 0) {
    const long long r = n2 * keep + t - 1;
    a = tok + r * D;
    pp = pos + (long long)(1 + ids_keep[r]) * D;
  }
  for (int c = threadIdx.x; c < dv; c += blockDim.x)
    st4<TX>(dst + c * 4, *reinterpret_cast<const f4_t*>(a + c * 4) + *reinterpret_cast<const f4_t*>(pp + c * 4));
}
extern "C" int csmae_embed_assemble(int x_dtype, long long B2, int keep, int D, const float* tok, const float* pos, const float* cls, const int* ids_keep, void* x,
                                    void* stream) {
  CSMAE_REQUIRE(B2 > 0 && keep >= 0 && D % 4 == 0, "csmae_embed_assemble: bad geometry");
  const dim3 grid((unsigned)B2, keep + 1), block(D >= 1024 ? 256 : 128);
  if (x_dtype == CSMAE_F32) hipLaunchKernelGGL(embed_assemble_kernel<float>, grid, block, 0, (hipStream_t)stream, keep, D, tok, pos, cls, ids_keep, (float*)x);
  else if (x_dtype == CSMAE_BF16) hipLaunchKernelGGL(embed_assemble_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, keep, D, tok, pos, cls, ids_keep, (bf16_t*)x);
  else { csmae_set_error("csmae_embed_assemble: bad dtype %d", x_dtype); return CSMAE_ERR_UNSUPPORTED; }
  return csmae_check_launch("csmae_embed_assemble");
}
// backward: dtok[n2*keep + t] = dx[n2, 1+t] (cast) ; dcls += sum_n2 dx[n2, 0]
template <typename TX, typename T>
__global__ __launch_bounds__(256) void embed_assemble_bwd_kernel(long long B2, int keep, int D, const TX* __restrict__ dx, T* __restrict__ dtok, float* __restrict__ dcls) {
  const int dv = D >> 2;
  if (blockIdx.y == 0) {  // cls column sums, deterministic: workgroup b owns columns [4b, 4b + 4) (float4 units); its 128 threads are
    // 4 columns x 32 sample groups, folded through LDS in a fixed order (a serial walk over all 2N samples by D/4 threads of one
    // workgroup used to be this kernel's whole duration)
    __shared__ f4_t part[32][4];
    const int cg = threadIdx.x & 3, sg = threadIdx.x >> 2;
    for (int c0 = blockIdx.x * 4; c0 < dv; c0 += gridDim.x * 4) {
      const int c = c0 + cg;
      f4_t s = {0.f, 0.f, 0.f, 0.f};
      if (c < dv) for (long long n = sg; n < B2; n += 32) s += ld4<TX>(dx + n * (keep + 1) * D + c * 4);
      __syncthreads();
      part[sg][cg] = s;
      __syncthreads();
      if (sg == 0 && c < dv) {
        f4_t o = *reinterpret_cast<f4_t*>(dcls + c * 4);
        for (int k = 0; k < 32; ++k) o += part[k][cg];
        *reinterpret_cast<f4_t*>(dcls + c * 4) = o;
      }
    }
    return;
  }
  const int t = blockIdx.y - 1;
  for (long long n2 = blockIdx.x; n2 < B2; n2 += gridDim.x)
    for (int c = threadIdx.x; c < dv; c += blockDim.x)
      st4<T>(dtok + (n2 * keep + t) * D + c * 4, ld4<TX>(dx + (n2 * (keep + 1) + 1 + t) * D + c * 4));
}
extern "C" int csmae_embed_assemble_bwd(int in_dtype, int dtype, long long B2, int keep, int D, const void* dx, void* dtok, float* dcls, void* stream) {
  CSMAE_REQUIRE(B2 > 0 && keep >= 0 && D % 4 == 0, "csmae_embed_assemble_bwd: bad geometry");
  dim3 grid((unsigned)fmin((double)B2, 512.0), keep + 1), block(128);
  hipStream_t st = (hipStream_t)stream;
  if (in_dtype == CSMAE_BF16 && dtype == CSMAE_BF16) hipLaunchKernelGGL((embed_assemble_bwd_kernel<bf16_t, bf16_t>), grid, block, 0, st, B2, keep, D, (const bf16_t*)dx, (bf16_t*)dtok, dcls);
  else if (in_dtype == CSMAE_F32 && dtype == CSMAE_BF16) hipLaunchKernelGGL((embed_assemble_bwd_kernel<float, bf16_t>), grid, block, 0, st, B2, keep, D, (const float*)dx, (bf16_t*)dtok, dcls);
  else if (in_dtype == CSMAE_F32 && dtype == CSMAE_F32) hipLaunchKernelGGL((embed_assemble_bwd_kernel<float, float>), grid, block, 0, st, B2, keep, D, (const float*)dx, (float*)dtok, dcls);
  else { csmae_set_error("csmae_embed_assemble_bwd: bad dtype %d", dtype); return CSMAE_ERR_UNSUPPORTED; }
  return csmae_check_launch("csmae_embed_assemble_bwd");
}

// ------------------------------------------------------------------------------------------ decoder unshuffle
// xd[n,0] = z[n,0] + dpos[0];  xd[n,1+j] = (r = ids_restore[n,j]) < keep ? z[n,1+r] : mask_token ;  + dpos[1+j]
template <typename TX>
__global__ __launch_bounds__(128) void unshuffle_fwd_kernel(int L, int keep, int Dd, const float* __restrict__ z, const float* __restrict__ mask_token,
                                                            const float* __restrict__ dpos, const long long* __restrict__ ids_restore, TX* __restrict__ xd) {
  const long long n = blockIdx.x;
  const int j = blockIdx.y;  // 0 = cls
  const int dv = Dd >> 2;
  const float* src;
  if (j == 0) src = z + n * (keep + 1) * Dd;
  else { long long r = ids_restore[n * L + j - 1]; src = r < keep ? z + (n * (keep + 1) + 1 + r) * Dd : mask_token; }
  TX* dst = xd + (n * (L + 1) + j) * Dd;
  const float* pp = dpos + (long long)j * Dd;
  for (int c = threadIdx.x; c < dv; c += blockDim.x)
    st4<TX>(dst + c * 4, *reinterpret_cast<const f4_t*>(src + c * 4) + *reinterpret_cast<const f4_t*>(pp + c * 4));
}
extern "C" int csmae_unshuffle_fwd(int x_dtype, long long B2, int L, int keep, int Dd, const float* z, const float* mask_token, const float* dpos,
                                   const long long* ids_restore, void* xd, void* stream) {
  CSMAE_REQUIRE(B2 > 0 && L > 0 && keep >= 0 && keep <= L && Dd % 4 == 0, "csmae_unshuffle_fwd: bad geometry");
  const dim3 grid((unsigned)B2, L + 1);
  if (x_dtype == CSMAE_F32) hipLaunchKernelGGL(unshuffle_fwd_kernel<float>, grid, dim3(128), 0, (hipStream_t)stream, L, keep, Dd, z, mask_token, dpos, ids_restore, (float*)xd);
  else if (x_dtype == CSMAE_BF16) hipLaunchKernelGGL(unshuffle_fwd_kernel<bf16_t>, grid, dim3(128), 0, (hipStream_t)stream, L, keep, Dd, z, mask_token, dpos, ids_restore, (bf16_t*)xd);
  else { csmae_set_error("csmae_unshuffle_fwd: bad dtype %d", x_dtype); return CSMAE_ERR_UNSUPPORTED; }
  return csmae_check_launch("csmae_unshuffle_fwd");
}
// backward: kept tokens are routed back (unique writers: ids_restore is a permutation), masked positions sum into dmask_token
template <typename TX, typename T>
__global__ __launch_bounds__(1024) void unshuffle_bwd_kernel(int L, int keep, int Dd, const TX* __restrict__ dxd, const long long* __restrict__ ids_restore,
                                                            T* __restrict__ dz, float* __restrict__ dmask_token) {
  // one workgroup per sample; its threads are (column unit, row group): row group g walks rows j = g, g + RG, ... (a single walk over
  // all L rows by D/4 threads left the kernel latency-bound), the mask-token partials are folded through LDS in a fixed order
  constexpr int RG = 8;   // (four row groups = 8 waves per CU at one workgroup per sample: 1.5 TB/s; eight: twice the loads in flight)
  __shared__ f4_t part[RG][128];
  const long long n = blockIdx.x;
  const int dv = Dd >> 2, cl = threadIdx.x & 127, g = threadIdx.x >> 7;
  for (int c0 = 0; c0 < dv; c0 += 128) {
    const int c = c0 + cl;
    f4_t acc = {0.f, 0.f, 0.f, 0.f};
    if (c < dv) {
      if (g == 0) st4<T>(dz + n * (keep + 1) * Dd + c * 4, ld4<TX>(dxd + n * (L + 1) * Dd + c * 4));
      for (int j = g; j < L; j += RG) {
        long long r = ids_restore[n * L + j];
        f4_t gr = ld4<TX>(dxd + (n * (L + 1) + 1 + j) * Dd + c * 4);
        if (r < keep) st4<T>(dz + (n * (keep + 1) + 1 + r) * Dd + c * 4, gr); else acc += gr;
      }
    }
    __syncthreads();
    part[g][cl] = acc;
    __syncthreads();
    if (g == 0 && c < dv) {
      f4_t s = part[0][cl];
      for (int k = 1; k < RG; ++k) s += part[k][cl];
      for (int k = 0; k < 4; ++k) unsafeAtomicAdd(dmask_token + c * 4 + k, s[k]);
    }
  }
}
extern "C" int csmae_unshuffle_bwd(int in_dtype, int dtype, long long B2, int L, int keep, int Dd, const void* dxd, const long long* ids_restore, void* dz,
                                   float* dmask_token, void* stream) {
  CSMAE_REQUIRE(B2 > 0 && L > 0 && keep >= 0 && keep <= L && Dd % 4 == 0, "csmae_unshuffle_bwd: bad geometry");
  hipStream_t st = (hipStream_t)stream;
  if (in_dtype == CSMAE_BF16 && dtype == CSMAE_BF16) hipLaunchKernelGGL((unshuffle_bwd_kernel<bf16_t, bf16_t>), dim3((unsigned)B2), dim3(1024), 0, st, L, keep, Dd, (const bf16_t*)dxd, ids_restore, (bf16_t*)dz, dmask_token);
  else if (in_dtype == CSMAE_F32 && dtype == CSMAE_BF16) hipLaunchKernelGGL((unshuffle_bwd_kernel<float, bf16_t>), dim3((unsigned)B2), dim3(1024), 0, st, L, keep, Dd, (const float*)dxd, ids_restore, (bf16_t*)dz, dmask_token);
  else if (in_dtype == CSMAE_F32 && dtype == CSMAE_F32) hipLaunchKernelGGL((unshuffle_bwd_kernel<float, float>), dim3((unsigned)B2), dim3(1024), 0, st, L, keep, Dd, (const float*)dxd, ids_restore, (float*)dz, dmask_token);
  else { csmae_set_error("csmae_unshuffle_bwd: bad dtype %d", dtype); return CSMAE_ERR_UNSUPPORTED; }
  return csmae_check_launch("csmae_unshuffle_bwd");
}

// ------------------------------------------------------------------------------------------ strided row views
// view row r -> storage row (r / group) * gstride + off + r % group
template <typename T>
__global__ __launch_bounds__(128) void rows_gather_kernel(long long rows, int D, const float* __restrict__ src, long long group, long long gstride, long long off, T* __restrict__ dst) {
  const int dv = D >> 2;
  for (long long r = blockIdx.x; r < rows; r += gridDim.x) {
    const float* s = src + ((r / group) * gstride + off + r % group) * D;
    for (int c = threadIdx.x; c < dv; c += blockDim.x) st4<T>(dst + r * D + c * 4, *reinterpret_cast<const f4_t*>(s + c * 4));
  }
}
template <typename T>
__global__ __launch_bounds__(128) void rows_scatter_add_kernel(long long rows, int D, const T* __restrict__ src, float scale, long long group, long long gstride, long long off, float* __restrict__ dst) {
  const int dv = D >> 2;
  for (long long r = blockIdx.x; r < rows; r += gridDim.x) {
    float* d = dst + ((r / group) * gstride + off + r % group) * D;
    for (int c = threadIdx.x; c < dv; c += blockDim.x) {
      f4_t o = *reinterpret_cast<f4_t*>(d + c * 4) + ld4<T>(src + r * D + c * 4) * scale;
      *reinterpret_cast<f4_t*>(d + c * 4) = o;
    }
  }
}
// dst[view(r, off_a)] += scale_a * a[r] and dst[view(r, off_b)] += scale_b * b[r] in one pass (the cross-decoder head's two contributions to the
// decoder-embedding gradient: - d/dv into the original's rows, the predictor's input gradient into the crop's — MAE_ViT_MsLdCeCd.py:56-59)
template <typename T>
__global__ __launch_bounds__(128) void rows_scatter_add2_kernel(long long rows, int D, const T* __restrict__ a, float sa, long long off_a, const T* __restrict__ b, float sb,
                                                                long long off_b, long long group, long long gstride, float* __restrict__ dst) {
  const int dv = D >> 2;
  for (long long r = blockIdx.x; r < rows; r += gridDim.x) {
    const long long base = (r / group) * gstride + r % group;
    float* da = dst + (base + off_a) * D;
    float* db = dst + (base + off_b) * D;
    for (int c = threadIdx.x; c < dv; c += blockDim.x) {
      const f4_t va = ld4<T>(a + r * D + c * 4), vb = ld4<T>(b + r * D + c * 4);
      const f4_t oa = *reinterpret_cast<f4_t*>(da + c * 4) + va * sa, ob = *reinterpret_cast<f4_t*>(db + c * 4) + vb * sb;
      *reinterpret_cast<f4_t*>(da + c * 4) = oa;
      *reinterpret_cast<f4_t*>(db + c * 4) = ob;
    }
  }
}
extern "C" int csmae_rows_scatter_add2(int dtype, long long rows, int D, const void* a, float scale_a, long long off_a, const void* b, float scale_b, long long off_b,
                                       long long group, long long gstride, float* dst, void* stream) {
  CSMAE_REQUIRE(rows > 0 && D % 4 == 0 && group > 0 && off_a != off_b, "csmae_rows_scatter_add2: bad geometry");
  dim3 grid((unsigned)fmin((double)rows, 8192.0)), block(128);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CSMAE_BF16) hipLaunchKernelGGL((rows_scatter_add2_kernel<bf16_t>), grid, block, 0, st, rows, D, (const bf16_t*)a, scale_a, off_a, (const bf16_t*)b, scale_b, off_b, group, gstride, dst);
  else if (dtype == CSMAE_F32) hipLaunchKernelGGL((rows_scatter_add2_kernel<float>), grid, block, 0, st, rows, D, (const float*)a, scale_a, off_a, (const float*)b, scale_b, off_b, group, gstride, dst);
  else { csmae_set_error("csmae_rows_scatter_add2: bad dtype %d", dtype); return CSMAE_ERR_UNSUPPORTED; }
  return csmae_check_launch("csmae_rows_scatter_add2");
}
// ---- the cross-decoder head's backward, run AHEAD of loss.backward() (csmae_hip/engine.py: the predictor chain pair loss -> Linear -> BatchNorm / ReLU ->
// Linear is the longest of the forward / backward junction and depends on the upstream gradient only through a scalar factor: every gradient is
// linear in it).  The chain runs at the end of the forward pass with unit upstream gradient; when the real one, g, arrives this kernel (a) adds
// g x (BatchNorm's dgamma, dbeta of the unit run) into the gradient buffer and (b) scales the chain's three bf16 gradient tensors by g in place — a
// no-op per block when g == 1 (every `loss.backward()` without gradient accumulation).  MAE_ViT_MsLdCeCd.py:56-59, MLP.py:4-10 backward.
__global__ __launch_bounds__(256) void spec_fixup_kernel(const float* __restrict__ g, bf16_t* b0, long long n0, bf16_t* b1, long long n1, bf16_t* b2, long long n2,
                                                         const float* __restrict__ tg, const float* __restrict__ tb, float* __restrict__ dg, float* __restrict__ db, int L) {
  const float s = g[0];
  if (blockIdx.x == 0)
    for (int i = threadIdx.x; i < L; i += blockDim.x) { dg[i] += s * tg[i]; db[i] += s * tb[i]; }
  if (s == 1.0f) return;
  bf16_t* bufs[3] = {b0, b1, b2};
  const long long cnt[3] = {n0, n1, n2};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    uint4* p = reinterpret_cast<uint4*>(bufs[k]);
    const long long n8 = cnt[k] >> 3;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
      uint4 v = p[i];
      unsigned* w = reinterpret_cast<unsigned*>(&v);
#pragma unroll
      for (int e = 0; e < 4; ++e) w[e] = pack2bf(__uint_as_float(w[e] << 16) * s, __uint_as_float(w[e] & 0xffff0000u) * s);
      p[i] = v;
    }
  }
}
extern "C" int csmae_spec_fixup(const float* g, void* b0, long long n0, void* b1, long long n1, void* b2, long long n2, const float* tmp_dgamma, const float* tmp_dbeta,
                                float* dgamma, float* dbeta, int L, void* stream) {
  CSMAE_REQUIRE(g && b0 && b1 && b2 && tmp_dgamma && tmp_dbeta && dgamma && dbeta && L > 0 && n0 % 8 == 0 && n1 % 8 == 0 && n2 % 8 == 0 &&
                ((((uintptr_t)b0 | (uintptr_t)b1 | (uintptr_t)b2) & 15) == 0), "csmae_spec_fixup: bf16 tensors of 8-element multiples, 16-byte aligned");
  hipLaunchKernelGGL(spec_fixup_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, g, (bf16_t*)b0, n0, (bf16_t*)b1, n1, (bf16_t*)b2, n2, tmp_dgamma, tmp_dbeta, dgamma, dbeta, L);
  return csmae_check_launch("csmae_spec_fixup");
}
extern "C" int csmae_rows_gather(int dtype, long long rows, int D, const float* src, long long group, long long gstride, long long off, void* dst, void* stream) {
  CSMAE_REQUIRE(rows > 0 && D % 4 == 0 && group > 0, "csmae_rows_gather: bad geometry");
  dim3 grid((unsigned)fmin((double)rows, 4096.0)), block(128);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CSMAE_BF16) hipLaunchKernelGGL((rows_gather_kernel<bf16_t>), grid, block, 0, st, rows, D, src, group, gstride, off, (bf16_t*)dst);
  else if (dtype == CSMAE_F32) hipLaunchKernelGGL((rows_gather_kernel<float>), grid, block, 0, st, rows, D, src, group, gstride, off, (float*)dst);
  else { csmae_set_error("csmae_rows_gather: bad dtype %d", dtype); return CSMAE_ERR_UNSUPPORTED; }
  return csmae_check_launch("csmae_rows_gather");
}
extern "C" int csmae_rows_scatter_add(int dtype, long long rows, int D, const void* src, float scale, long long group, long long gstride, long long off, float* dst, void* stream) {
  CSMAE_REQUIRE(rows > 0 && D % 4 == 0 && group > 0, "csmae_rows_scatter_add: bad geometry");
  dim3 grid((unsigned)fmin((double)rows, 4096.0)), block(128);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CSMAE_BF16) hipLaunchKernelGGL((rows_scatter_add_kernel<bf16_t>), grid, block, 0, st, rows, D, (const bf16_t*)src, scale, group, gstride, off, dst);
  else if (dtype == CSMAE_F32) hipLaunchKernelGGL((rows_scatter_add_kernel<float>), grid, block, 0, st, rows, D, (const float*)src, scale, group, gstride, off, dst);
  else { csmae_set_error("csmae_rows_scatter_add: bad dtype %d", dtype); return CSMAE_ERR_UNSUPPORTED; }
  return csmae_check_launch("csmae_rows_scatter_add");
}

// out[n, k, :] = x[n, ids[n, k], :]  — the row gather of the stand-alone `random_masking` (MAE_ViT_Shared.py:77: torch.gather over ids_keep)
__global__ __launch_bounds__(128) void rows_gather_idx_kernel(long long rows, int keep, int L, int D, const float* __restrict__ x, const int* __restrict__ ids,
                                                              long long ids_ld, float* __restrict__ out) {
  for (long long r = blockIdx.x; r < rows; r += gridDim.x) {
    const long long n = r / keep; const int k = (int)(r - n * keep);
    const float* s = x + (n * L + ids[n * ids_ld + k]) * (long long)D;
    float* d = out + r * (long long)D;
    if ((D & 3) == 0) { for (int c = threadIdx.x; c < (D >> 2); c += blockDim.x) *reinterpret_cast<f4_t*>(d + c * 4) = *reinterpret_cast<const f4_t*>(s + c * 4); }
    else for (int c = threadIdx.x; c < D; c += blockDim.x) d[c] = s[c];
  }
}
extern "C" int csmae_rows_gather_idx(long long N, int L, int keep, int D, const float* x, const int* ids, long long ids_ld, float* out, void* stream) {
  CSMAE_REQUIRE(N > 0 && L > 0 && keep > 0 && keep <= L && D > 0 && ids_ld >= keep, "csmae_rows_gather_idx: bad geometry N=%lld L=%d keep=%d D=%d", N, L, keep, D);
  const long long rows = N * keep;
  hipLaunchKernelGGL(rows_gather_idx_kernel, dim3((unsigned)fmin((double)rows, 8192.0)), dim3(128), 0, (hipStream_t)stream, rows, keep, L, D, x, ids, ids_ld, out);
  return csmae_check_launch("csmae_rows_gather_idx");
}

// ------------------------------------------------------------------------------------------ input step
// The reference's training transform (util/datasets.py:120-136) on the GPU, one kernel per batch of decoded uint8 images:
//   ToTensor (u8 / 255) -> Normalize(mean, std) -> RandomHorizontalFlip -> RandomVerticalFlip ->
//   RandomResizedCrop(S, scale (0.25, 1), bicubic, antialias)
// The random decisions are drawn on the host (torchvision's RNG order) and arrive as `meta[n] = {H, W, i, j, h, w, hflip, vflip}`;
// `src` is [N, Hmax, Wmax, C] uint8 (HWC, each image in the top-left H x W corner), `dst` [N, C, S, S] fp32.
// Resampling follows ATen's separable anti-aliased bicubic (a = -0.5; support = 2 * max(scale, 1); weights renormalised).
__device__ __forceinline__ float cubic_aa(float x) {
  x = fabsf(x);
  const float a = -0.5f;
  if (x < 1.f) return ((a + 2.f) * x - (a + 3.f)) * x * x + 1.f;
  if (x < 2.f) return (((x - 5.f) * x + 8.f) * x - 4.f) * a;
  return 0.f;
}
#define AUG_MAX_TAPS 96
#define AUG_MAX_C 8
#define AUG_WX_TAPS 24   // horizontal weights kept in LDS (24 KiB: several workgroups per CU); taps beyond are recomputed (scale > 5.5)
// One workgroup per (image, output row).  The normalised vertical weights live in LDS, each thread (output column) keeps its
// normalised horizontal weights in LDS too (column-major: conflict-free), so the tap loop is one byte load + convert + fma per
// channel.  Normalisation is applied once at the end: sum_k w_k (v_k / 255 - mean) / std = (sum_k w_k v_k) / (255 std) - mean / std
// because the weights sum to one.
template <int CT>  // CT = channel count known at compile time (3, 4) or 0 = any C <= 8
__global__ __launch_bounds__(256) void augment_u8_kernel(int Crt, int Hmax, int Wmax, int S, const unsigned char* __restrict__ src,
                                                         const int* __restrict__ meta, const float* __restrict__ mean,
                                                         const float* __restrict__ inv_std, float* __restrict__ dst) {
  __shared__ float wy[AUG_MAX_TAPS];
  __shared__ float wx[AUG_WX_TAPS][256];
  const int C = CT ? CT : Crt;
  constexpr int NC = CT ? CT : AUG_MAX_C;
  __shared__ int ylo_s, yn_s;
  const long long n = blockIdx.x;
  const int oy = blockIdx.y;
  const int* mt = meta + n * 8;
  const int H = mt[0], W = mt[1], bi = mt[2], bj = mt[3], bh = mt[4], bw = mt[5], hf = mt[6], vf = mt[7];
  const float sy = (float)bh / (float)S, sx = (float)bw / (float)S;
  const float supy = sy >= 1.f ? 2.f * sy : 2.f, invy = sy >= 1.f ? 1.f / sy : 1.f;
  const float supx = sx >= 1.f ? 2.f * sx : 2.f, invx = sx >= 1.f ? 1.f / sx : 1.f;
  if (threadIdx.x == 0) {
    const float cy = sy * (oy + 0.5f);
    int lo = (int)(cy - supy + 0.5f); lo = lo < 0 ? 0 : lo;
    int hi = (int)(cy + supy + 0.5f); hi = hi > bh ? bh : hi;
    int nn = hi - lo; nn = nn > AUG_MAX_TAPS ? AUG_MAX_TAPS : nn;
    float tot = 0.f;
    for (int k = 0; k < nn; ++k) { const float w = cubic_aa((k + lo - cy + 0.5f) * invy); wy[k] = w; tot += w; }
    for (int k = 0; k < nn; ++k) wy[k] /= tot;
    ylo_s = lo; yn_s = nn;
  }
  __syncthreads();
  const int ylo = ylo_s, yn = yn_s;
  const unsigned char* img = src + n * (long long)Hmax * Wmax * C;
  const int tid = threadIdx.x;
  for (int ox = tid; ox < S; ox += blockDim.x) {
    const float cx = sx * (ox + 0.5f);
    int xlo = (int)(cx - supx + 0.5f); xlo = xlo < 0 ? 0 : xlo;
    int xhi = (int)(cx + supx + 0.5f); xhi = xhi > bw ? bw : xhi;
    int xn = xhi - xlo; xn = xn > AUG_MAX_TAPS ? AUG_MAX_TAPS : xn;
    float totx = 0.f;
    for (int b = 0; b < xn; ++b) { const float w = cubic_aa((b + xlo - cx + 0.5f) * invx); if (b < AUG_WX_TAPS) wx[b][tid] = w; totx += w; }
    const float rtx = 1.f / totx;
    float acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = 0.f;
    // source column of tap b in the stored (un-flipped) image, stepping by +-C bytes
    const int x0 = hf ? W - 1 - (bj + xlo) : bj + xlo;
    const int xstep = hf ? -C : C;
    for (int a = 0; a < yn; ++a) {
      int yy = bi + ylo + a; yy = vf ? H - 1 - yy : yy;          // row of the flipped image -> row of the stored image
      const unsigned char* px = img + ((long long)yy * Wmax + x0) * C;
      const float wa = wy[a] * rtx;
      for (int b = 0; b < xn; ++b, px += xstep) {
        const float w = wa * (b < AUG_WX_TAPS ? wx[b][tid] : cubic_aa((b + xlo - cx + 0.5f) * invx));
#pragma unroll
        for (int c = 0; c < NC; ++c) if (c < C) acc[c] = fmaf(w, (float)px[c], acc[c]);
      }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c)
      if (c < C) dst[((n * C + c) * S + oy) * (long long)S + ox] = acc[c] * (inv_std[c] * (1.f / 255.f)) - mean[c] * inv_std[c];
  }
}
extern "C" int csmae_augment_u8(long long N, int C, int Hmax, int Wmax, int S, const unsigned char* src, const int* meta, const float* mean,
                                const float* inv_std, float* dst, void* stream) {
  CSMAE_REQUIRE(N > 0 && C > 0 && C <= AUG_MAX_C && Hmax > 0 && Wmax > 0 && S > 0 && src && meta && mean && inv_std && dst, "csmae_augment_u8: bad arguments (C <= 8)");
  dim3 grid((unsigned)N, (unsigned)S);
  hipStream_t st = (hipStream_t)stream;
  if (C == 3) hipLaunchKernelGGL((augment_u8_kernel<3>), grid, dim3(256), 0, st, C, Hmax, Wmax, S, src, meta, mean, inv_std, dst);
  else if (C == 4) hipLaunchKernelGGL((augment_u8_kernel<4>), grid, dim3(256), 0, st, C, Hmax, Wmax, S, src, meta, mean, inv_std, dst);
  else hipLaunchKernelGGL((augment_u8_kernel<0>), grid, dim3(256), 0, st, C, Hmax, Wmax, S, src, meta, mean, inv_std, dst);
  return csmae_check_launch("csmae_augment_u8");
}
