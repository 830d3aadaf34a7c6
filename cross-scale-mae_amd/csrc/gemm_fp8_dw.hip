// fp8 weight gradients (BASELINE.json configs[4]; round 6): dW[M = out][N = in] += dq_y dq_x sum_k dY8[k][m] X8[k][n] over the tokens k, both operands
// K-STRIDED OCP fp8 bytes as their producers left them ([tokens][features]: dY e5m2 — the gradient copies the dX products already read —, X e4m3 — the
// activation copies the forward products read), fp32 accumulation in v_mfma_scale_f32_16x16x128_f8f6f4.  The grouped launch, the K slices, the ordered fold
// and the direct accumulate of csmae_gemm_dw_group (gemm.hip) on the fp8 kernel's pipelined ring (gemm_fp8_pipe_kernel): a K step is 128 tokens = one
// [128 k-rows][256 columns] byte image per operand (32 KiB, the ring's unit), read with ds_read_b64_tr_b8 (tools/tr8_probe.hip: within a 16-lane group
// lane t supplies the address of 8 bytes — k-row t / 2, half t % 2 of an [8 k-rows][16 columns] block — and receives column t's 8 k-values): four reads
// per 16 x 128 fragment (k = 32 g + 8 r .. + 7).  Image rows are 256 B = one pass over all 64 banks, so the 16-byte granule a k-row's columns 16 c .. 16 c + 15
// live in is swizzled: physical granule = c ^ key(k-row), key = (k-row & 7) | (((k-row >> 5) & 1) << 3): the 16 k-rows a half-wave's read touches
// ({8 r .. 8 r + 7} + 32 g, g = 0, 1 or 2, 3) fall on 16 distinct granules.  Per wave and K step: 48 reads + 8 DMA pieces around 32 MFMAs of 8 passes —
// the bf16 weight-gradient loop's instruction count for twice the tokens.
#include "gemm_common.h"

typedef int i8v_t __attribute__((ext_vector_type(8)));
struct Dw8Desc { const void* dY; const void* X; float* dW; float* db; const float* dq_y; const float* dq_x; int M, N; long long ldy, ldx; int tiles_n, tile0; };
struct Dw8GroupArgs {
  Dw8Desc d[DW_GROUP_MAX];
  int n, K, ktiles, ktiles_per_split, nsplit, total_tiles;
  float* slab; float* cs_slab;
};

__device__ __forceinline__ void fp8_dw_tile(char* smem, const Dw8Desc& d, const int K, const int tm, const int tn, const int split, const int kt_begin, const int kt_end,
                                            float* slab, const int nsplit, const int slot, const int nslots) {
  constexpr int WM = 128, WN = 64, NWN = 4, NW = 8, FM = 8, FN = 4, UNIT = 128 * 256, NUNIT = 5, PP = UNIT / 1024 / NW;   // 4 pieces per wave and image
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int t = lane & 15, g = lane >> 4;
  const int m0 = tm * 256, n0 = tn * 256;
  const i4_t rsA = make_rsrc(d.dY, (unsigned)((long long)K * d.ldy)), rsB = make_rsrc(d.X, (unsigned)((long long)K * d.ldx));
  // DMA: piece = 4 k-rows x 256 B; lane -> k-row rho = 16 w + 4 q + (lane >> 4), physical granule p = lane & 15 holds logical granule p ^ key(rho);
  // key(rho) = ((4 (q & 1) + (lane >> 4)) & 7) | (((w >> 1) & 1) << 3): even / odd pieces differ.  k-rows beyond K are beyond the resource: zeros.
  unsigned avo[2], bvo[2];
  {
    const int kr = lane >> 4, p = lane & 15;
#pragma unroll
    for (int qo = 0; qo < 2; ++qo) {
      const int key = ((4 * qo + kr) & 7) | (((w >> 1) & 1) << 3);
      avo[qo] = (unsigned)((long long)(16 * w + kr) * d.ldy + m0 + ((p ^ key) << 4));
      bvo[qo] = (unsigned)((long long)(16 * w + kr) * d.ldx + n0 + ((p ^ key) << 4));
    }
  }
  const unsigned aqs = (unsigned)(4 * d.ldy), bqs = (unsigned)(4 * d.ldx), ajs = (unsigned)(128 * d.ldy), bjs = (unsigned)(128 * d.ldx);
  const unsigned lds0 = (unsigned)(size_t)LDS_PTR(char, smem);
  const unsigned ldsW = lds0 + (unsigned)(w * PP) * 1024u;
  auto dma_a = [&](int slot_, int j, int q) { lds_dma16u(rsA, avo[q & 1] + ((unsigned)j * ajs + (unsigned)q * aqs), ldsW + (unsigned)(slot_ * UNIT + q * 1024)); };
  auto dma_b = [&](int slot_, int j, int q) { lds_dma16u(rsB, bvo[q & 1] + ((unsigned)j * bjs + (unsigned)q * bqs), ldsW + (unsigned)(slot_ * UNIT + q * 1024)); };
  f4_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f4_t{0.f, 0.f, 0.f, 0.f};
  const int wm = (w / NWN) * WM, wn = (w % NWN) * WN;
  // fragment read r of fragment i: k-row 32 g + 8 r + (t >> 1), granule ((wm >> 4) + i) ^ kx, kx = (t >> 1) | ((g & 1) << 3), the 8 bytes (t & 1): the fragment
  // index only XORs bits 4 .. 6 of the offset; read r is 2 KiB further (an immediate)
  const int kx = (t >> 1) | ((g & 1) << 3);
  const int ra0 = (32 * g + (t >> 1)) * 256 + ((((wm >> 4)) ^ kx) << 4) + (t & 1) * 8;
  const int rb0 = (32 * g + (t >> 1)) * 256 + ((((wn >> 4)) ^ kx) << 4) + (t & 1) * 8;
  uint2 fa[FM][4], fb[FN][4];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) fa[i][r] = make_uint2(0u, 0u);
#pragma unroll
  for (int j = 0; j < FN; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) fb[j][r] = make_uint2(0u, 0u);
  auto read_frag = [&](int slot_, int base, int i, uint2 (&f)[4]) {
    const unsigned a = lds0 + (unsigned)(slot_ * UNIT) + (unsigned)(base ^ (i << 4));
    asm volatile("ds_read_b64_tr_b8 %0, %1" : "+v"(f[0]) : "v"(a));
    asm volatile("ds_read_b64_tr_b8 %0, %1 offset:2048" : "+v"(f[1]) : "v"(a));
    asm volatile("ds_read_b64_tr_b8 %0, %1 offset:4096" : "+v"(f[2]) : "v"(a));
    asm volatile("ds_read_b64_tr_b8 %0, %1 offset:6144" : "+v"(f[3]) : "v"(a));
  };
  auto read_a = [&](int slot_, auto ic) { constexpr int i = decltype(ic)::value; read_frag(slot_, ra0, i, fa[i]); };
  auto read_b = [&](int slot_, auto jc) { constexpr int j = decltype(jc)::value; read_frag(slot_, rb0, j, fb[j]); };
  auto join8 = [](const uint2 (&f)[4]) { return i8v_t{(int)f[0].x, (int)f[0].y, (int)f[1].x, (int)f[1].y, (int)f[2].x, (int)f[2].y, (int)f[3].x, (int)f[3].y}; };
  // (the bias gradient — column sums of dY — is NOT made here: beside 224 accumulator / fragment registers the conversions of a fragment's bytes spill, and
  // a run-time wave condition around them splits the straight-line K step into basic blocks (700 spilled registers): fp8_colsum_* below, 1 B per element)
  const int nsteps = kt_end - kt_begin, U = 2 * nsteps;
  const int issued0 = min(NUNIT - 1, U - 1);
#pragma unroll
  for (int u = 0; u < NUNIT; ++u)
    if (u <= issued0) {
#pragma unroll
      for (int q = 0; q < PP; ++q) { if (u & 1) dma_b(u, kt_begin + (u >> 1), q); else dma_a(u, kt_begin + (u >> 1), q); }
    }
  if (issued0 >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PP) : "memory");
  else if (issued0 == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PP) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  static_for<FN>([&](auto jc) { read_b(1, jc); });
  static_for<FM>([&](auto ic) { read_a(0, ic); });
  auto nxt = [](int s, int k) { s += k; return s >= NUNIT ? s - NUNIT : s; };
  // dW = dY^T X: the MFMA's "A" is dY (e5m2: blgp = 1), its "B" is X (e4m3: cbsz = 0); operands swapped so that a lane ends with 4 consecutive output columns
#define FP8_MMA(i, j) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(join8(fb[j]), join8(fa[i]), acc[i][j], 0, 1, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f)
  // the step of gemm_fp8_pipe_kernel: every read of a step is consumed in the NEXT step behind the lgkmcnt(0) at its head; A fragment i is re-read in place
  // behind row i, the B fragments behind the last two rows (column by column); MODE 0 fetches units 2j+5 and 2j+6, 1 only 2j+5, 2 nothing, 3 = last step
  auto step = [&](int j, int sl, auto mode_tag) {
    constexpr int MODE = decltype(mode_tag)::value;
    constexpr bool more = MODE < 3;
    if (MODE <= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PP) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    const int sb_cur = nxt(sl, 1), sa = nxt(sl, 2), sb = nxt(sl, 3);
    static_for<FM - 2>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
#pragma unroll
      for (int jj = 0; jj < FN; ++jj) FP8_MMA(i, jj);
      if (more) read_a(sa, ic);
      if (i < PP) { if (MODE <= 1) dma_b(sl, kt_begin + j + 2, i); }
      else if (MODE == 0) dma_a(sb_cur, kt_begin + j + 3, i - PP);
      __builtin_amdgcn_sched_barrier(0);
    });
    static_for<FN>([&](auto jc) {
      constexpr int jj = decltype(jc)::value;
      FP8_MMA(FM - 2, jj); FP8_MMA(FM - 1, jj);
      if (more) read_b(sb, jc);
      if (MODE == 0 && jj + (FM - 2 - PP) < PP) dma_a(sb_cur, kt_begin + j + 3, jj + (FM - 2 - PP));
      __builtin_amdgcn_sched_barrier(0);
    });
    if (more) { read_a(sa, std::integral_constant<int, FM - 2>{}); read_a(sa, std::integral_constant<int, FM - 1>{}); }
    __builtin_amdgcn_sched_barrier(0);
  };
  int j = 0, sl = 0;
  for (; j < nsteps - 3; ++j, sl = nxt(sl, 2)) step(j, sl, std::integral_constant<int, 0>{});
  if (nsteps >= 3) { step(j, sl, std::integral_constant<int, 1>{}); ++j; sl = nxt(sl, 2); }
  if (nsteps >= 2) { step(j, sl, std::integral_constant<int, 2>{}); ++j; sl = nxt(sl, 2); }
  step(j, sl, std::integral_constant<int, 3>{});
#undef FP8_MMA
  const float alpha = (d.dq_y ? d.dq_y[0] : 1.f) * (d.dq_x ? d.dq_x[0] : 1.f);
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int jj = 0; jj < FN; ++jj) acc[i][jj] *= alpha;
  // ---- epilogue of a grouped weight-gradient tile (k64_tile<.., GROUP>): one slice -> dW += acc, db += sums; several -> dense fp32 slab + partials for the ordered fold
  constexpr int ESTR = WN + 4, LPR = WN / 4, RPP = 64 / LPR, EROWS = 32;
  float* ew = reinterpret_cast<float*>(smem) + w * (EROWS * ESTR);
  GemmArgs p;
  p.A = nullptr; p.B = nullptr; p.C = d.dW; p.bias = nullptr; p.aux = nullptr; p.resid = d.dW; p.lda = p.ldb = 0; p.ldc = d.N; p.ldaux = 0; p.ldr = d.N;
  p.M = d.M; p.N = d.N; p.K = K; p.c_dtype = CSMAE_F32; p.epi = EPI_RESID; p.splitk = nsplit; p.tiles_m = 0; p.tiles_n = d.tiles_n; p.ktiles = 0; p.ktiles_per_split = 0;
  p.a_bytes = p.b_bytes = 0; p.force_cfg = 0; p.split_stride = 0; p.colsum = d.db; p.dq_a = p.dq_b = nullptr; p.a_fmt = 0; p.aux_q8 = 0; p.q_out = nullptr;
  __syncthreads();   // the ring becomes the epilogue strip: every wave is past its last fragment reads
  if (nsplit > 1) {
    GemmArgs q = p;
    q.M = 256; q.N = 256; q.ldc = 256;
    float* mine = slab + ((long long)split * nslots + slot) * (256 * 256);
    epilogue_rows<float, EPI_NONE, FM, FN, WM, EROWS, ESTR, LPR, RPP>(q, mine, acc, ew, wm, wn, lane, t, g);
    return;
  }
  epilogue_rows<float, EPI_RESID, FM, FN, WM, EROWS, ESTR, LPR, RPP>(p, d.dW, acc, ew, m0 + wm, n0 + wn, lane, t, g);   // dW = dW + acc
}

__global__ __launch_bounds__(512, 1) void gemm_fp8_dw_group_kernel(Dw8GroupArgs ga) {
  const int wg = xcd_remap(blockIdx.x, ga.total_tiles * ga.nsplit);   // slice-major: the workgroups of one K slice are neighbours on an XCD
  const int split = wg / ga.total_tiles, tile_id = wg - split * ga.total_tiles;
  Dw8Desc d = ga.d[0];
#pragma unroll
  for (int i = 1; i < DW_GROUP_MAX; ++i) if (i < ga.n && tile_id >= ga.d[i].tile0) d = ga.d[i];
  const int tile = tile_id - d.tile0;
  const int tm = tile / d.tiles_n, tn = tile - tm * d.tiles_n;
  const int kt_begin = split * ga.ktiles_per_split;
  const int kt_end = min(kt_begin + ga.ktiles_per_split, ga.ktiles);
  __shared__ __attribute__((aligned(16))) char smem[5 * 128 * 256];
fp8_dw_tile(smem, d, ga.K, tm, tn, split, kt_begin, kt_end, ga.slab, ga.nsplit, tile_id, ga.total_tiles);
}
// ordered fold of the K slices (dw_group_reduce_kernel of gemm.hip on this file's descriptor table)
__global__ __launch_bounds__(256) void fp8_dw_group_reduce_kernel(Dw8GroupArgs ga) {
  const int tile_id = blockIdx.x / DWR_PARTS, quarter = blockIdx.x % DWR_PARTS;
  Dw8Desc d = ga.d[0];
#pragma unroll
  for (int i = 1; i < DW_GROUP_MAX; ++i) if (i < ga.n && tile_id >= ga.d[i].tile0) d = ga.d[i];
  const int tile = tile_id - d.tile0;
  const int tm = tile / d.tiles_n, tn = tile - tm * d.tiles_n;
  const int m0 = tm * 256, n0 = tn * 256;
  const int c4 = threadIdx.x & 63, r0 = threadIdx.x >> 6;
  const long long sstride = (long long)ga.total_tiles * 256 * 256;
  const float* base = ga.slab + (long long)tile_id * 256 * 256;
  const int n = n0 + c4 * 4;
  constexpr int RPB = 256 / DWR_PARTS;
#pragma unroll 1
  for (int r = quarter * RPB + r0; r < quarter * RPB + RPB; r += 4) {
    const int m = m0 + r;
    if (m >= d.M || n >= d.N) continue;
    float* dst = d.dW + (long long)m * d.N + n;
    f4_t a = *reinterpret_cast<f4_t*>(dst);
    for (int sl = 0; sl < ga.nsplit; ++sl) a += *reinterpret_cast<const f4_t*>(base + sl * sstride + r * 256 + c4 * 4);
    *reinterpret_cast<f4_t*>(dst) = a;
  }
}

// ---- bias gradients: db[m] += dq_y sum_k dY8[k][m] (e5m2 bytes).  Two launches per group, fixed summation order: partial sums of CS_RB row blocks per
// 256-column block (thread = 16-byte column group x row lane; 16 row lanes folded through LDS in lane order), then the row blocks folded in order.
#define CS_RB 64
__global__ __launch_bounds__(256) void fp8_colsum_partial_kernel(Dw8GroupArgs ga, float* __restrict__ part /* [n][CS_RB][pitch] */, int pitch) {
  const Dw8Desc d = ga.d[blockIdx.z];
  if (d.db == nullptr || (int)blockIdx.x * 256 >= d.M) return;
  __shared__ float red[16][256 + 4];
  const int cg = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int col = blockIdx.x * 256 + cg * 16;
  const long long rows_per = ((long long)ga.K + CS_RB - 1) / CS_RB, r0 = (long long)blockIdx.y * rows_per, r1 = r0 + rows_per < ga.K ? r0 + rows_per : ga.K;
  float s[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) s[e] = 0.f;
  if (col < d.ldy) {   // (rows are 16-byte multiples: a group is inside the row or beyond it; columns >= M inside the row hold padding and are dropped by the fold)
    const unsigned char* src = reinterpret_cast<const unsigned char*>(d.dY) + col;
    auto add16 = [&](const uint4& v) {
      const unsigned wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const auto lo = __builtin_amdgcn_cvt_pk_f32_bf8((int)wv[q], false), hi = __builtin_amdgcn_cvt_pk_f32_bf8((int)wv[q], true);
        s[4 * q] += lo[0]; s[4 * q + 1] += lo[1]; s[4 * q + 2] += hi[0]; s[4 * q + 3] += hi[1];
      }
    };
    long long r = r0 + rl;
    for (; r + 48 < r1; r += 64) {   // four independent 16-byte loads in flight per thread (one per iteration left the kernel at a third of the HBM rate)
      const uint4 v0 = *reinterpret_cast<const uint4*>(src + r * d.ldy), v1 = *reinterpret_cast<const uint4*>(src + (r + 16) * d.ldy);
      const uint4 v2 = *reinterpret_cast<const uint4*>(src + (r + 32) * d.ldy), v3 = *reinterpret_cast<const uint4*>(src + (r + 48) * d.ldy);
      add16(v0); add16(v1); add16(v2); add16(v3);
    }
    for (; r < r1; r += 16) add16(*reinterpret_cast<const uint4*>(src + r * d.ldy));
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) red[rl][cg * 16 + e] = s[e];
  __syncthreads();
  float a = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) a += red[k][threadIdx.x];
  part[((long long)blockIdx.z * CS_RB + blockIdx.y) * pitch + blockIdx.x * 256 + threadIdx.x] = a;
}
__global__ __launch_bounds__(256) void fp8_colsum_fold_kernel(Dw8GroupArgs ga, const float* __restrict__ part, int pitch) {
  const Dw8Desc d = ga.d[blockIdx.z];
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (d.db == nullptr || m >= d.M) return;
  const float* src = part + (long long)blockIdx.z * CS_RB * pitch + m;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;   // (four independent chains: 64 dependent loads were 26 us for a 5-block launch; the order of the sum is fixed either way)
#pragma unroll 4
  for (int rb = 0; rb < CS_RB; rb += 4) {
    a0 += src[(long long)rb * pitch]; a1 += src[(long long)(rb + 1) * pitch]; a2 += src[(long long)(rb + 2) * pitch]; a3 += src[(long long)(rb + 3) * pitch];
  }
  d.db[m] += ((a0 + a1) + (a2 + a3)) * (d.dq_y ? d.dq_y[0] : 1.f);
}

// The weight gradients of several nn.Linear layers over the same K tokens in ONE launch, fp8 operands: product i is dW[i] (fp32 [M][N], contiguous) +=
// dq_y[i] dq_x[i] dY8[i]^T X8[i], db[i] (nullable) += dq_y[i] colsum(dY8[i]); dY8 e5m2 / X8 e4m3 bytes [K][ld], dq_* device scalars (the de-quantisation
// factors their producers left).  Geometry, K slices, workspace and fold as csmae_gemm_dw_group; M, N >= 256, N % 4 == 0, ld % 16 == 0.
extern "C" int csmae_gemm_dw_group_fp8(int count, long long K, const void* const* dY, const long long* ldy, const float* const* dq_y, const void* const* X,
                                       const long long* ldx, const float* const* dq_x, float* const* dW, float* const* db, const long long* M, const long long* N,
                                       int slots, float* workspace, long long ws_elems, void* stream) {
  CSMAE_REQUIRE(count > 0 && count <= DW_GROUP_MAX && K > 0 && dY && X && dW && M && N && ldy && ldx && dq_y && dq_x && workspace, "csmae_gemm_dw_group_fp8: bad arguments (1..%d products)", DW_GROUP_MAX);
  Dw8GroupArgs ga;
  int tiles = 0;
  for (int i = 0; i < count; ++i) {
    CSMAE_REQUIRE(M[i] >= 256 && N[i] >= 256 && N[i] % 4 == 0 && ldy[i] % 16 == 0 && ldx[i] % 16 == 0 && ldy[i] >= (M[i] + 15) / 16 * 16 && ldx[i] >= (N[i] + 15) / 16 * 16 &&
                  (K + 128) * ldy[i] < 0xFFFFFFF0ll && (K + 128) * ldx[i] < 0xFFFFFFF0ll && ((((uintptr_t)dY[i] | (uintptr_t)X[i] | (uintptr_t)dW[i]) & 15) == 0),
                  "csmae_gemm_dw_group_fp8: product %d: need M, N >= 256, N %% 4 == 0, rows of 16-byte multiples covering the rounded-up widths, 16-byte aligned operands", i);
    Dw8Desc& d = ga.d[i];
    d.dY = dY[i]; d.X = X[i]; d.dW = dW[i]; d.db = db ? db[i] : nullptr; d.dq_y = dq_y[i]; d.dq_x = dq_x[i]; d.M = (int)M[i]; d.N = (int)N[i]; d.ldy = ldy[i]; d.ldx = ldx[i];
    d.tiles_n = cdiv(N[i], 256); d.tile0 = tiles;
    tiles += cdiv(M[i], 256) * d.tiles_n;
  }
  for (int i = count; i < DW_GROUP_MAX; ++i) ga.d[i] = ga.d[0];
  ga.n = count; ga.K = (int)K; ga.total_tiles = tiles;
  ga.ktiles = cdiv(K, 128);
  if (slots <= 0) slots = 128;
  long long S = slots / tiles;
  if (S > ga.ktiles / 4) S = ga.ktiles / 4;          // (a slice is at least 4 K steps of 128 tokens)
  const long long per_slice = (long long)tiles * (256 * 256);
  if (S > ws_elems / per_slice) S = ws_elems / per_slice;
  if (S < 1) S = 1;
  ga.ktiles_per_split = cdiv(ga.ktiles, S);
  ga.nsplit = cdiv(ga.ktiles, ga.ktiles_per_split);
  ga.slab = workspace; ga.cs_slab = nullptr;
  CSMAE_REQUIRE(ga.nsplit == 1 || ws_elems >= ga.nsplit * per_slice, "csmae_gemm_dw_group_fp8: workspace too small");
  hipLaunchKernelGGL(gemm_fp8_dw_group_kernel, dim3(tiles * ga.nsplit), dim3(512), 0, (hipStream_t)stream, ga);
  if (ga.nsplit > 1) hipLaunchKernelGGL(fp8_dw_group_reduce_kernel, dim3(tiles * DWR_PARTS), dim3(256), 0, (hipStream_t)stream, ga);
  if (db) {   // bias gradients (column sums of dY8): partial sums behind the slabs in the workspace
    int maxM = 0; bool any = false;
    for (int i = 0; i < count; ++i) if (db[i]) { any = true; if (M[i] > maxM) maxM = (int)M[i]; }
    if (any) {
      const int pitch = cdiv(maxM, 256) * 256;
      float* part = workspace + (long long)ga.nsplit * per_slice;
      CSMAE_REQUIRE(ws_elems >= (long long)ga.nsplit * per_slice + (long long)count * CS_RB * pitch, "csmae_gemm_dw_group_fp8: workspace too small for the bias-gradient partial sums");
      hipLaunchKernelGGL(fp8_colsum_partial_kernel, dim3(pitch / 256, CS_RB, count), dim3(256), 0, (hipStream_t)stream, ga, part, pitch);
      hipLaunchKernelGGL(fp8_colsum_fold_kernel, dim3(pitch / 256, 1, count), dim3(256), 0, (hipStream_t)stream, ga, part, pitch);
    }
  }
  return csmae_check_launch("csmae_gemm_dw_group_fp8");
}
