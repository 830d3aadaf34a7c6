#include "gemm_common.h"

template <bool TA, bool TB, int BM, int BN, int WM, int WN, int GEMM_STAGES, bool DB, int MINW = 1>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64, MINW) void gemm_bf16_kernel(GemmArgs p) {
  constexpr int NWN = BN / WN, NW = (BM / WM) * NWN, FM = WM / 16, FN = WN / 16;
  constexpr int A_BYTES = BM * GEMM_BK * 2, B_BYTES = BN * GEMM_BK * 2, STAGE = A_BYTES + B_BYTES;
  constexpr int PA = A_BYTES / 1024, PB = B_BYTES / 1024, PPW = (PA + PB) / NW;
  static_assert((PA + PB) % NW == 0, "DMA pieces must divide evenly over the waves");
  __shared__ __attribute__((aligned(16))) char smem[GEMM_STAGES * STAGE];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int t = lane & 15, g = lane >> 4;
  const int tiles = p.tiles_m * p.tiles_n;
  const int wg = xcd_remap(blockIdx.x, tiles * p.splitk);
  const int split = wg / tiles, tile = wg - split * tiles;
  const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int kt_begin = split * p.ktiles_per_split;
  const int kt_end = min(kt_begin + p.ktiles_per_split, p.ktiles);
  if (kt_begin >= kt_end) return;  // only possible for surplus split-K slices
  if (p.epi == EPI_SPLIT) p.C = reinterpret_cast<float*>(p.C) + (long long)split * p.split_stride;

  const i4_t rsA = make_rsrc(p.A, p.a_bytes), rsB = make_rsrc(p.B, p.b_bytes);
  const unsigned kstepA = TA ? (unsigned)(GEMM_BK * p.lda * 2) : (unsigned)(GEMM_BK * 2);
  const unsigned kstepB = TB ? (unsigned)(GEMM_BK * p.ldb * 2) : (unsigned)(GEMM_BK * 2);

  unsigned poff[PPW]; int pk[PPW]; bool pok[PPW];
#pragma unroll
  for (int q = 0; q < PPW; ++q) {
    const int pi = w * PPW + q;
    if (pi < PA) { PieceDesc<TA, BM> d; d.init(pi, lane, m0, p.M, p.lda); poff[q] = d.off; pk[q] = d.kidx; pok[q] = d.ok; }
    else { PieceDesc<TB, BN> d; d.init(pi - PA, lane, n0, p.N, p.ldb); poff[q] = d.off; pk[q] = d.kidx; pok[q] = d.ok; }
  }
  auto stage = [&](int kt) {
    char* base = smem + ((kt - kt_begin) % GEMM_STAGES) * STAGE;
    const int k0 = kt * GEMM_BK;
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
      const int pi = w * PPW + q;
      const bool ok = pok[q] && (k0 + pk[q] < p.K);
      if (pi < PA) {
        unsigned o = ok ? poff[q] + (unsigned)kt * kstepA : OOB_OFF;
        lds_dma16(rsA, o, base + pi * 1024);
      } else {
        unsigned o = ok ? poff[q] + (unsigned)kt * kstepB : OOB_OFF;
        lds_dma16(rsB, o, base + A_BYTES + (pi - PA) * 1024);
      }
    }
  };

  f4_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f4_t{0.f, 0.f, 0.f, 0.f};
  const int wm = (w / NWN) * WM, wn = (w % NWN) * WN;
  // bias gradient for free: on the tn == 0 column of tiles the waves owning wn == 0 run one extra MFMA per A fragment against
  // an all-ones operand, i.e. sum_k A(m,k), instead of a separate pass that re-reads dY from HBM
  const bool do_cs = TA && TB && p.colsum != nullptr && tn == 0 && wn == 0;
  f4_t accb[FM];
#pragma unroll
  for (int i = 0; i < FM; ++i) accb[i] = f4_t{0.f, 0.f, 0.f, 0.f};
  const s8_t ones = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};

  // per-lane fragment offsets inside an operand image
  int ra[FM], rb[FN];
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    if (!TA) { int row = wm + i * 16 + t; ra[i] = row * 64 + ((g ^ ((row >> 1) & 3)) << 4); }
    else { int q = (wm >> 4) + i, key = (t >> 2) | ((g & 1) << 2); ra[i] = ((q ^ key) << 5) + (t & 3) * 8 + (8 * g + (t >> 2)) * (BM * 2); }
  }
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    if (!TB) { int row = wn + j * 16 + t; rb[j] = row * 64 + ((g ^ ((row >> 1) & 3)) << 4); }
    else { int q = (wn >> 4) + j, key = (t >> 2) | ((g & 1) << 2); rb[j] = ((q ^ key) << 5) + (t & 3) * 8 + (8 * g + (t >> 2)) * (BN * 2); }
  }

  // ---- main loop.  Fragments are double-buffered in registers: while the MFMAs of K-tile kt run on one register set, the LDS
  // reads of K-tile kt+1 (made visible by the barrier at the top of the iteration) land in the other, so neither the barrier
  // skew nor the ds_read latency sits in front of the matrix pipe.  The stage whose fragments were consumed into registers is
  // refilled right after the barrier: GEMM_STAGES - 1 K-tiles of buffer_load..lds stay in flight (counted vmcnt, raw s_barrier).
  auto read_frags = [&](int kt, s8_t (&fa)[FM], s8_t (&fb)[FN]) {
    const char* sa = smem + ((kt - kt_begin) % GEMM_STAGES) * STAGE;
    const char* sb = sa + A_BYTES;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      if (!TA) fa[i] = *reinterpret_cast<const s8_t*>(sa + ra[i]);
      else {
        s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s4_t, sa + ra[i]));
        s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s4_t, sa + ra[i] + 4 * BM * 2));
        fa[i] = join_s4(lo, hi);
      }
    }
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      if (!TB) fb[j] = *reinterpret_cast<const s8_t*>(sb + rb[j]);
      else {
        s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s4_t, sb + rb[j]));
        s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s4_t, sb + rb[j] + 4 * BN * 2));
        fb[j] = join_s4(lo, hi);
      }
    }
  };
  auto mma = [&](const s8_t (&fa)[FM], const s8_t (&fb)[FN]) {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8_t, fb[j]), __builtin_bit_cast(bf8_t, fa[i]), acc[i][j], 0, 0, 0);
    if (do_cs) {
#pragma unroll
      for (int i = 0; i < FM; ++i)
        accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8_t, ones), __builtin_bit_cast(bf8_t, fa[i]), accb[i], 0, 0, 0);
    }
  };
  // make K-tile kt visible: wait for this wave's pieces of it (`later` younger tiles may stay in flight), then meet the other waves
  auto arrive = [&](int kt, int issued_last) {
    const int later = issued_last - kt;
    if (later >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PPW) : "memory");
    else if (later == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
    else if (later == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // our own earlier fragment reads have left the stage that is about to be refilled
    __builtin_amdgcn_s_barrier();
  };
  const int last = kt_end - 1;
  const int dbg = p.force_cfg;  // tuning aid (csmae_gemm_force_tile): 16 main loop only, 32 no DMA, 64 no MFMA, 128 no fragment reads
  if (DB) {
    // Ping-pong schedule.  The two waves that share a SIMD (w and w + NW/2: a workgroup's waves are dealt to the SIMDs cyclically)
    // run half an iteration apart: while one group issues its DMA pieces and pulls fragments out of LDS ("R" phase), the other
    // group owns the matrix pipe ("M" phase).  Two barriers per K-tile keep the groups in that lock step; group 1 enters the
    // loop one barrier late and leaves it one barrier early, so every wave executes the same number of s_barriers.
    //   interval 2k  : group 0 R(k)   | group 1 M(k-1)          interval 2k+1 : group 0 M(k) | group 1 R(k)
    // R(k) = refill the ring slot of tile k-1 (read by both groups before interval 2k), read tile k, then retire this wave's
    // own pieces of tile k+1 so that they are visible to the other group one barrier before anybody reads them.
    const int grp = w / (NW / 2);
    int issued = min(kt_begin + GEMM_STAGES - 2, last);
    for (int kt = kt_begin; kt <= issued; ++kt) stage(kt);
    auto retire = [&](int later) {
      if (later >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
      else if (later == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    retire(issued - kt_begin);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (grp == 1) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
    s8_t fa0[FM] = {}, fb0[FN] = {};
    for (int kt = kt_begin; kt < kt_end; ++kt) {
      if (issued < last) { ++issued; if (!(dbg & 32)) stage(issued); }
      if (!(dbg & 128)) read_frags(kt, fa0, fb0);
      if (kt < last) retire(issued - (kt + 1));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
      if (!(dbg & 64)) mma(fa0, fb0);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      if (!(grp == 1 && kt == last)) __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
    // single register set (the 128x64 wave tile leaves no room for a second one): fragments are read after the barrier
    int issued = min(kt_begin + GEMM_STAGES - 2, last);
    for (int kt = kt_begin; kt <= issued; ++kt) stage(kt);
    s8_t fa0[FM] = {}, fb0[FN] = {};
    for (int kt = kt_begin; kt < kt_end; ++kt) {
      arrive(kt, issued);
      if (issued < last) { ++issued; if (!(dbg & 32)) stage(issued); }  // into the stage read during iteration kt-1 (every wave is past it: barrier above)
      if (!(dbg & 128)) read_frags(kt, fa0, fb0);
      if (!(dbg & 64)) mma(fa0, fb0);
    }
  }
  if (do_cs && g == 0) {
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int m = m0 + wm + i * 16 + t;
      if (m < p.M) p.colsum[(long long)split * p.M + m] = accb[i][0];
    }
  }
  if ((p.force_cfg & 16) && acc[0][0][0] != 123456.0f) return;  // tuning aid: main loop only
  // lane (t,g) holds C[m = .. + t][n = .. + 4g + r]
  if (p.epi == EPI_ATOMIC) {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
        epi_dispatch(p, m0 + wm + i * 16 + t, n0 + wn + j * 16 + 4 * g, acc[i][j]);
    return;
  }
  // Output path.  A fragment-shaped store touches 16 rows x 32 B per instruction (16 partial cache lines) and is store-issue
  // bound — with K = 512..768 the C tile is as expensive as the whole K loop.  So accumulators go through a wave-private LDS
  // strip (32 rows at a time, padded rows: conflict-free b128 writes) and leave as full 128/256-B row segments; bias, GELU,
  // residual and dGELU are applied on the way out with equally coalesced aux / residual accesses.
  constexpr int ESTR = WN + 4, LPR = WN / 4, RPP = 64 / LPR;
  constexpr int EROWS = (NW * 32 * ESTR * 4 <= GEMM_STAGES * STAGE) ? 32 : 16;  // strip height that fits the staging ring
  static_assert(NW * EROWS * ESTR * 4 <= GEMM_STAGES * STAGE, "epilogue strip must fit the staging ring");
  if (!DB) __syncthreads();  // (ping-pong: every fragment read is already behind the loop's last barrier)
  void* Cptr = p.C;
  float* ew = reinterpret_cast<float*>(smem) + w * (EROWS * ESTR);
#define EPI_CALL(TC_, E_) epilogue_rows<TC_, E_, FM, FN, WM, EROWS, ESTR, LPR, RPP>(p, Cptr, acc, ew, m0 + wm, n0 + wn, lane, t, g)
  if (p.c_dtype == CSMAE_BF16) {
    if (p.epi == EPI_GELU) EPI_CALL(bf16_t, EPI_GELU); else if (p.epi == EPI_DGELU) EPI_CALL(bf16_t, EPI_DGELU);
    else if (p.epi == EPI_RESID) EPI_CALL(bf16_t, EPI_RESID); else EPI_CALL(bf16_t, EPI_NONE);
  } else {
    if (p.epi == EPI_GELU) EPI_CALL(float, EPI_GELU); else if (p.epi == EPI_DGELU) EPI_CALL(float, EPI_DGELU);
    else if (p.epi == EPI_RESID) EPI_CALL(float, EPI_RESID); else EPI_CALL(float, EPI_NONE);
  }
#undef EPI_CALL
}

// ------------------------------------------------------------------------------------ bf16 MFMA, 64-wide K steps
// Measured with tools/dma_probe.hip: `buffer_load ... lds` of 64-B row segments (a 32-wide K tile of a K-contiguous bf16
// operand = half a cache line per row) moves 28 B/clk/CU out of a warm L2, 128-B segments move 56 B/clk/CU.  At 256x256x32 the
// first figure is 1175 clk per K-tile against 1088 clk of MFMA work: the forward / dX kernels were bound by staging.  This
// variant therefore steps K by 64 so every K-contiguous row is fetched as one full line.  LDS is a ring of five 32-KiB units
// (all 160 KiB), one unit = one operand's [256 x 64] image; a K step consumes units (2j, 2j+1) while 2j+2 .. 2j+4 are in flight.
//   K-contiguous image: [256 rows][64 k], 128-B rows, 16-B chunk swizzle c ^ (row & 7) (conflict-free ds_read_b128)
//   K-strided image   : two [32 k][256] images of the 32-wide kernel back to back (ds_read_b64_tr_b16)
#ifdef GEMM_TIMING
extern "C" int csmae_debug_gemm_ts(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gemm_ts), sizeof(g_gemm_ts)); }
#endif
// main-loop ablations of the pipelined kernel, compile-time only (tools/gemm_ablate.sh builds variant libraries; never in the product build):
// -DGEMM_ABL=1 no fragment reads | 2 no DMA pieces | 4 no MFMAs | 8 no bias-gradient sums | 16 MFMA rows at raised s_setprio (measured: no gain)
#ifndef GEMM_ABL
#define GEMM_ABL 0
#endif
// Grouped weight-gradient launches (csmae_gemm_dw_group): where K slice `split` of tile `slot` parks its fp32 tile ([nsplit][nslots]
// dense 256 x 256 slabs) and its column-sum partials ([nsplit][nslots][256]) when a tile is cut into several slices.
struct DwFold { float* slab; float* cs_slab; int nsplit; int slot; int nslots; };
#define K64_LDS_BYTES (5 * 256 * 64 * 2)   // ring of five 32-KiB units: all 160 KiB

// CSV: bias-gradient sums (K-strided-A products only) 0 = decided at run time (p.colsum && tn == 0) | 1 = never | 2 = always: the grouped weight-gradient
// kernel branches ONCE per tile into instantiation 1 or 2, so that the loop of the tiles that do not sum (most) carries none of the 16 wave-uniform
// branches per K step the run-time form costs it (35 of 231 instructions per wave and step).
template <bool TA, bool TB, int BM, bool GROUP, int CSV = 0>
__device__ __forceinline__ void k64_tile(char* smem, const GemmArgs& p, const int tm, const int tn, const int split, const int kt_begin, const int kt_end, const DwFold fold) {
  static_assert(BM == 256 || (BM == 192 && !TA), "192-row tiles exist for K-contiguous A only");
  constexpr int BN = 256, WM = BM / 2, WN = 64, NWN = BN / WN, NW = 8, FM = WM / 16, FN = WN / 16;
  constexpr int UNIT = 256 * 64 * 2, NUNIT = 5, PPU = UNIT / 1024 / NW;  // ring slot = 32 KiB; a B image fills it, a 192-row A image uses 24 KiB
  constexpr int PPA = BM * 64 * 2 / 1024 / NW;                            // DMA pieces per wave of an A image (4 or 3)
  static_assert(NUNIT * UNIT == K64_LDS_BYTES, "LDS of the pipelined kernels");
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int t = lane & 15, g = lane >> 4;
  const int m0 = tm * BM, n0 = tn * BN;
  void* Cptr = (!GROUP && p.epi == EPI_SPLIT) ? static_cast<void*>(reinterpret_cast<float*>(p.C) + (long long)split * p.split_stride) : p.C;

  GTS(0);
  const i4_t rsA = make_rsrc(p.A, p.a_bytes), rsB = make_rsrc(p.B, p.b_bytes);
  const unsigned kstepA = TA ? (unsigned)(64 * p.lda * 2) : 128u;
  const unsigned kstepB = TB ? (unsigned)(64 * p.ldb * 2) : 128u;

  // DMA descriptors.  Every piece of a wave has the same lane pattern, so the per-lane state is two byte offsets per operand
  // (even / odd piece: they differ only in the K-strided image's swizzle); the piece's rows and the K step are uniform adds.
  //   K-contiguous image: piece = 8 rows x 128 B : lane -> row (lane >> 3), physical 16-B chunk (lane & 7), logical chunk ^ (row & 7)
  //   K-strided image   : piece = 2 k-rows x 512 B: k-row = 8w + 2q + (lane >> 5), 32-B granule swizzled by tr_key(k-row)
  const int l3 = lane >> 3, l5 = lane >> 5, s16 = lane & 31;
  const int kc = ((lane & 7) ^ l3) * 8;
  auto tr_ch = [&](int qodd) { const int key = l5 | (qodd << 1) | ((w & 1) << 2); return (((s16 >> 1) ^ key) << 1) | (s16 & 1); };
  unsigned avo[2], bvo[2];
  if (!TA) avo[0] = avo[1] = (unsigned)(((long long)(m0 + w * (PPA * 8) + l3) * p.lda + kc) * 2);
  else {
    avo[0] = (unsigned)(((long long)(w * 8 + l5) * p.lda + m0 + tr_ch(0) * 8) * 2);
    avo[1] = (unsigned)(((long long)(w * 8 + l5) * p.lda + m0 + tr_ch(1) * 8) * 2);
  }
  if (!TB) bvo[0] = bvo[1] = (unsigned)(((long long)(n0 + w * 32 + l3) * p.ldb + kc) * 2);
  else {
    bvo[0] = (unsigned)(((long long)(w * 8 + l5) * p.ldb + n0 + tr_ch(0) * 8) * 2);
    bvo[1] = (unsigned)(((long long)(w * 8 + l5) * p.ldb + n0 + tr_ch(1) * 8) * 2);
  }
  const unsigned aqs = (unsigned)((TA ? 2 : 8) * p.lda * 2), bqs = (unsigned)((TB ? 2 : 8) * p.ldb * 2);  // advance per piece
  const int dbg = p.force_cfg;  // tuning aid (csmae_gemm_force_tile): 16 = main loop only
  // piece q of this wave's share of the A / B image of K step j (absolute), into ring slot `slot`
  // No predicate per piece: the operands' buffer resources end with their last row (gemm_core), so rows beyond M / N and — K-strided
  // images — k-rows beyond K are out of range and come back as zeros; K-contiguous operands are only routed here with K % 64 == 0 (a
  // chunk beyond K would be the start of the next row, which is in range).  A piece is then ONE vector add (lane offset + the uniform
  // K-step / piece advance) and three scalar instructions in front of its buffer_load: the loop is bound by instruction issue.
  const unsigned ldsA = (unsigned)(size_t)LDS_PTR(char, smem) + (unsigned)(w * PPA) * 1024u, ldsB = (unsigned)(size_t)LDS_PTR(char, smem) + (unsigned)(w * PPU) * 1024u;
  auto dma_a = [&](int slot, int j, int q) {
    if (GEMM_ABL & 2) return;
    lds_dma16u(rsA, avo[q & 1] + ((unsigned)j * kstepA + (unsigned)q * aqs), ldsA + (unsigned)(slot * UNIT + q * 1024));
  };
  auto dma_b = [&](int slot, int j, int q) {
    if (GEMM_ABL & 2) return;
    lds_dma16u(rsB, bvo[q & 1] + ((unsigned)j * kstepB + (unsigned)q * bqs), ldsB + (unsigned)(slot * UNIT + q * 1024));
  };

  f4_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f4_t{0.f, 0.f, 0.f, 0.f};
  const int wm = (w / NWN) * WM, wn = (w % NWN) * WN;
  // Fragment offsets inside a unit.  K-contiguous image: fragment i of K half h at `(base + i * 2048) ^ (h << 6)`.
  // K-strided image: fragment i sits at granule (q0 + i) ^ key of its k-rows; bits 5..7 of the offset hold only that XOR, so
  // fragment i is `base ^ (i << 5)`; K half h is 32 k-rows further.
  const int ra0 = !TA ? (wm + t) * 128 + ((g ^ (t & 7)) << 4)
                      : ((((wm >> 4) ^ ((t >> 2) | ((g & 1) << 2))) << 5) + (t & 3) * 8 + (8 * g + (t >> 2)) * (BM * 2));
  const int rb0 = !TB ? (wn + t) * 128 + ((g ^ (t & 7)) << 4)
                      : ((((wn >> 4) ^ ((t >> 2) | ((g & 1) << 2))) << 5) + (t & 3) * 8 + (8 * g + (t >> 2)) * (BN * 2));
  // K-contiguous-A kernels issue their fragment reads from inline assembly and count lgkmcnt by hand:
  //  * "+v" re-reads an A fragment into the very registers the row's MFMAs just consumed — through plain C++ the compiler renames
  //    the value and carries up to 32 extra fragment VGPRs through the loop;
  //  * at the loop head the compiler cannot know how many reads the previous iteration left in flight and drains them all
  //    (lgkmcnt(0) right after this step's four B reads were issued: ~150 clk per K step); the hand count lets them fly.
  // Read order per step (steady state): B1 x BOPS | row i: A1_i | barrier | A0'_0 A0'_1 B0' x BOPS | row i >= 2: A0'_i.
  constexpr bool USE_ASM = !TA;
  constexpr int BOPS = TB ? 2 * FN : FN;                 // LDS instructions of one B fragment set
  constexpr int W0_0 = (FM - 2) + BOPS;                  // before row 0 of a step: A0'_2.. and B1 may be in flight
  constexpr int W0_I = (FM - 1) + BOPS;                  // before row i >= 2: (FM-1-i) A0' + B1 + i A1 reads are younger
  constexpr int W1 = FM - 2;                             // before the two pre-barrier rows of the second half
  static_assert(W0_I <= 15, "lgkmcnt is a 4-bit counter");
  const unsigned lds0 = (unsigned)(size_t)LDS_PTR(char, smem);
  auto read_a = [&](int slot, int h, auto ic, s8_t& fa) {
    constexpr int i = decltype(ic)::value;
    if (GEMM_ABL & 1) return;
    const char* sa = smem + slot * UNIT;
    if (!TA) {  // one address register per (slot, half); the fragment index is the instruction's immediate offset
      const unsigned addr = lds0 + (unsigned)(slot * UNIT) + (unsigned)(ra0 ^ (h << 6));
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(fa) : "v"(addr), "n"(i * 2048));
    } else {
      s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s4_t, sa + (ra0 ^ (i << 5)) + h * (32 * BM * 2)));
      s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s4_t, sa + (ra0 ^ (i << 5)) + h * (32 * BM * 2) + 4 * BM * 2));
      fa = join_s4(lo, hi);
    }
  };
  auto read_b = [&](int slot, int h, s8_t (&fb)[FN]) {
    if (GEMM_ABL & 1) return;
    const char* sb = smem + slot * UNIT;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      if (USE_ASM && !TB) {
        const unsigned addr = lds0 + (unsigned)(slot * UNIT) + (unsigned)(rb0 ^ (h << 6));
        if (j == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(fb[0]) : "v"(addr));
        else if (j == 1) asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(fb[1]) : "v"(addr));
        else if (j == 2) asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(fb[2]) : "v"(addr));
        else asm volatile("ds_read_b128 %0, %1 offset:6144" : "=v"(fb[3]) : "v"(addr));
      } else if (USE_ASM) {
        const unsigned addr = lds0 + (unsigned)(slot * UNIT) + (unsigned)((rb0 ^ (j << 5)) + h * (32 * BN * 2));
        s4_t lo, hi;
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(addr));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(hi) : "v"(addr));
        fb[j] = join_s4(lo, hi);
      } else if (!TB) fb[j] = *reinterpret_cast<const s8_t*>(sb + (rb0 ^ (h << 6)) + j * 2048);
      else {
        s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s4_t, sb + (rb0 ^ (j << 5)) + h * (32 * BN * 2)));
        s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s4_t, sb + (rb0 ^ (j << 5)) + h * (32 * BN * 2) + 4 * BN * 2));
        fb[j] = join_s4(lo, hi);
      }
    }
  };
  // hand-counted wait, tied to the registers the following MFMAs read so that the compiler cannot hoist them above it
  auto wait_a = [&](auto n, s8_t& f0) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f0) : "n"(decltype(n)::value)); };
  auto wait_ab = [&](auto n, s8_t& f0, s8_t& f1, s8_t (&fb)[FN]) {
    static_assert(FN == 4, "four B fragments");
    asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(f0), "+v"(f1), "+v"(fb[0]), "+v"(fb[1]), "+v"(fb[2]), "+v"(fb[3]) : "n"(decltype(n)::value));
  };
  // bias gradient (sum_k A(m,k), weight-gradient products only) on the VALU slots the MFMAs leave free: on the tn == 0 tiles wave
  // (wm, wq) sums fragments 2wq and 2wq+1, which its three wn-neighbours hold as well.  The branch is wave-uniform and contains
  // no memory operation, so it does not disturb the counters of the pipelined loop.
  const bool do_cs = TA && TB && (CSV == 2 || (CSV == 0 && p.colsum != nullptr && tn == 0));
  const int wq = w % NWN;
  float cs[2] = {0.f, 0.f};
  auto mma_row = [&](int i, const s8_t& fa, const s8_t (&fb)[FN]) {
    if (!(GEMM_ABL & 4)) {
      if (GEMM_ABL & 16) __builtin_amdgcn_s_setprio(2);   // (experiment: MFMA rows at raised priority)
#pragma unroll
      for (int j = 0; j < FN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8_t, fb[j]), __builtin_bit_cast(bf8_t, fa), acc[i][j], 0, 0, 0);
      if (GEMM_ABL & 16) __builtin_amdgcn_s_setprio(0);
    }
    if (TA && TB && !(GEMM_ABL & 8) && do_cs && wq == (i >> 1)) {
      const u4_t d = __builtin_bit_cast(u4_t, fa);
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) { s0 += __uint_as_float(d[e] << 16); s1 += __uint_as_float(d[e] & 0xFFFF0000u); }
      cs[i & 1] += s0 + s1;
    }
  };
  // ---- software-pipelined main loop.  A K step is 16 "rows" of 4 MFMAs (8 A fragments x 2 K halves).  The LDS reads of the next
  // half and the DMA pieces of the units freed by the step's barrier are issued one per row, between the MFMAs, so that their
  // issue cost (~100 clk per `buffer_load .. lds`) and latency run under the matrix pipe instead of in front of it: an A fragment
  // is re-read in place as soon as its row has been issued, only the B fragments are double-buffered (64 fragment VGPRs in all).
  // The one barrier of the step sits after row 1 of the second half: at that point every wave has all fragments of the step in
  // registers (lgkmcnt(0)), so its two units can be refilled, and has retired its own pieces of the next step's units.
  // The body is straight-line code (a branch with memory operations would make the compiler drain lgkmcnt at every block head),
  // so the last three steps, which have fewer or no units left to fetch and nothing to prefetch, are separate instantiations.
  // Ring bookkeeping is scalar: unit u lives in slot u mod 5; `sl` is the slot of the current step's A unit.
  const int nsteps = kt_end - kt_begin, U = 2 * nsteps;
  const int issued0 = min(NUNIT - 1, U - 1);
#pragma unroll
  for (int u = 0; u < NUNIT; ++u)
    if (u <= issued0) {
#pragma unroll
      for (int q = 0; q < PPU; ++q) { if (u & 1) dma_b(u, kt_begin + (u >> 1), q); else if (q < PPA) dma_a(u, kt_begin + (u >> 1), q); }
    }
  if (issued0 >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPA + PPU) : "memory");  // units 2 (A), 3 (B), 4 (A) may stay in flight
  else if (issued0 == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPA + PPU) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  GTS(1);
  s8_t fa[FM], fb0[FN], fb1[FN];
  if (USE_ASM) {  // (asm reads carry their destination as "+v": give the registers a defined value once)
#pragma unroll
    for (int i = 0; i < FM; ++i) fa[i] = s8_t{0, 0, 0, 0, 0, 0, 0, 0};
  }
  read_a(0, 0, std::integral_constant<int, 0>{}, fa[0]); read_a(0, 0, std::integral_constant<int, 1>{}, fa[1]);  // order of the in-loop prefetch: A_0 A_1 B A_2 ...
  read_b(1, 0, fb0);
  static_for<FM - 2>([&](auto ic) { constexpr int i = decltype(ic)::value + 2; read_a(0, 0, std::integral_constant<int, i>{}, fa[i]); });
  auto nxt = [](int s, int k) { s += k; return s >= NUNIT ? s - NUNIT : s; };
  // A step fetches the units of step j+2, spread over both of its halves (all eight waves issue their pieces at the same rows, and
  // 64 pieces inside one half step run the CU's vector-memory path at ~90 % of what it moves: a piece then costs its wave ~100 clk):
  // the A image (unit 2j+4, into the slot step j-1's B image left at its barrier) one piece per second row of the FIRST half, the
  // B image (unit 2j+5, into the slot of this step's A image) after the barrier.
  // MODE 0: both   1: first step (unit 4 came with the prologue): B image only   2: nothing left to fetch   3: last step (nothing
  // to prefetch either)
  auto step = [&](int jabs, int sl, auto mode_tag) {
    constexpr int MODE = decltype(mode_tag)::value;
    constexpr bool more = MODE < 3;
    const int sl1 = nxt(sl, 1), sl2 = nxt(sl, 2), sl3 = nxt(sl, 3), sl4 = nxt(sl, 4);
    read_b(sl1, 1, fb1);
    static_for<FM>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      if (USE_ASM) {
        if (i == 0) wait_ab(std::integral_constant<int, W0_0>{}, fa[0], fa[1], fb0);
        else if (i >= 2) wait_a(std::integral_constant<int, W0_I>{}, fa[i]);
      }
      mma_row(i, fa[i], fb0);
      read_a(sl, 1, ic, fa[i]);
      if (MODE == 0 && (i & 1) && (i >> 1) < PPA) dma_a(sl4, jabs + 2, i >> 1);
      __builtin_amdgcn_sched_barrier(0);
    });
    if (USE_ASM) wait_ab(std::integral_constant<int, W1>{}, fa[0], fa[1], fb1);
    mma_row(0, fa[0], fb1);
    mma_row(1, fa[1], fb1);
    __builtin_amdgcn_sched_barrier(0);
    if (MODE <= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPA) : "memory");  // own pieces of units 2j+2, 2j+3 (2j+4, an A image, may stay in flight)
    else if (MODE == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (more) { read_a(sl2, 0, std::integral_constant<int, 0>{}, fa[0]); read_a(sl2, 0, std::integral_constant<int, 1>{}, fa[1]); read_b(sl3, 0, fb0); }
    // the PPU pieces of the B image: spread over the FM - 2 remaining rows (the last one takes what is left)
    static_for<FM - 2>([&](auto ic) {
      constexpr int i = decltype(ic)::value + 2, r = i - 2, NR = FM - 2;
      mma_row(i, fa[i], fb1);
      if (more) read_a(sl2, 0, std::integral_constant<int, i>{}, fa[i]);
      if (MODE <= 1) {
#pragma unroll
        for (int pc = r * PPU / NR; pc < (r + 1) * PPU / NR; ++pc) dma_b(sl, jabs + 2, pc);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  int j = 0, sl = 0;
  if (nsteps >= 3) { step(kt_begin, 0, std::integral_constant<int, 1>{}); j = 1; sl = 2; }
  for (; j < nsteps - 2; ++j, sl = nxt(sl, 2)) step(kt_begin + j, sl, std::integral_constant<int, 0>{});
  if (nsteps >= 2) { step(kt_begin + j, sl, std::integral_constant<int, 2>{}); ++j; sl = nxt(sl, 2); }
  step(kt_begin + j, sl, std::integral_constant<int, 3>{});
  GTS(2);
  if (TA && TB && do_cs) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      float v = cs[e];
      v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
      cs[e] = v;
      const int m = m0 + wm + (2 * wq + e) * 16 + t;
      if (!GROUP && g == 0 && m < p.M) p.colsum[(long long)split * p.M + m] = v;
    }
  }
  if ((dbg & 16) && acc[0][0][0] != 123456.0f) return;  // tuning aid: main loop only
  if (GROUP) {
    // ---- weight-gradient tile of a grouped launch.  One K slice: dW += acc, db += column sums, straight into the gradient buffer.
    // Several slices: row-major fp32 slab + column-sum partials for dw_group_reduce_kernel.  (Folding the slices inside this kernel —
    // last arriver per tile behind an agent-scope fence — was built and measured: on this 8-XCD part every release / acquire fence
    // writes back / invalidates a whole L2, which slowed the concurrent main-stream kernels by 15 %: 24.6 -> 28.4 ms per step.)
    constexpr int ESTR = WN + 4, LPR = WN / 4, RPP = 64 / LPR, EROWS = 32;
    float* ew = reinterpret_cast<float*>(smem) + w * (EROWS * ESTR);
    if (fold.nsplit > 1) {
      if (do_cs && g == 0) {
        float* csm = fold.cs_slab + ((long long)split * fold.nslots + fold.slot) * 256;
        csm[wm + (2 * wq) * 16 + t] = cs[0];
        csm[wm + (2 * wq + 1) * 16 + t] = cs[1];
      }
      GemmArgs q = p;   // the slab is a dense 256 x 256 tile: local coordinates, no edge clipping (the reduce clips)
      q.M = 256; q.N = 256; q.ldc = 256; q.bias = nullptr;
      float* mine = fold.slab + ((long long)split * fold.nslots + fold.slot) * (256 * 256);
      epilogue_rows<float, EPI_NONE, FM, FN, WM, EROWS, ESTR, LPR, RPP>(q, mine, acc, ew, wm, wn, lane, t, g);
      return;
    }
    if (do_cs && g == 0) {   // (one writer per row: the tn == 0 tile's wave (wm, wq))
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int m = m0 + wm + (2 * wq + e) * 16 + t;
        if (m < p.M) p.colsum[m] += cs[e];
      }
    }
    epilogue_rows<float, EPI_RESID, FM, FN, WM, EROWS, ESTR, LPR, RPP>(p, Cptr, acc, ew, m0 + wm, n0 + wn, lane, t, g);   // dW = dW + acc (resid = C)
    return;
  }
  if (p.epi == EPI_ATOMIC) {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int jj = 0; jj < FN; ++jj)
        epi_dispatch(p, m0 + wm + i * 16 + t, n0 + wn + jj * 16 + 4 * g, acc[i][jj]);
    return;
  }
  constexpr int ESTR = WN + 4, LPR = WN / 4, RPP = 64 / LPR, EROWS = 32;
  constexpr int STRIP = NW * EROWS * ESTR * 4;
  static_assert(STRIP <= NUNIT * UNIT, "epilogue strip must fit the staging ring");
  // (no barrier: nothing has read or written the ring since the last step's barrier)
  float* ew = reinterpret_cast<float*>(smem) + w * (EROWS * ESTR);
#define EPI_CALL(TC_, E_) epilogue_rows<TC_, E_, FM, FN, WM, EROWS, ESTR, LPR, RPP>(p, Cptr, acc, ew, m0 + wm, n0 + wn, lane, t, g)
#define EPI_CALL8(E_) epilogue_rows_bf16x8<E_, FM, FN, WM, EROWS, ESTR>(p, Cptr, acc, ew, m0 + wm, n0 + wn, lane, t, g)
  const bool wide = (p.ldc % 8 == 0) && (p.epi == EPI_NONE || (p.epi == EPI_RESID ? (p.ldr % 8 == 0 && (uintptr_t)p.resid % 16 == 0)
                                                                                   : (p.ldaux % 8 == 0 && (uintptr_t)p.aux % 16 == 0)));  // 16-byte row segments
  if (p.c_dtype == CSMAE_BF16) {
#define EPI_CALL8B(E_) epilogue_rows_bf16x8b<E_, FM, FN, WM, EROWS, ESTR>(p, acc, ew, m0 + wm, n0 + wn, lane, t, g)
    if (wide && p.epi == EPI_NONE && p.q_out == nullptr)
      epilogue_rows_bf16_plain<FM, FN, WM>(p, Cptr, acc, reinterpret_cast<char*>(ew), m0 + wm, n0 + wn, lane, t, g);
    else if (wide && (p.a_fmt & 256)) { if (p.epi == EPI_GELU) EPI_CALL8B(EPI_GELU); else if (p.epi == EPI_DGELU) EPI_CALL8B(EPI_DGELU); else EPI_CALL8B(EPI_RESID); }   // buffer addressing (bit 8 of a_fmt: set by the host when the sizes allow it)
    else if (wide) { if (p.epi == EPI_GELU) EPI_CALL8(EPI_GELU); else if (p.epi == EPI_DGELU) EPI_CALL8(EPI_DGELU); else if (p.epi == EPI_RESID) EPI_CALL8(EPI_RESID); else EPI_CALL8(EPI_NONE); }
    else if (p.epi == EPI_RESID) EPI_CALL(bf16_t, EPI_RESID);
    else if (p.epi == EPI_GELU) EPI_CALL(bf16_t, EPI_GELU); else if (p.epi == EPI_DGELU) EPI_CALL(bf16_t, EPI_DGELU);
    else EPI_CALL(bf16_t, EPI_NONE);
#undef EPI_CALL8B
  } else {
    if (p.epi == EPI_GELU) EPI_CALL(float, EPI_GELU); else if (p.epi == EPI_DGELU) EPI_CALL(float, EPI_DGELU);
    else if (p.epi == EPI_RESID) EPI_CALL(float, EPI_RESID); else EPI_CALL(float, EPI_NONE);
  }
#undef EPI_CALL
#undef EPI_CALL8
  GTS(3);
}

template <bool TA, bool TB, int BM = 256>  // BM = 192 (K-contiguous A only): 6 instead of 8 A fragments per wave, for outputs whose 256-row
__global__ __launch_bounds__(512, 1) void gemm_bf16_k64_kernel(GemmArgs p) {  // tiling leaves too many CUs idle (N = 768: 150 -> 201 tiles)
  const int tiles = p.tiles_m * p.tiles_n;
  const int wg = xcd_remap(blockIdx.x, tiles * p.splitk);
  const int split = wg / tiles, tile = wg - split * tiles;
  const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
  const int kt_begin = split * p.ktiles_per_split;
  const int kt_end = min(kt_begin + p.ktiles_per_split, p.ktiles);
  if (kt_begin >= kt_end) return;  // only possible for surplus split-K slices
  __shared__ __attribute__((aligned(16))) char smem[K64_LDS_BYTES];
  k64_tile<TA, TB, BM, false>(smem, p, tm, tn, split, kt_begin, kt_end, DwFold{});
}
// ---- grouped weight gradients: the dW products of a transformer block (qkv, proj, fc1, fc2 — same token axis K) in ONE launch.
// A product on its own cannot fill the chip without cutting K into 10 .. 60 slices (768 x 768: 9 tiles), each of which pays a 256-KiB
// fp32 slab store and a share of a separate reduce kernel; together the block's 108 (ViT-B encoder) / 48 (decoder) tiles need no or
// two slices, the fold happens inside the kernel (DwFold above) and the result is accumulated straight into the fp32 gradient.
__global__ __launch_bounds__(512, 1) void gemm_dw_group_kernel(DwGroupArgs ga) {
  // slice-major order: the workgroups of one K slice are neighbours (same XCD after the remap) and walk the same rows of dY / X
  const int wg = xcd_remap(blockIdx.x, ga.total_tiles * ga.nsplit);
  const int split = wg / ga.total_tiles, tile_id = wg - split * ga.total_tiles;
  DwDesc d = ga.d[0];   // (selected by a chain of uniform moves: a dynamic index into the by-value argument would go through scratch memory)
#pragma unroll
  for (int i = 1; i < DW_GROUP_MAX; ++i) if (i < ga.n && tile_id >= ga.d[i].tile0) d = ga.d[i];
  const int tile = tile_id - d.tile0;
  const int tm = tile / d.tiles_n, tn = tile - tm * d.tiles_n;
  const int kt_begin = split * ga.ktiles_per_split;
  const int kt_end = min(kt_begin + ga.ktiles_per_split, ga.ktiles);
  GemmArgs p;
  p.A = d.dY; p.B = d.X; p.C = d.dW; p.bias = nullptr; p.aux = nullptr; p.resid = d.dW;
  p.lda = d.ldy; p.ldb = d.ldx; p.ldc = d.N; p.ldaux = 0; p.ldr = d.N;
  p.M = d.M; p.N = d.N; p.K = ga.K;
  p.c_dtype = CSMAE_F32; p.epi = EPI_RESID; p.splitk = ga.nsplit; p.tiles_m = 0; p.tiles_n = d.tiles_n; p.ktiles = ga.ktiles; p.ktiles_per_split = ga.ktiles_per_split;
  p.a_bytes = (unsigned)((long long)ga.K * d.ldy * 2); p.b_bytes = (unsigned)((long long)ga.K * d.ldx * 2);
  p.force_cfg = ga.force_cfg; p.split_stride = 0; p.colsum = d.db; p.dq_a = p.dq_b = nullptr; p.a_fmt = 0; p.aux_q8 = 0; p.q_out = nullptr;
  __shared__ __attribute__((aligned(16))) char smem[K64_LDS_BYTES];
  const DwFold fold{ga.slab, ga.cs_slab, ga.nsplit, tile_id, ga.total_tiles};
  if (d.db != nullptr && tn == 0) k64_tile<true, true, 256, true, 2>(smem, p, tm, tn, split, kt_begin, kt_end, fold);
  else k64_tile<true, true, 256, true, 1>(smem, p, tm, tn, split, kt_begin, kt_end, fold);
}
// fold of the K slices of a grouped launch: workgroup (tile, part) adds the tile's slabs in slice order and accumulates 16 rows into
// dW (16-byte accesses along rows); part 0 of the tn == 0 tiles does the same for the bias gradient.  Ordered: bit-reproducible.
__global__ __launch_bounds__(256) void dw_group_reduce_kernel(DwGroupArgs ga) {
  const int tile_id = blockIdx.x / DWR_PARTS, quarter = blockIdx.x % DWR_PARTS;
  DwDesc d = ga.d[0];
#pragma unroll
  for (int i = 1; i < DW_GROUP_MAX; ++i) if (i < ga.n && tile_id >= ga.d[i].tile0) d = ga.d[i];
  const int tile = tile_id - d.tile0;
  const int tm = tile / d.tiles_n, tn = tile - tm * d.tiles_n;
  const int m0 = tm * 256, n0 = tn * 256;
  const int c4 = threadIdx.x & 63, r0 = threadIdx.x >> 6;   // 64 float4 columns x 4 rows per pass
  const long long sstride = (long long)ga.total_tiles * 256 * 256;
  const float* base = ga.slab + (long long)tile_id * 256 * 256;
  const int n = n0 + c4 * 4;
  constexpr int RPB = 256 / DWR_PARTS;
#pragma unroll
  for (int r = quarter * RPB + r0; r < quarter * RPB + RPB; r += 4) {
    const int m = m0 + r;
    if (m >= d.M || n >= d.N) continue;
    float* dst = d.dW + (long long)m * d.N + n;
    f4_t a = *reinterpret_cast<f4_t*>(dst);
    for (int sl = 0; sl < ga.nsplit; ++sl) a += *reinterpret_cast<const f4_t*>(base + sl * sstride + r * 256 + c4 * 4);
    *reinterpret_cast<f4_t*>(dst) = a;
  }
  if (quarter == 0 && tn == 0 && d.db != nullptr) {
    const int m = m0 + threadIdx.x;
    if (m < d.M) {
      float a = d.db[m];
      for (int sl = 0; sl < ga.nsplit; ++sl) a += ga.cs_slab[((long long)sl * ga.total_tiles + tile_id) * 256 + threadIdx.x];
      d.db[m] = a;
    }
  }
}

// ------------------------------------------------------------------------------------ fp8 MFMA (BASELINE.json configs[4])
// C[M,N] = dq_a * dq_b * sum_k A8(m,k) B8(n,k): both operands K-contiguous OCP fp8 bytes (A e4m3 or e5m2, B e4m3), per-tensor scales,
// fp32 accumulation in v_mfma_scale_f32_16x16x128_f8f6f4 (unit block scales) — 2.1x the bf16 MFMA rate (tools/mfma_rate_probe.hip) at
// the SAME LDS bytes per K step: a 128-wide fp8 K step is the 128-byte row image of the pipelined bf16 kernel (16-B chunk swizzle
// c ^ (row & 7)); lane (t, g) of a fragment owns the 32 bytes k = 32g .. 32g+31 of row t = two ds_read_b128.  Every fp8 product of the
// step is brought into this one layout (dX reads a pre-transposed fp8 weight mirror), so there is no K-strided variant.
// First version: two-stage LDS ring (2 x 64 KiB), LDS-DMA from inline asm, one barrier per K step, compiler-scheduled fragment reads.
typedef int i8v_t __attribute__((ext_vector_type(8)));
template <int AFMT>
__global__ __launch_bounds__(512, 1) void gemm_fp8_kernel(GemmArgs p) {
  constexpr int BM = 256, BN = 256, WM = 128, WN = 64, NWN = 4, NW = 8, FM = 8, FN = 4, IMG = 256 * 128, STAGE = 2 * IMG;
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int t = lane & 15, g = lane >> 4;
  const int tiles = p.tiles_m * p.tiles_n;
  const int tile = xcd_remap(blockIdx.x, tiles);
  const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const i4_t rsA = make_rsrc(p.A, p.a_bytes), rsB = make_rsrc(p.B, p.b_bytes);
  // DMA: piece = 8 rows x 128 B; lane -> row (lane >> 3), physical chunk (lane & 7) = logical chunk ^ (row & 7); wave w owns pieces 4w .. 4w+3
  const int l3 = lane >> 3, kc = ((lane & 7) ^ l3) * 16;
  const int arow = m0 + w * 32 + l3, brow = n0 + w * 32 + l3;
  const unsigned avo = (unsigned)((long long)arow * p.lda + kc), bvo = (unsigned)((long long)brow * p.ldb + kc);
  auto stage = [&](int kt, int buf) {
    char* base = smem + buf * STAGE;
    const bool kok = kt * 128 + kc < p.K;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const unsigned oa = (kok && arow + q * 8 < p.M) ? avo + (unsigned)(kt * 128) + (unsigned)(q * 8 * p.lda) : OOB_OFF;
      lds_dma16(rsA, oa, base + (w * 4 + q) * 1024);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const unsigned ob = (kok && brow + q * 8 < p.N) ? bvo + (unsigned)(kt * 128) + (unsigned)(q * 8 * p.ldb) : OOB_OFF;
      lds_dma16(rsB, ob, base + IMG + (w * 4 + q) * 1024);
    }
  };
  f4_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f4_t{0.f, 0.f, 0.f, 0.f};
  const int wm = (w / NWN) * WM, wn = (w % NWN) * WN;
  const int ra0 = (wm + t) * 128, rb0 = (wn + t) * 128, sw = t & 7;   // (rows wm + 16 i + t share t's swizzle key)
  const int c0 = ((2 * g) ^ sw) << 4, c1 = ((2 * g + 1) ^ sw) << 4;
  auto frag = [&](const char* img, int roff) {
    const u4_t lo = *reinterpret_cast<const u4_t*>(img + roff + c0), hi = *reinterpret_cast<const u4_t*>(img + roff + c1);
    return i8v_t{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
  };
  const int nk = p.ktiles;
  stage(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of K step kt have landed ...
    __syncthreads();                                    // ... everybody's have, and everybody is done reading the other buffer
    if (kt + 1 < nk) stage(kt + 1, buf ^ 1);
    const char* sa = smem + buf * STAGE;
    const char* sb = sa + IMG;
    i8v_t fb[FN];
#pragma unroll
    for (int j = 0; j < FN; ++j) fb[j] = frag(sb, rb0 + j * 2048);
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const i8v_t fa = frag(sa, ra0 + i * 2048);
#pragma unroll
      for (int j = 0; j < FN; ++j)   // operands swapped (D = B_frag x A_frag): a lane ends with 4 consecutive output columns; cbsz = B's format, blgp = A's
        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fb[j], fa, acc[i][j], 0, AFMT, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    }
  }
  const float alpha = (p.dq_a ? p.dq_a[0] : 1.f) * (p.dq_b ? p.dq_b[0] : 1.f);
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] *= alpha;
  __syncthreads();   // the ring becomes the epilogue strip
  constexpr int ESTR = WN + 4, LPR = WN / 4, RPP = 64 / LPR, EROWS = 32;
  void* Cptr = p.C;
  float* ew = reinterpret_cast<float*>(smem) + w * (EROWS * ESTR);
#define EPI_CALL(TC_, E_) epilogue_rows<TC_, E_, FM, FN, WM, EROWS, ESTR, LPR, RPP>(p, Cptr, acc, ew, m0 + wm, n0 + wn, lane, t, g)
#define EPI_CALL8(E_) epilogue_rows_bf16x8<E_, FM, FN, WM, EROWS, ESTR>(p, Cptr, acc, ew, m0 + wm, n0 + wn, lane, t, g)
  const bool wide = (p.ldc % 8 == 0) && (p.epi == EPI_NONE || (p.epi == EPI_RESID ? (p.ldr % 8 == 0 && (uintptr_t)p.resid % 16 == 0)
                                                                                   : (p.ldaux % 8 == 0 && (uintptr_t)p.aux % 16 == 0)));
  if (p.c_dtype == CSMAE_BF16) {
    if (wide && (p.a_fmt & 256) && p.q_out == nullptr) {   // buffer addressing (epilogue_rows_bf16x8b)
      if (p.epi == EPI_GELU) epilogue_rows_bf16x8b<EPI_GELU, FM, FN, WM, EROWS, ESTR>(p, acc, ew, m0 + wm, n0 + wn, lane, t, g);
      else if (p.epi == EPI_DGELU) epilogue_rows_bf16x8b<EPI_DGELU, FM, FN, WM, EROWS, ESTR>(p, acc, ew, m0 + wm, n0 + wn, lane, t, g);
      else if (p.epi == EPI_RESID) epilogue_rows_bf16x8b<EPI_RESID, FM, FN, WM, EROWS, ESTR>(p, acc, ew, m0 + wm, n0 + wn, lane, t, g);
      else epilogue_rows_bf16x8b<EPI_NONE, FM, FN, WM, EROWS, ESTR>(p, acc, ew, m0 + wm, n0 + wn, lane, t, g);
    } else if (wide) { if (p.epi == EPI_GELU) EPI_CALL8(EPI_GELU); else if (p.epi == EPI_DGELU) EPI_CALL8(EPI_DGELU); else if (p.epi == EPI_RESID) EPI_CALL8(EPI_RESID); else EPI_CALL8(EPI_NONE); }
    else if (p.epi == EPI_RESID) EPI_CALL(bf16_t, EPI_RESID);
    else if (p.epi == EPI_GELU) EPI_CALL(bf16_t, EPI_GELU); else if (p.epi == EPI_DGELU) EPI_CALL(bf16_t, EPI_DGELU);
    else EPI_CALL(bf16_t, EPI_NONE);
  } else {
    if (p.epi == EPI_GELU) EPI_CALL(float, EPI_GELU); else if (p.epi == EPI_DGELU) EPI_CALL(float, EPI_DGELU);
    else if (p.epi == EPI_RESID) EPI_CALL(float, EPI_RESID); else EPI_CALL(float, EPI_NONE);
  }
#undef EPI_CALL
#undef EPI_CALL8
}

// ---- the same product, pipelined (round 3).  The kernel above waits, at every K step, for the DMA it issued one step earlier: two 64-KiB
// stages leave one step (~2.3 k clocks of MFMA work) to cover an L2 / HBM round trip of 2-4 k clocks under load — it ran at 1.4-1.5 PFLOP/s
// where the matrix pipe delivers 4.4-4.6 from registers.  Here LDS is the bf16 kernel's ring of five 32-KiB units (one operand's
// [256 rows][128 B] image each: the byte layout of a bf16 K step of 64), a step consumes units (2j, 2j+1) while 2j+2 .. 2j+4 are in
// flight or landed, and a step's fragments are read DURING the previous step — fragment i of A right behind the MFMAs of row i, in
// place; the B fragments behind the last rows, which run column by column so that each B fragment retires early.  One barrier per
// step, at its head: by then every wave holds the step's fragments in registers, so the two units can be refilled at once.
template <int AFMT>
__global__ __launch_bounds__(512, 1) void gemm_fp8_pipe_kernel(GemmArgs p) {
  constexpr int WM = 128, WN = 64, NWN = 4, NW = 8, FM = 8, FN = 4, UNIT = 256 * 128, NUNIT = 5, PP = UNIT / 1024 / NW;
  __shared__ __attribute__((aligned(16))) char smem[NUNIT * UNIT];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int t = lane & 15, g = lane >> 4;
  const int tiles = p.tiles_m * p.tiles_n;
  const int tile = xcd_remap(blockIdx.x, tiles);
  const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
  const int m0 = tm * 256, n0 = tn * 256;
  GTS(0);
  const i4_t rsA = make_rsrc(p.A, p.a_bytes), rsB = make_rsrc(p.B, p.b_bytes);
  // DMA: piece = 8 rows x 128 B; lane -> row (lane >> 3), physical chunk (lane & 7) = logical chunk ^ (row & 7); wave w owns rows 32 w .. 32 w + 31 of an image
  const int l3 = lane >> 3, kc = ((lane & 7) ^ l3) * 16;
  const unsigned avo = (unsigned)((long long)(m0 + w * 32 + l3) * p.lda + kc), bvo = (unsigned)((long long)(n0 + w * 32 + l3) * p.ldb + kc);
  // (no predicate per piece, as in k64_tile: rows beyond M / N are beyond the buffer resources' ends, and the host routes only K % 128 == 0 here)
  const unsigned ldsW = (unsigned)(size_t)LDS_PTR(char, smem) + (unsigned)(w * PP) * 1024u;
  auto dma_a = [&](int slot, int j, int q) {
    lds_dma16u(rsA, avo + ((unsigned)(j * 128) + (unsigned)(q * 8 * p.lda)), ldsW + (unsigned)(slot * UNIT + q * 1024));
  };
  auto dma_b = [&](int slot, int j, int q) {
    lds_dma16u(rsB, bvo + ((unsigned)(j * 128) + (unsigned)(q * 8 * p.ldb)), ldsW + (unsigned)(slot * UNIT + q * 1024));
  };
  f4_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f4_t{0.f, 0.f, 0.f, 0.f};
  const int wm = (w / NWN) * WM, wn = (w % NWN) * WN;
  const int ra0 = (wm + t) * 128, rb0 = (wn + t) * 128, sw = t & 7;   // (rows wm + 16 i + t share t's swizzle key)
  const int c0 = ((2 * g) ^ sw) << 4, c1 = ((2 * g + 1) ^ sw) << 4;
  // Fragment reads are inline assembly into the registers the row's MFMAs have just consumed ("+v": through plain C++ the compiler renames the
  // re-read values and spills ~90 registers), so their completion is tracked by hand: every read of a step is consumed in the NEXT
  // step, behind the `lgkmcnt(0)` at its head.  A fragment is two 16-byte halves (chunks 2g and 2g + 1 of the lane's row).
  const unsigned lds0 = (unsigned)(size_t)LDS_PTR(char, smem);
  u4_t alo[FM], ahi[FM], blo[FN], bhi[FN];
#pragma unroll
  for (int i = 0; i < FM; ++i) { alo[i] = u4_t{0, 0, 0, 0}; ahi[i] = u4_t{0, 0, 0, 0}; }
#pragma unroll
  for (int j = 0; j < FN; ++j) { blo[j] = u4_t{0, 0, 0, 0}; bhi[j] = u4_t{0, 0, 0, 0}; }
  auto read_frag = [&](int slot, int r0, auto ic, u4_t& lo, u4_t& hi) {   // fragment `ic` (16 rows = 2048 B apart) of the image in `slot`, rows from r0
    constexpr int i = decltype(ic)::value;
    const unsigned a0 = lds0 + (unsigned)(slot * UNIT + r0 + c0), a1 = lds0 + (unsigned)(slot * UNIT + r0 + c1);
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(lo) : "v"(a0), "n"(i * 2048));
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(hi) : "v"(a1), "n"(i * 2048));
  };
  auto read_a = [&](int slot, auto ic) { read_frag(slot, ra0, ic, alo[decltype(ic)::value], ahi[decltype(ic)::value]); };
  auto read_b = [&](int slot, auto jc) { read_frag(slot, rb0, jc, blo[decltype(jc)::value], bhi[decltype(jc)::value]); };
  auto join8 = [](u4_t lo, u4_t hi) { return i8v_t{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]}; };
  const int nsteps = p.ktiles, U = 2 * nsteps;
  const int issued0 = min(NUNIT - 1, U - 1);
#pragma unroll
  for (int u = 0; u < NUNIT; ++u)
    if (u <= issued0) {
#pragma unroll
      for (int q = 0; q < PP; ++q) { if (u & 1) dma_b(u, u >> 1, q); else dma_a(u, u >> 1, q); }
    }
  if (issued0 >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PP) : "memory");   // units 2, 3, 4 may stay in flight
  else if (issued0 == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PP) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  GTS(1);
  static_for<FN>([&](auto jc) { read_b(1, jc); });
  static_for<FM>([&](auto ic) { read_a(0, ic); });
  auto nxt = [](int s, int k) { s += k; return s >= NUNIT ? s - NUNIT : s; };
#define FP8_MMA(i, j) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(join8(blo[j], bhi[j]), join8(alo[i], ahi[i]), acc[i][j], 0, AFMT, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f)
  // A step is straight-line code (run-time conditions around the reads and DMA pieces made the compiler copy fragment and accumulator
  // tuples between basic blocks and spill ~90 registers), so — as in the bf16 kernel — the last three steps, which fetch less or nothing
  // and read no next fragments, are instantiations of their own: MODE 0 fetches units 2j+5 and 2j+6, 1 only 2j+5, 2 nothing, 3 = last step.
  auto step = [&](int j, int sl, auto mode_tag) {
    constexpr int MODE = decltype(mode_tag)::value;
    constexpr bool more = MODE < 3;
    // this wave's pieces of the NEXT step's units (2 j + 2, 2 j + 3) have landed; unit 2 j + 4 (issued during the previous step) may stay in flight
    // (the first step too: the prologue only waited for units 0 and 1, this step reads the next step's fragments out of units 2 and 3)
    if (MODE <= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PP) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this step's fragments have arrived (and, with the barrier, everybody's reads of its two units are done)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    const int sb_cur = nxt(sl, 1), sa = nxt(sl, 2), sb = nxt(sl, 3);
    static_for<FM - 2>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
#pragma unroll
      for (int jj = 0; jj < FN; ++jj) FP8_MMA(i, jj);
      if (more) read_a(sa, ic);
      // the 2 x PP pieces of the two freed units, ONE per row / column group: unit 2 j + 5 (B of step j + 2) into this step's A slot first — the
      // next step's head needs it, and its `vmcnt(PP)` leaves exactly the PP younger pieces of unit 2 j + 6 (into this step's B slot) in flight
      if (i < PP) { if (MODE <= 1) dma_b(sl, j + 2, i); }
      else if (MODE == 0) dma_a(sb_cur, j + 3, i - PP);
      __builtin_amdgcn_sched_barrier(0);
    });
    static_for<FN>([&](auto jc) {   // the last two rows column by column: B fragment jj retires here and is re-read for the next step
      constexpr int jj = decltype(jc)::value;
      FP8_MMA(FM - 2, jj); FP8_MMA(FM - 1, jj);
      if (more) read_b(sb, jc);
      if (MODE == 0 && jj + (FM - 2 - PP) < PP) dma_a(sb_cur, j + 3, jj + (FM - 2 - PP));
      __builtin_amdgcn_sched_barrier(0);
    });
    if (more) { read_a(sa, std::integral_constant<int, FM - 2>{}); read_a(sa, std::integral_constant<int, FM - 1>{}); }
    __builtin_amdgcn_sched_barrier(0);
  };
  int j = 0, sl = 0;   // sl: slot of the current step's A unit (unit 2 j); its B unit sits in nxt(sl, 1), the next step's in nxt(sl, 2) / nxt(sl, 3)
  for (; j < nsteps - 3; ++j, sl = nxt(sl, 2)) step(j, sl, std::integral_constant<int, 0>{});
  if (nsteps >= 3) { step(j, sl, std::integral_constant<int, 1>{}); ++j; sl = nxt(sl, 2); }
  if (nsteps >= 2) { step(j, sl, std::integral_constant<int, 2>{}); ++j; sl = nxt(sl, 2); }
  step(j, sl, std::integral_constant<int, 3>{});
  GTS(2);
#undef FP8_MMA
  const float alpha = (p.dq_a ? p.dq_a[0] : 1.f) * (p.dq_b ? p.dq_b[0] : 1.f);
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] *= alpha;
  // (no barrier: the last step reads nothing from the ring after its head barrier, and every DMA has landed)
  constexpr int ESTR = WN + 4, LPR = WN / 4, RPP = 64 / LPR, EROWS = 32;
  void* Cptr = p.C;
  float* ew = reinterpret_cast<float*>(smem) + w * (EROWS * ESTR);
#define EPI_CALL(TC_, E_) epilogue_rows<TC_, E_, FM, FN, WM, EROWS, ESTR, LPR, RPP>(p, Cptr, acc, ew, m0 + wm, n0 + wn, lane, t, g)
#define EPI_CALL8(E_) epilogue_rows_bf16x8<E_, FM, FN, WM, EROWS, ESTR>(p, Cptr, acc, ew, m0 + wm, n0 + wn, lane, t, g)
  const bool wide = (p.ldc % 8 == 0) && (p.epi == EPI_NONE || (p.epi == EPI_RESID ? (p.ldr % 8 == 0 && (uintptr_t)p.resid % 16 == 0)
                                                                                   : (p.ldaux % 8 == 0 && (uintptr_t)p.aux % 16 == 0)));
  if (p.c_dtype == CSMAE_BF16) {
    if (wide && (p.a_fmt & 256) && p.q_out == nullptr) {   // buffer addressing (epilogue_rows_bf16x8b)
      if (p.epi == EPI_GELU) epilogue_rows_bf16x8b<EPI_GELU, FM, FN, WM, EROWS, ESTR>(p, acc, ew, m0 + wm, n0 + wn, lane, t, g);
      else if (p.epi == EPI_DGELU) epilogue_rows_bf16x8b<EPI_DGELU, FM, FN, WM, EROWS, ESTR>(p, acc, ew, m0 + wm, n0 + wn, lane, t, g);
      else if (p.epi == EPI_RESID) epilogue_rows_bf16x8b<EPI_RESID, FM, FN, WM, EROWS, ESTR>(p, acc, ew, m0 + wm, n0 + wn, lane, t, g);
      else epilogue_rows_bf16x8b<EPI_NONE, FM, FN, WM, EROWS, ESTR>(p, acc, ew, m0 + wm, n0 + wn, lane, t, g);
    } else if (wide) { if (p.epi == EPI_GELU) EPI_CALL8(EPI_GELU); else if (p.epi == EPI_DGELU) EPI_CALL8(EPI_DGELU); else if (p.epi == EPI_RESID) EPI_CALL8(EPI_RESID); else EPI_CALL8(EPI_NONE); }
    else if (p.epi == EPI_RESID) EPI_CALL(bf16_t, EPI_RESID);
    else if (p.epi == EPI_GELU) EPI_CALL(bf16_t, EPI_GELU); else if (p.epi == EPI_DGELU) EPI_CALL(bf16_t, EPI_DGELU);
    else EPI_CALL(bf16_t, EPI_NONE);
  } else {
    if (p.epi == EPI_GELU) EPI_CALL(float, EPI_GELU); else if (p.epi == EPI_DGELU) EPI_CALL(float, EPI_DGELU);
    else if (p.epi == EPI_RESID) EPI_CALL(float, EPI_RESID); else EPI_CALL(float, EPI_NONE);
  }
#undef EPI_CALL
#undef EPI_CALL8
  GTS(3);
}

extern "C" int csmae_gemm_fp8(int a_fmt, long long M, long long N, long long K, const void* A, long long lda, const void* B, long long ldb,
                              void* C, long long ldc, int c_dtype, const float* bias, int epilogue, void* aux, long long ldaux,
                              const void* resid, long long ldr, const float* dq_a, const float* dq_b, void* q_out, long long ldq, int q_fmt,
                              const float* q_amax_prev, float* q_amax_next, float* q_dq, void* stream) {
  CSMAE_REQUIRE(M > 0 && N > 0 && K > 0 && N % 4 == 0 && ldc % 4 == 0, "csmae_gemm_fp8: bad geometry M=%lld N=%lld K=%lld", M, N, K);
  CSMAE_REQUIRE(A && B && (C || (q_out && c_dtype == CSMAE_BF16 && (epilogue == EPI_DGELU || epilogue == 6 || epilogue == 7 || epilogue == EPI_NONE))),
                "csmae_gemm_fp8: null operand (C may be null only with the fused fp8 copy and a non-residual epilogue: then that copy is the product's only output)");
  CSMAE_REQUIRE(a_fmt == 0 || a_fmt == 1, "csmae_gemm_fp8: a_fmt 0 (e4m3) or 1 (e5m2)");
  const int q8 = (epilogue == 6 || epilogue == 7);
  if (q8) epilogue = epilogue == 6 ? EPI_GELU : EPI_DGELU;
  CSMAE_REQUIRE(!q8 || c_dtype == CSMAE_BF16, "csmae_gemm_fp8: the 8-bit gelu' epilogues write bf16");
  CSMAE_REQUIRE(epilogue >= EPI_NONE && epilogue <= EPI_DGELU, "csmae_gemm_fp8: epilogue %d", epilogue);
  CSMAE_REQUIRE(K % 16 == 0 && lda % 16 == 0 && ldb % 16 == 0 && (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15) == 0, "csmae_gemm_fp8: K, lda, ldb multiples of 16 bytes, 16-byte aligned operands");
  CSMAE_REQUIRE(!(epilogue == EPI_GELU || epilogue == EPI_DGELU) || (aux && ldaux % 4 == 0), "csmae_gemm_fp8: gelu epilogues need aux");
  CSMAE_REQUIRE(epilogue != EPI_RESID || (resid && ldr % 4 == 0), "csmae_gemm_fp8: residual epilogue needs resid");
  CSMAE_REQUIRE(M * lda < 0xFFFFFFF0ll && N * ldb < 0xFFFFFFF0ll, "csmae_gemm_fp8: operand larger than 4 GiB");
  GemmArgs p;
  p.force_cfg = 0; p.split_stride = 0; p.colsum = nullptr;
  p.A = A; p.B = B; p.C = C; p.bias = (epilogue >= EPI_DGELU) ? nullptr : bias; p.aux = aux; p.resid = resid;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldaux = ldaux; p.ldr = ldr;
  p.M = (int)M; p.N = (int)N; p.K = (int)K; p.c_dtype = c_dtype; p.epi = epilogue; p.splitk = 1;
  p.a_bytes = (unsigned)(M * lda); p.b_bytes = (unsigned)(N * ldb);
  p.ktiles = cdiv(K, 128); p.ktiles_per_split = p.ktiles;
  p.tiles_m = cdiv(M, 256); p.tiles_n = cdiv(N, 256);
  p.dq_a = dq_a; p.dq_b = dq_b; p.a_fmt = a_fmt; p.aux_q8 = q8;
  if (c_dtype == CSMAE_BF16 && N % 8 == 0 && ldc % 8 == 0 && (M + 256) * ldc * 2 < 0xFFFFFFF0ll && (epilogue != EPI_RESID || (M + 256) * ldr * 2 < 0xFFFFFFF0ll) &&
      ((epilogue != EPI_GELU && epilogue != EPI_DGELU) || (M + 256) * ldaux * 2 < 0xFFFFFFF0ll) && !csmae_debug_opt("epi_pointers"))
    p.a_fmt |= 256;   // buffer-addressed epilogue allowed (see gemm_core)
  p.q_out = reinterpret_cast<unsigned char*>(q_out); p.ldq = ldq; p.q_fmt = q_fmt; p.q_amax_prev = q_amax_prev; p.q_amax_next = q_amax_next; p.q_dq = q_dq;
  if (q_out) {   // (the fp8 copy leaves through the 16-byte-row epilogue only)
    const bool wide = (ldc % 8 == 0) && (epilogue == EPI_NONE || (epilogue == EPI_RESID ? (ldr % 8 == 0 && (uintptr_t)resid % 16 == 0) : (ldaux % 8 == 0 && (uintptr_t)aux % 16 == 0)));
    CSMAE_REQUIRE(c_dtype == CSMAE_BF16 && wide && ldq % 8 == 0 && ((uintptr_t)q_out & 7) == 0 && q_amax_prev && q_amax_next && q_dq && (q_fmt == 0 || q_fmt == 1),
                  "csmae_gemm_fp8: the fused fp8 copy needs a bf16 output with 8-element aligned rows and the three scale pointers");
  }
  dim3 grid(p.tiles_m * p.tiles_n);
  static const bool two_stage_env = csmae_debug_opt("fp8_two_stage") != nullptr;   // A/B aid: the first (two-stage, barrier-per-step) kernel
  // the pipelined kernel fetches whole 128-byte K steps without predicates; a byte offset one tile past an operand's end must not wrap
  const bool two_stage = two_stage_env || K % 128 != 0 || (M + 256) * lda >= 0xFFFFFFF0ll || (N + 256) * ldb >= 0xFFFFFFF0ll;
  if (two_stage) {
    if (a_fmt == 0) hipLaunchKernelGGL(gemm_fp8_kernel<0>, grid, dim3(512), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(gemm_fp8_kernel<1>, grid, dim3(512), 0, (hipStream_t)stream, p);
  } else if (a_fmt == 0) CSMAE_LAUNCH(gemm_fp8_pipe_kernel<0>, grid, dim3(512), 0, (hipStream_t)stream, p);
  else CSMAE_LAUNCH(gemm_fp8_pipe_kernel<1>, grid, dim3(512), 0, (hipStream_t)stream, p);
  return csmae_check_launch("csmae_gemm_fp8");
}

// ------------------------------------------------------------------------------------ fp32 exact
template <bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs p) {
  __shared__ float As[16][68], Bs[16][68];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int tiles = p.tiles_m * p.tiles_n;
  const int wg = blockIdx.x;
  const int split = wg / tiles, tile = wg - split * tiles;
  const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
  const int m0 = tm * 64, n0 = tn * 64;
  const float* A = reinterpret_cast<const float*>(p.A);
  const float* B = reinterpret_cast<const float*>(p.B);
  const int kt_begin = split * p.ktiles_per_split, kt_end = min(kt_begin + p.ktiles_per_split, p.ktiles);
  float acc[4][4] = {};
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const int k0 = kt * 16;
#pragma unroll
    for (int e = tid; e < 1024; e += 256) {
      int mm, kk;
      if (!TA) { mm = e >> 4; kk = e & 15; } else { kk = e >> 6; mm = e & 63; }
      long long gm = m0 + mm, gk = k0 + kk;
      float v = 0.f;
      if (gm < p.M && gk < p.K) v = TA ? A[gk * p.lda + gm] : A[gm * p.lda + gk];
      As[kk][mm] = v;
      int nn;
      if (!TB) { nn = e >> 4; kk = e & 15; } else { kk = e >> 6; nn = e & 63; }
      long long gn = n0 + nn; gk = k0 + kk;
      v = 0.f;
      if (gn < p.N && gk < p.K) v = TB ? B[gk * p.ldb + gn] : B[gn * p.ldb + gk];
      Bs[kk][nn] = v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { a[i] = As[k][ty * 4 + i]; b[i] = Bs[k][tx * 4 + i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
  if (kt_begin >= kt_end && p.epi == EPI_ATOMIC) return;
  if (p.epi == EPI_SPLIT) p.C = reinterpret_cast<float*>(p.C) + (long long)split * p.split_stride;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    epi_dispatch(p, m0 + ty * 4 + i, n0 + tx * 4, f4_t{acc[i][0], acc[i][1], acc[i][2], acc[i][3]});
}

// ------------------------------------------------------------------------------------ C ABI
static int g_force_cfg = -1;  // tuning hook (tools/gemm_bench.py): -1 = heuristic
extern "C" int csmae_gemm_force_tile(int cfg) { g_force_cfg = cfg; return 0; }
// Which products run on the two-workgroups-per-CU kernel (gemm_k2.hip), per use (nn: dX through csmae_gemm, nt: csmae_gemm_ks):
// 0 never | 3 wherever eligible, whatever the size (tools) | 1 where a tile's time outside the MFMA loop is large — GELU / x gelu' epilogues, or K <= 512: the shapes it wins alone on the
// chip (tools/k2_check.py) | 2 wherever it is eligible.  Defaults (measured in the step, gpurun_out/r05f): forward products 2 — beside the other
// view's stream half-CU workgroups interleave better than whole-CU ones even where they lose alone (-0.4 ms per step) —, dX products 1 (2: +0.2 ms
// beside the whole-CU weight-gradient launches).  csmae_gemm_k2_mode / CSMAE_K2 (A/B aids).
// CSMAE_K2="nt,nn,dw" (e.g. "2,1,0", the defaults): which products run on the two-workgroups-per-CU kernels
static int k2_env(int idx, int dflt) {
  const char* e = getenv("CSMAE_K2");
  int v[3] = {-1, -1, -1};
  if (e) sscanf(e, "%d,%d,%d", &v[0], &v[1], &v[2]);
  return v[idx] >= 0 ? v[idx] : dflt;
}
static int g_k2_nn = k2_env(1, 1);
static int g_k2_nt = k2_env(0, 2);
// grouped weight-gradient launches: 0 = 256 x 256 tiles, one workgroup per CU (default) | 1 = 128 x 256 tiles, two per CU (k2_tile_tn).  Measured
// (gpurun_out/r05k-m): alone on the chip 1 is 25 % faster at 128 slots (it spreads over all 256 CUs), 0-8 % slower at 160 / 256; in the step
// +0.55 ms at the best slot setting (21.85 vs 21.30 ms): the main chain loses more to 12 DMA pieces per wave and K step on every CU than its
// LayerNorm / attention kernels gain from finding half a CU free.  Kept as an option (csmae_gemm_dw_mode) with its tests.
static int g_k2_dw = 0;   // (CSMAE_K2's third field is ignored unless the build has the kernel: csmae_gemm_dw_mode)
extern "C" int csmae_gemm_k2_mode(int nn, int nt) { g_k2_nn = nn; g_k2_nt = nt; return 0; }
bool gemm_k2_dw_built();   // gemm_k2.hip
extern "C" int csmae_gemm_dw_mode(int k2) {
  if (k2 && !gemm_k2_dw_built()) { csmae_set_error("csmae_gemm_dw_mode(1): the two-workgroups-per-CU weight-gradient kernel is not in this build (-DCSMAE_K2_DW)"); return CSMAE_ERR_UNSUPPORTED; }
  g_k2_dw = k2; return 0;
}
int gemm_force_cfg() { return g_force_cfg; }
static bool k2_wanted(int mode, int epilogue, long long K, long long N, long long M) {
  // size bounds (all modes but 3 = force): the kernel stages 1.5x the bytes of the 256 x 256 tile, which a weight that fits an XCD's L2 hides and a
  // larger one does not (ViT-H/14: every product on it +5.6 % per step, gpurun_out/r05z) — weight elements N x K and launch rows M
  static const long long wmax = csmae_debug_opt("k2_wmax") ? atoll(csmae_debug_opt("k2_wmax")) : 2400000ll;
  static const long long mmax = csmae_debug_opt("k2_mmax") ? atoll(csmae_debug_opt("k2_mmax")) : (1ll << 40);
  if (mode == 3) return true;
  if (N * K > wmax || M > mmax) return false;
  // beyond K = 512: products up to K = 1 536 with N <= 512 (ViT-B: the decoder's qkv-dX) — 21.39 against 21.50 ms over three interleaved rounds
  // (gpurun_out/r05y; K <= 2 048 / N <= 512: 21.45, K <= 768 / N <= 768: 21.45).  CSMAE_DEBUG=k2_kmax=..,k2_nmax=.. re-opens the question.
  static const int kmax = csmae_debug_opt("k2_kmax") ? atoi(csmae_debug_opt("k2_kmax")) : 1536;
  static const int nmax = csmae_debug_opt("k2_nmax") ? atoi(csmae_debug_opt("k2_nmax")) : 512;
  return mode >= 2 || (mode == 1 && (epilogue == EPI_GELU || epilogue == EPI_DGELU || K <= 512 || (K <= kmax && N <= nmax)));
}
bool gemm_k2_nn_wanted(int epilogue, long long K, long long N, long long M) { return k2_wanted(g_k2_nn, epilogue, K, N, M); }
bool gemm_k2_nt_wanted(int epilogue, long long K, long long N, long long M) { return k2_wanted(g_k2_nt, epilogue, K, N, M); }

int gemm_core(int dtype, int transA, int transB, long long M, long long N, long long K,
                     const void* A, long long lda, const void* B, long long ldb,
                     void* C, long long ldc, int c_dtype, const float* bias, int epilogue,
                     void* aux, long long ldaux, const void* resid, long long ldr,
                     int splitk, void* stream) {
  CSMAE_REQUIRE(M > 0 && N > 0 && K > 0, "csmae_gemm: empty problem M=%lld N=%lld K=%lld", M, N, K);
  CSMAE_REQUIRE(N % 4 == 0 && ldc % 4 == 0, "csmae_gemm: N and ldc must be multiples of 4 (N=%lld ldc=%lld)", N, ldc);
  const int q8 = (epilogue == 6 || epilogue == 7);   // CSMAE_EPI_GELU_Q8 / CSMAE_EPI_DGELU_Q8: gelu' as one byte per element
  if (q8) epilogue = epilogue == 6 ? EPI_GELU : EPI_DGELU;
  CSMAE_REQUIRE(!q8 || (dtype == CSMAE_BF16 && c_dtype == CSMAE_BF16), "csmae_gemm: the 8-bit gelu' epilogues belong to the bf16 path");
  CSMAE_REQUIRE(epilogue >= EPI_NONE && epilogue <= EPI_SPLIT, "csmae_gemm: bad epilogue %d", epilogue);
  CSMAE_REQUIRE((epilogue != EPI_ATOMIC && epilogue != EPI_SPLIT) || c_dtype == CSMAE_F32, "csmae_gemm: split-K accumulate needs fp32 C");
  CSMAE_REQUIRE(!(epilogue == EPI_GELU || epilogue == EPI_DGELU) || (aux && ldaux % 4 == 0), "csmae_gemm: gelu epilogues need aux");
  CSMAE_REQUIRE(epilogue != EPI_RESID || (resid && ldr % 4 == 0), "csmae_gemm: residual epilogue needs resid");
  CSMAE_REQUIRE(splitk >= 1 && (splitk == 1 || epilogue == EPI_ATOMIC || epilogue == EPI_SPLIT), "csmae_gemm: split-K only with the accumulate epilogues");
  GemmArgs p;
  p.force_cfg = g_force_cfg;
  p.split_stride = M * ldc;
  p.colsum = (epilogue == EPI_SPLIT && transA && transB && dtype == CSMAE_BF16) ? reinterpret_cast<float*>(aux) : nullptr;
  p.dq_a = p.dq_b = nullptr; p.a_fmt = 0; p.aux_q8 = q8; p.q_out = nullptr;
  p.A = A; p.B = B; p.C = C; p.bias = (epilogue >= EPI_DGELU) ? nullptr : bias; p.aux = aux; p.resid = resid;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldaux = ldaux; p.ldr = ldr;
  p.M = (int)M; p.N = (int)N; p.K = (int)K; p.c_dtype = c_dtype; p.epi = epilogue; p.splitk = splitk;
  // buffer-addressed epilogue (epilogue_rows_bf16x8b): bit 8 of a_fmt (the field is the fp8 kernels' otherwise).  A lane's 8 columns are valid as a
  // whole, and a byte offset up to 256 rows past the end of any tensor the epilogue touches must not wrap (rows beyond M are range-checked away)
  if (dtype == CSMAE_BF16 && c_dtype == CSMAE_BF16 && N % 8 == 0 && ldc % 8 == 0 && (M + 256) * ldc * 2 < 0xFFFFFFF0ll &&
      (epilogue != EPI_RESID || (M + 256) * ldr * 2 < 0xFFFFFFF0ll) && ((epilogue != EPI_GELU && epilogue != EPI_DGELU) || (M + 256) * ldaux * 2 < 0xFFFFFFF0ll) &&
      !csmae_debug_opt("epi_pointers"))   // (A/B aid: the pointer-addressed epilogues)
    p.a_fmt |= 256;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CSMAE_BF16) {
    CSMAE_REQUIRE(lda % 8 == 0 && ldb % 8 == 0, "csmae_gemm(bf16): lda and ldb must be multiples of 8 (lda=%lld ldb=%lld)", lda, ldb);
    // K-contiguous operands are fetched in 8-element K chunks; K-strided ones are zero-filled by whole rows, any K works
    CSMAE_REQUIRE((transA && transB) || K % 8 == 0, "csmae_gemm(bf16): K must be a multiple of 8 unless both operands are K-strided (K=%lld)", K);
    CSMAE_REQUIRE((((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15) == 0, "csmae_gemm(bf16): A, B, C must be 16-byte aligned");
    // K-strided operands are fetched in 8-column chunks: the row allocation must cover the rounded-up width
    CSMAE_REQUIRE(!transA || lda >= (M + 7) / 8 * 8, "csmae_gemm(bf16): transposed A needs lda >= roundup8(M)");
    CSMAE_REQUIRE(!transB || ldb >= (N + 7) / 8 * 8, "csmae_gemm(bf16): transposed B needs ldb >= roundup8(N)");
    long long abytes = (transA ? K : M) * lda * 2, bbytes = (transB ? K : N) * ldb * 2;
    CSMAE_REQUIRE(abytes < 0xFFFFFFF0ll && bbytes < 0xFFFFFFF0ll, "csmae_gemm(bf16): operand larger than 4 GiB");
    p.a_bytes = (unsigned)abytes; p.b_bytes = (unsigned)bbytes;
    p.ktiles = cdiv(K, GEMM_BK);
    // tile choice: as large as keeps >= ~1.5 rounds of the 256 CUs busy (bytes staged per flop ~ 1/BM + 1/BN)
    // tile choice (measured on the step's shapes, tools/gemm_bench.py): 256x256 wins whenever it fits, also when it leaves
    // fewer tiles than CUs (N = 768 outputs: 150 tiles) because it halves the bytes staged per flop; 256x128 never won.
    // 0: 128x128x32 (4 waves, 2 blocks/CU)   2: 256x256x32 (8 waves, 1 block/CU)   4: 256x256x64 software-pipelined (K-contiguous A)
    int cfg = (M >= 256 && N >= 256) ? ((!transA || transB) ? 4 : 2) : 0;
    if (cfg == 4 && !transA && splitk == 1) {  // 5: the same kernel with 192-row tiles, when it fills the CUs at least 10 % better
      const long long t256 = (long long)cdiv(M, 256) * cdiv(N, 256), t192 = (long long)cdiv(M, 192) * cdiv(N, 256);
      const long long c256 = cdiv(t256, 256) * 256, c192 = cdiv(t192, 256) * 192;
      if (c192 * 10 <= c256 * 9) cfg = 5;
    }
    int stg = 4;
    // 6: 128 x 256 tiles, two 4-wave workgroups per CU (k2_tile): K-contiguous A, K-strided B (dX products), whole 64-wide K steps
    const bool k2_ok = !transA && transB && splitk == 1 && K % 64 == 0 && M >= 128 && N >= 256 && (M + 128) * lda * 2 < 0xFFFFFFF0ll && (K + 64) * ldb * 2 < 0xFFFFFFF0ll;
    if (cfg >= 4 && k2_ok && gemm_k2_nn_wanted(epilogue, K, N, M)) cfg = 6;
    if (p.force_cfg >= 0) cfg = p.force_cfg & 7; else p.force_cfg = 0;
    if (cfg == 6 && !k2_ok) cfg = 4;
    if (cfg > 6) cfg = 4;
    if (cfg == 5 && (transA || splitk != 1)) cfg = 4;
    if (cfg == 4 && transA && !transB) cfg = 2;  // (no 64-wide-K instantiation for K-strided A with K-contiguous B: unused by the step)
    // the pipelined kernel fetches without per-piece predicates (k64_tile): K-contiguous operands need whole 64-wide K steps, and a byte offset up
    // to one tile past the end of an operand must not wrap
    if (cfg >= 4 && ((!(transA && transB) && K % 64 != 0) || ((transA ? K + 64 : M + 256) * lda * 2 >= 0xFFFFFFF0ll) || ((transB ? K + 64 : N + 256) * ldb * 2 >= 0xFFFFFFF0ll))) {
      cfg = 2;
    }
    if (cfg >= 4) p.ktiles = cdiv(K, 64);
    (void)stg;
    const int bm = (cfg == 0 || cfg == 6) ? 128 : (cfg == 5 ? 192 : 256), bn = cfg >= 2 ? 256 : 128;
    p.tiles_m = cdiv(M, bm); p.tiles_n = cdiv(N, bn);
    p.ktiles_per_split = cdiv(p.ktiles, splitk);
    p.splitk = cdiv(p.ktiles, p.ktiles_per_split);
    dim3 grid(p.tiles_m * p.tiles_n * p.splitk);
#define LAUNCH_CFG(TA_, TB_)                                                                                              \
    if (cfg == 1) hipLaunchKernelGGL((gemm_bf16_kernel<TA_, TB_, 256, 128, 64, 64, 3, false, 4>), grid, dim3(512), 0, st, p);     \
    else if (cfg == 2) hipLaunchKernelGGL((gemm_bf16_kernel<TA_, TB_, 256, 256, 128, 64, 4, false>), grid, dim3(512), 0, st, p);  \
    else if (cfg == 3) hipLaunchKernelGGL((gemm_bf16_kernel<TA_, TB_, 256, 256, 128, 64, 4, true>), grid, dim3(512), 0, st, p);   \
    else hipLaunchKernelGGL((gemm_bf16_kernel<TA_, TB_, 128, 128, 64, 64, 4, false>), grid, dim3(256), 0, st, p);
    if (cfg == 6) gemm_k2_launch_nn(p, st);
    else if (cfg == 5 && !transB) CSMAE_LAUNCH((gemm_bf16_k64_kernel<false, false, 192>), grid, dim3(512), 0, st, p);
    else if (cfg == 5) CSMAE_LAUNCH((gemm_bf16_k64_kernel<false, true, 192>), grid, dim3(512), 0, st, p);
    else if (cfg == 4 && transA) CSMAE_LAUNCH((gemm_bf16_k64_kernel<true, true>), grid, dim3(512), 0, st, p);
    else if (cfg == 4 && !transB) CSMAE_LAUNCH((gemm_bf16_k64_kernel<false, false>), grid, dim3(512), 0, st, p);
    else if (cfg == 4) CSMAE_LAUNCH((gemm_bf16_k64_kernel<false, true>), grid, dim3(512), 0, st, p);
    else if (!transA && !transB) { LAUNCH_CFG(false, false) }
    else if (!transA && transB) { LAUNCH_CFG(false, true) }
    else if (transA && transB) { LAUNCH_CFG(true, true) }
    else { LAUNCH_CFG(true, false) }
#undef LAUNCH_CFG
  } else if (dtype == CSMAE_F32) {
    p.a_bytes = p.b_bytes = 0;
    p.tiles_m = cdiv(M, 64); p.tiles_n = cdiv(N, 64); p.ktiles = cdiv(K, 16);
    p.ktiles_per_split = cdiv(p.ktiles, splitk);
    p.splitk = cdiv(p.ktiles, p.ktiles_per_split);
    dim3 grid(p.tiles_m * p.tiles_n * p.splitk), block(256);
    if (!transA && !transB) hipLaunchKernelGGL((gemm_f32_kernel<false, false>), grid, block, 0, st, p);
    else if (!transA && transB) hipLaunchKernelGGL((gemm_f32_kernel<false, true>), grid, block, 0, st, p);
    else if (transA && transB) hipLaunchKernelGGL((gemm_f32_kernel<true, true>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((gemm_f32_kernel<true, false>), grid, block, 0, st, p);
  } else {
    csmae_set_error("csmae_gemm: unsupported dtype %d", dtype);
    return CSMAE_ERR_UNSUPPORTED;
  }
  return csmae_check_launch("csmae_gemm");
}
extern "C" int csmae_gemm(int dtype, int transA, int transB, long long M, long long N, long long K,
                          const void* A, long long lda, const void* B, long long ldb,
                          void* C, long long ldc, int c_dtype, const float* bias, int epilogue,
                          void* aux, long long ldaux, const void* resid, long long ldr,
                          int splitk, void* stream) {
  return gemm_core(dtype, transA, transB, M, N, K, A, lda, B, ldb, C, ldc, c_dtype, bias, epilogue, aux, ldaux, resid, ldr, splitk, stream);
}

// ------------------------------------------------------------------------------------ weight gradients
// dW[M=out, N=in] += dY^T X with the token axis (K = 12 800 .. 65 792) split over the whole chip.  Each slice stores its fp32
// tile to a workspace slab with plain coalesced stores; one small kernel then folds the slabs into dW.  (The first version
// accumulated slices with fp32 atomics: 12 M L2 atomics per GEMM cost more than the GEMM itself.)
__global__ __launch_bounds__(256) void dw_reduce_kernel(long long n4, int S, long long slab4, const f4_t* __restrict__ ws, f4_t* __restrict__ dw,
                                                        const float* __restrict__ wsb, float* __restrict__ db, int M) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    f4_t a = dw[i];
    for (int s = 0; s < S; ++s) a += ws[s * slab4 + i];
    dw[i] = a;
  }
  if (db && blockIdx.x == 0)
    for (int m = threadIdx.x; m < M; m += blockDim.x) {
      float a = db[m];
      for (int s = 0; s < S; ++s) a += wsb[(long long)s * M + m];
      db[m] = a;
    }
}
extern int csmae_colsum_launch(int dtype, long long M, int N, const void* x, long long ld, float* out, void* stream);

extern "C" int csmae_gemm_dw(int dtype, long long M, long long N, long long K, const void* dY, long long ldy, const void* X, long long ldx,
                             float* dW, float* db, float* workspace, long long ws_elems, void* stream);
// Workspace of a grouped launch: [nsplit][tiles] dense 256 x 256 fp32 slabs, then [nsplit][tiles][256] column-sum partials (only
// written / read when a tile is cut into more than one K slice).
extern "C" int csmae_gemm_dw_group(int dtype, int count, long long K, const void* const* dY, const long long* ldy, const void* const* X,
                                   const long long* ldx, float* const* dW, float* const* db, const long long* M, const long long* N, int slots,
                                   float* workspace, long long ws_elems, void* stream) {
  CSMAE_REQUIRE(count > 0 && count <= DW_GROUP_MAX && K > 0 && dY && X && dW && M && N && ldy && ldx && workspace, "csmae_gemm_dw_group: bad arguments (1..%d products)", DW_GROUP_MAX);
  bool fused = dtype == CSMAE_BF16 && g_force_cfg < 0;
  for (int i = 0; i < count && fused; ++i) fused = M[i] >= 256 && N[i] >= 256 && N[i] % 4 == 0 && ldy[i] % 8 == 0 && ldx[i] % 8 == 0 && ldy[i] >= (M[i] + 7) / 8 * 8 && ldx[i] >= (N[i] + 7) / 8 * 8 &&
                                                   K * ldy[i] * 2 < 0xFFFFFFF0ll && K * ldx[i] * 2 < 0xFFFFFFF0ll && ((((uintptr_t)dY[i] | (uintptr_t)X[i] | (uintptr_t)dW[i]) & 15) == 0);
  if (!fused) {  // exact-fp32 parity mode, small products, odd layouts: one product at a time through the slab path
    for (int i = 0; i < count; ++i) {
      int rc = csmae_gemm_dw(dtype, M[i], N[i], K, dY[i], ldy[i], X[i], ldx[i], dW[i], db ? db[i] : nullptr, workspace, ws_elems, stream);
      if (rc) return rc;
    }
    return CSMAE_OK;
  }
  DwGroupArgs ga;
  if (g_k2_dw) {   // weight gradients on the two-workgroups-per-CU kernel (gemm_k2.hip k2_tile_tn)
    for (int i = 0; i < count; ++i) {
      DwDesc& d = ga.d[i];
      d.dY = dY[i]; d.X = X[i]; d.dW = dW[i]; d.db = db ? db[i] : nullptr; d.M = (int)M[i]; d.N = (int)N[i]; d.ldy = ldy[i]; d.ldx = ldx[i];
    }
    ga.n = count; ga.K = (int)K; ga.force_cfg = 0; ga.ktiles = cdiv(K, 64);
    return gemm_k2_launch_dw(ga, count, slots <= 0 ? 128 : slots, workspace, ws_elems, (hipStream_t)stream);
  }
  int tiles = 0;
  for (int i = 0; i < count; ++i) {
    DwDesc& d = ga.d[i];
    d.dY = dY[i]; d.X = X[i]; d.dW = dW[i]; d.db = db ? db[i] : nullptr; d.M = (int)M[i]; d.N = (int)N[i]; d.ldy = ldy[i]; d.ldx = ldx[i];
    d.tiles_n = cdiv(N[i], 256); d.tile0 = tiles;
    tiles += cdiv(M[i], 256) * d.tiles_n;
  }
  for (int i = count; i < DW_GROUP_MAX; ++i) ga.d[i] = ga.d[0];
  ga.n = count; ga.K = (int)K; ga.total_tiles = tiles; ga.force_cfg = 0;
  ga.ktiles = cdiv(K, 64);
  if (slots <= 0) slots = 128;
  long long S = slots / tiles;                       // K slices per tile: as few as fill `slots` workgroups (the rest of the chip runs the main stream)
  if (S > ga.ktiles / 8) S = ga.ktiles / 8;          // (a slice is at least 8 K steps)
  const long long per_slice = (long long)tiles * (256 * 256 + 256);
  if (S > ws_elems / per_slice) S = ws_elems / per_slice;
  if (S < 1) S = 1;
  ga.ktiles_per_split = cdiv(ga.ktiles, S);
  ga.nsplit = cdiv(ga.ktiles, ga.ktiles_per_split);
  ga.slab = workspace; ga.cs_slab = workspace + (long long)ga.nsplit * tiles * 256 * 256;
  CSMAE_REQUIRE(ga.nsplit == 1 || ws_elems >= ga.nsplit * per_slice, "csmae_gemm_dw_group: workspace too small");
  hipLaunchKernelGGL(gemm_dw_group_kernel, dim3(tiles * ga.nsplit), dim3(512), 0, (hipStream_t)stream, ga);
  if (ga.nsplit > 1) hipLaunchKernelGGL(dw_group_reduce_kernel, dim3(tiles * DWR_PARTS), dim3(256), 0, (hipStream_t)stream, ga);
  return csmae_check_launch("csmae_gemm_dw_group");
}
extern "C" int csmae_gemm_dw(int dtype, long long M, long long N, long long K, const void* dY, long long ldy, const void* X, long long ldx,
                             float* dW, float* db, float* workspace, long long ws_elems, void* stream) {
  CSMAE_REQUIRE(M > 0 && N > 0 && K > 0 && N % 4 == 0 && workspace && ws_elems >= M * N + M, "csmae_gemm_dw: bad arguments / workspace too small");
  const int tile = dtype == CSMAE_BF16 ? ((M >= 256 && N >= 256) ? 256 : 128) : 64;
  const bool k64 = dtype == CSMAE_BF16 && tile == 256 && (g_force_cfg < 0 || (g_force_cfg & 7) == 4);  // same choice as csmae_gemm
  const int kt = dtype == CSMAE_BF16 ? (k64 ? 64 : GEMM_BK) : 16;
  const long long tiles = (long long)cdiv(M, tile) * cdiv(N, tile), ktiles = cdiv(K, kt);
  static const int slots256 = csmae_debug_opt("dw_slots") ? atoi(csmae_debug_opt("dw_slots")) : 160;  // blocks per weight-gradient GEMM (tuning aid): ~5/8 of the CUs — the rest runs the main stream — and fewer fp32 slabs than a full-chip split (flat optimum 128..192)
  const long long slots = dtype == CSMAE_BF16 ? (tile == 256 ? slots256 : 512) : 2048;
  long long S = slots / tiles;
  if (S > ktiles / (k64 ? 4 : 6)) S = ktiles / (k64 ? 4 : 6);
  if (S > ws_elems / (M * N + M)) S = ws_elems / (M * N + M);
  if (S < 1) S = 1;
  const int kps = cdiv(ktiles, S);
  S = cdiv(ktiles, kps);
  float* wsb = workspace + S * M * N;  // bias-gradient slab [S][M] (filled by the bf16 kernel's ones-operand MFMAs)
  const bool fused_bias = db && dtype == CSMAE_BF16;
  int rc = csmae_gemm(dtype, 1, 1, M, N, K, dY, ldy, X, ldx, workspace, N, CSMAE_F32, nullptr, EPI_SPLIT, fused_bias ? wsb : nullptr, 0, nullptr, 0, (int)S, stream);
  if (rc) return rc;
  const long long n4 = M * N / 4;
  hipLaunchKernelGGL(dw_reduce_kernel, dim3((unsigned)fmin((double)cdiv(n4, 256), 2048.0)), dim3(256), 0, (hipStream_t)stream, n4, (int)S, n4,
                     reinterpret_cast<const f4_t*>(workspace), reinterpret_cast<f4_t*>(dW), wsb, fused_bias ? db : nullptr, (int)M);
  if (db && !fused_bias) return csmae_colsum_launch(dtype, K, (int)M, dY, ldy, db, stream);
  return csmae_check_launch("csmae_gemm_dw");
}

