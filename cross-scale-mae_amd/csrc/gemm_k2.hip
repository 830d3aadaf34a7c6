// Two-workgroups-per-CU bf16 GEMM (round 5) and the K-slab weight mirror it reads.  See gemm_common.h for the shared epilogues.
#include "gemm_common.h"
#ifdef GEMM_TIMING
extern "C" int csmae_debug_k2_ts(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gemm_ts), sizeof(g_gemm_ts)); }
#endif

// ------------------------------------------------------------------------------------ bf16 MFMA, two workgroups per CU (round 5)
// The 256 x 256 workgroup above owns its CU (160 KiB of LDS, every VGPR): while it waits for its first operands (prologue, ~4 k clocks) and
// while it stores its C tile (5 k .. 15 k clocks) the CU's matrix pipes idle — 28-36 % of a tile's life at K = 512 .. 768, which is most of
// the step.  This kernel halves the workgroup instead of tuning it: 128 x 256 tile, FOUR waves (one per SIMD, each the same
// 128 x 64 wave tile and the same software-pipelined K step as k64_tile), 80 KiB of LDS, <= 256 VGPRs — so TWO workgroups share a CU and
// one's prologue / epilogue runs under the other's main loop.  What makes the ring fit in 80 KiB:
//   * the four waves sit side by side along N (1 x 4), so a wave's 64 B columns are read by nobody else: every wave stages ITS OWN B rows
//     into a wave-private ring (no barrier, no cross-wave hazard: the wave's own counted vmcnt orders DMA against its reads);
//   * B is fetched in 32-wide K HALVES (a half is consumed in half a K step), which needs rows of 64 B that are still whole cache lines
//     for the DMA — the "K-slab" weight mirror Wk[K/32][N][32] (csmae_weights_kslab): 16 consecutive rows of a slab are 1 KiB contiguous;
//     (K-strided B, i.e. dX = dY W with W [K][N]: 8 k-rows x 128 B per piece, whole lines by nature);
//   * A (activations, shared by the four waves) stays a [128 rows][64 k] image of whole 128-B lines, two slots, one barrier per K step.
// LDS: A 2 x 16 KiB | per wave 3 x 4 KiB of B halves (B_j^0, B_j^1, B_{j+1}^0 rotate through three slots) = 80 KiB; every unit is in flight
// for exactly one K step.  The epilogue strip of a wave is its own B ring (8.5 of 12 KiB): no barrier between loop and epilogue either.
// Cost: 96 instead of 64 KiB staged per 256 x 256 x 64 of MFMA work (the A image serves half as many columns).
// main-loop ablations, compile-time only (tools/k2_variants.sh builds variant libraries; never in the product build):
// -DK2_ABL=1 no fragment reads | 2 no DMA pieces | 4 no MFMAs | 8 no barrier | 16 every DMA piece re-fetches K step 0 (cache-resident operands)
#ifndef K2_ABL
#define K2_ABL 0
#endif
template <bool TB>
__device__ __forceinline__ void k2_tile(const GemmArgs& p, const int tm, const int tn) {
  constexpr int WM = 128, WN = 64, NW = 4, FM = 8, FN = 4;
  constexpr int ASLOT = 128 * 64 * 2, BSLOT = 64 * 32 * 2, BWAVE = 3 * BSLOT, BBASE = 2 * ASLOT;   // 16 KiB | 4 KiB | 12 KiB | 32 KiB
  __shared__ __attribute__((aligned(16))) char smem[BBASE + NW * BWAVE];                            // 80 KiB
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int t = lane & 15, g = lane >> 4;
  const int m0 = tm * 128, n0 = tn * 256, wn = w * WN;
  GTS(0);
  const i4_t rsA = make_rsrc(p.A, p.a_bytes), rsB = make_rsrc(p.B, p.b_bytes);
  // ---- DMA descriptors
  // A image [128 rows][64 k]: piece = 8 rows x 128 B, lane -> row (lane >> 3), physical 16-B chunk (lane & 7) holds logical chunk ^ (row & 7);
  //   wave w stages rows 32 w .. 32 w + 31 (pieces 4 w .. 4 w + 3)
  // B half [64 rows][32 k] (K-contiguous, K-slab mirror): piece = 16 rows x 64 B = 1 KiB contiguous in the slab, lane -> row (lane >> 2),
  //   physical chunk (lane & 3) holds logical chunk ^ ((row >> 1) & 3)
  // B half [32 k][64 cols] (K-strided): piece = 8 k-rows x 128 B, 32-B granule swizzle of the transposing read (see read_b)
  const int l3 = lane >> 3;
  const unsigned avo = (unsigned)(((long long)(m0 + w * 32 + l3) * p.lda + ((lane & 7) ^ l3) * 8) * 2);
  const unsigned aqs = (unsigned)(8 * p.lda * 2);
  unsigned bvo[2], bqs, bhs;   // lane offset (even / odd piece: they differ in the K-strided image's swizzle only) | advance per piece | advance per K half
  if (!TB) {
    const int r = lane >> 2, c = (lane & 3) ^ ((r >> 1) & 3);
    bvo[0] = bvo[1] = (unsigned)(((long long)(n0 + wn + r) * 32 + c * 8) * 2);
    bqs = 16u * 64u;
    bhs = (unsigned)(p.ldb * 64);        // ldb = rows per K slab (the weight's out-features)
  } else {
    // k-row = 8 q + (lane >> 3), 16-B chunk (lane & 7) of its 128 B; the 32-B granule (chunk >> 1) holds logical granule ^ key(k-row),
    // key(r) = ((r >> 1) & 1) | (((r >> 3) & 1) << 1): the 8 k-rows one LDS cycle of a transposing read touches ({0..3, 8..11} + 16 s + 4 hi)
    // fall on 8 distinct 32-B slots of the 256-B bank row
    const int kr = lane >> 3, ch = lane & 7;
#pragma unroll
    for (int qo = 0; qo < 2; ++qo) {
      const int key = ((kr >> 1) & 1) | (qo << 1);
      bvo[qo] = (unsigned)(((long long)kr * p.ldb + n0 + wn + ((((ch >> 1) ^ key) << 1) | (ch & 1)) * 8) * 2);
    }
    bqs = (unsigned)(8 * p.ldb * 2);
    bhs = (unsigned)(32 * p.ldb * 2);
  }
  const unsigned lds0 = (unsigned)(size_t)LDS_PTR(char, smem);
  const unsigned ldsA = lds0 + (unsigned)(w * 4) * 1024u, ldsB = lds0 + (unsigned)(BBASE + w * BWAVE);
  auto dma_a = [&](int slot, int j, int q) {   // piece q of this wave's share of A_j
    if (K2_ABL & 2) return;
    if (K2_ABL & 16) j = 0;
    lds_dma16u(rsA, avo + ((unsigned)j * 128u + (unsigned)q * aqs), ldsA + (unsigned)(slot * ASLOT + q * 1024));
  };
  auto dma_b = [&](int slot, int u, int q) {   // piece q of this wave's B half u (= 2 j + h)
    if (K2_ABL & 2) return;
    if (K2_ABL & 16) u &= 1;
    lds_dma16u(rsB, bvo[q & 1] + ((unsigned)u * bhs + (unsigned)q * bqs), ldsB + (unsigned)(slot * BSLOT + q * 1024));
  };
  f4_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int jj = 0; jj < FN; ++jj) acc[i][jj] = f4_t{0.f, 0.f, 0.f, 0.f};
  // fragment offsets: A as in k64_tile (K half h: ^ (h << 6), fragment i: + i * 2048); B half (K-contiguous): row (16 jn + t) * 64 B,
  // chunk g ^ ((t >> 1) & 3) — the four lane groups of a ds_read_b128 fall on 16 distinct 16-B slots of the 256-B bank row
  const int ra0 = t * 128 + ((g ^ (t & 7)) << 4);
  // B half (K-strided, [32 k][64 cols], 128-B rows): lane (t, g) addresses k-row 8 g + (t >> 2) (+ 4 for the fragment's high half), the 8 bytes
  // (t & 3) of granule jn ^ key — bits 5..6 of the offset hold only that XOR, so fragment jn is `rb0 ^ (jn << 5)`
  const int rb0 = !TB ? t * 64 + ((g ^ ((t >> 1) & 3)) << 4)
                      : (8 * g + (t >> 2)) * 128 + ((((t >> 3) & 1) | ((g & 1) << 1)) << 5) + (t & 3) * 8;
  constexpr int BOPS = TB ? 2 * FN : FN;
  constexpr int W0_0 = (FM - 2) + BOPS, W0_I = (FM - 1) + BOPS, W1 = FM - 2;
  static_assert(W0_I <= 15, "lgkmcnt is a 4-bit counter");
  auto read_a = [&](int slot, int h, auto ic, s8_t& fa) {
    constexpr int i = decltype(ic)::value;
    if (K2_ABL & 1) return;
    const unsigned addr = lds0 + (unsigned)(slot * ASLOT) + (unsigned)(ra0 ^ (h << 6));
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(fa) : "v"(addr), "n"(i * 2048));
  };
  auto read_b = [&](int slot, s8_t (&fb)[FN]) {
    if (K2_ABL & 1) return;
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
    if (K2_ABL & 4) return;
#pragma unroll
    for (int jj = 0; jj < FN; ++jj)
      acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8_t, fb[jj]), __builtin_bit_cast(bf8_t, fa), acc[i][jj], 0, 0, 0);
  };
  // ---- prologue: A_0, B_0^0, B_0^1 (needed at once), A_1, B_1^0 (needed at step 0's barrier).  B half u lives in slot u % 3, A_j in slot j & 1.
  const int nsteps = p.ktiles;
#pragma unroll
  for (int q = 0; q < 4; ++q) dma_a(0, 0, q);
#pragma unroll
  for (int q = 0; q < 4; ++q) dma_b(0, 0, q);
#pragma unroll
  for (int q = 0; q < 4; ++q) dma_b(1, 1, q);
  if (nsteps >= 2) {
#pragma unroll
    for (int q = 0; q < 4; ++q) dma_a(1, 1, q);
#pragma unroll
    for (int q = 0; q < 4; ++q) dma_b(2, 2, q);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
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
  // One K step (64 wide) = 16 rows of 4 MFMAs, straight-line code.  Slots on entry: A_j in `sa`, B_j^0 (already in fb0) in `sb`, B_j^1 in sb + 1,
  // B_{j+1}^0 in sb + 2 (mod 3).  DMA of the step, one piece per row: B_{j+1}^1 into B_j^0's slot (rows 1, 3, 5, 7: fb0 was complete at row 0),
  // B_{j+2}^0 into B_j^1's slot (rows 8, 9, 14, 15: fb1 was complete at row 8), A_{j+2} into A_j's slot behind the barrier (rows 10 .. 13).
  // vmcnt (in issue order per wave and step: 4 b1 | 2 b0 | 4 a | 2 b0): at the step's head B_j^1 — issued in the previous step's first half —
  // has the 8 pieces of that step's second half behind it; at the barrier A_{j+1} and B_{j+1}^0 have this step's 4 + 2 behind them.
  // MODE 0: steady state   2: second-to-last step (nothing left to fetch but B_{j+1}^1)   3: last step (no fetch, no prefetch, no barrier)
  auto step = [&](int j, int sa, int sb, auto mode_tag) {
    constexpr int MODE = decltype(mode_tag)::value;
    constexpr bool more = MODE < 3;
    const int sb1 = nx3(sb, 1), sb2 = nx3(sb, 2);
    if (MODE == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
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
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every fragment of the step is in registers (the last step too: rows 2 .. 7 below read fa[2 .. 7])
    if (more && !(K2_ABL & 8)) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (more) {
      read_a(sa ^ 1, 0, std::integral_constant<int, 0>{}, fa[0]); read_a(sa ^ 1, 0, std::integral_constant<int, 1>{}, fa[1]); read_b(sb2, fb0);
    }
    static_for<FM - 2>([&](auto ic) {
      constexpr int i = decltype(ic)::value + 2;
      mma_row(i, fa[i], fb1);
      if (more) read_a(sa ^ 1, 0, std::integral_constant<int, i>{}, fa[i]);
      if (MODE == 0) { if (i < 6) dma_a(sa, j + 2, i - 2); else dma_b(sb1, 2 * j + 4, i - 4); }
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  int j = 0, sa = 0, sb = 0;
  for (; j < nsteps - 2; ++j, sa ^= 1, sb = nx3(sb, 2)) step(j, sa, sb, std::integral_constant<int, 0>{});
  if (nsteps >= 2) { step(j, sa, sb, std::integral_constant<int, 2>{}); ++j; sa ^= 1; sb = nx3(sb, 2); }
  step(j, sa, sb, std::integral_constant<int, 3>{});
  GTS(2);
  if ((p.force_cfg & 16) && acc[0][0][0] != 123456.0f) return;  // tuning aid: main loop only
  // ---- epilogue: the k64 kernel's row-segment epilogues; the strip is this wave's own B ring (its last reads were waited for in the last step)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  constexpr int ESTR = WN + 4, LPR = WN / 4, RPP = 64 / LPR, EROWS = 32;
  static_assert(EROWS * ESTR * 4 <= BWAVE, "epilogue strip must fit the wave's B ring");
  void* Cptr = p.C;
  float* ew = reinterpret_cast<float*>(smem + BBASE + w * BWAVE);
#define EPI_CALL(TC_, E_) epilogue_rows<TC_, E_, FM, FN, WM, EROWS, ESTR, LPR, RPP>(p, Cptr, acc, ew, m0, n0 + wn, lane, t, g)
#define EPI_CALL8(E_) epilogue_rows_bf16x8<E_, FM, FN, WM, EROWS, ESTR>(p, Cptr, acc, ew, m0, n0 + wn, lane, t, g)
#define EPI_CALL8B(E_) epilogue_rows_bf16x8b<E_, FM, FN, WM, EROWS, ESTR>(p, acc, ew, m0, n0 + wn, lane, t, g)
  const bool wide = (p.ldc % 8 == 0) && (p.epi == EPI_NONE || (p.epi == EPI_RESID ? (p.ldr % 8 == 0 && (uintptr_t)p.resid % 16 == 0)
                                                                                   : (p.ldaux % 8 == 0 && (uintptr_t)p.aux % 16 == 0)));
  if (p.c_dtype == CSMAE_BF16) {
    if (wide && p.epi == EPI_NONE) epilogue_rows_bf16_plain<FM, FN, WM>(p, Cptr, acc, reinterpret_cast<char*>(ew), m0, n0 + wn, lane, t, g);
    else if (wide && (p.a_fmt & 256)) { if (p.epi == EPI_GELU) EPI_CALL8B(EPI_GELU); else if (p.epi == EPI_DGELU) EPI_CALL8B(EPI_DGELU); else EPI_CALL8B(EPI_RESID); }
    else if (wide) { if (p.epi == EPI_GELU) EPI_CALL8(EPI_GELU); else if (p.epi == EPI_DGELU) EPI_CALL8(EPI_DGELU); else EPI_CALL8(EPI_RESID); }
    else if (p.epi == EPI_RESID) EPI_CALL(bf16_t, EPI_RESID);
    else if (p.epi == EPI_GELU) EPI_CALL(bf16_t, EPI_GELU); else if (p.epi == EPI_DGELU) EPI_CALL(bf16_t, EPI_DGELU);
    else EPI_CALL(bf16_t, EPI_NONE);
  } else {
    if (p.epi == EPI_GELU) EPI_CALL(float, EPI_GELU); else if (p.epi == EPI_DGELU) EPI_CALL(float, EPI_DGELU);
    else if (p.epi == EPI_RESID) EPI_CALL(float, EPI_RESID); else EPI_CALL(float, EPI_NONE);
  }
#undef EPI_CALL
#undef EPI_CALL8
#undef EPI_CALL8B
  GTS(3);
}
template <bool TB>
__global__ __launch_bounds__(256, 2) void gemm_bf16_k2_kernel(GemmArgs p) {
  const int tile = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
  const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
#if K2_ABL & 32
  // Phase offset between the two workgroups of a CU (ablation build only: -DK2_ABL=32, tools/k2_variants.sh).  A launch's first 512 workgroups start
  // together and every tile takes the same time, so the pair on a CU stays IN phase for the whole launch; here the second workgroup of every CU
  // (ids 256 .. 511) sleeps about half a tile time once.  Measured (gpurun_out/r05d): the delayed workgroup's loop runs at 2 100 instead of 2 480
  // clocks per K step and the launch as a whole no faster — a workgroup alone on the matrix pipes is bound by its own DMA issue.  Not in the product.
  if (blockIdx.x >= 256 && blockIdx.x < 512) {
    for (int i = (int)p.split_stride; i > 0; --i) __builtin_amdgcn_s_sleep(127);   // 127 x 64 clocks per iteration
  }
#endif
  k2_tile<TB>(p, tm, tn);
}
#if K2_ABL & 32
static long long k2_stagger_sleeps(int nsteps, int epi) {   // CSMAE_DEBUG=k2_stagger=a:b: a + b * K steps clocks (ablation build only)
  static int a = -1, b = -1;
  if (a < 0) { const char* e = csmae_debug_opt("k2_stagger"); if (!e || sscanf(e, "%d:%d", &a, &b) != 2) { a = 0; b = 0; } }
  if (a == 0 && b == 0) return 0;
  const long long clk = a + (long long)b * nsteps + ((epi == EPI_GELU || epi == EPI_DGELU) ? 4000 : 0);
  return clk / (127 * 64);
}
#else
static long long k2_stagger_sleeps(int, int) { return 0; }
#endif
int gemm_k2_launch_nn(const GemmArgs& p0, hipStream_t st) {
  GemmArgs p = p0;
  p.split_stride = k2_stagger_sleeps(p.ktiles, p.epi);
  CSMAE_LAUNCH((gemm_bf16_k2_kernel<true>), dim3(p.tiles_m * p.tiles_n), dim3(256), 0, st, p);
  return 0;
}

// ---- y = x W^T with W given as its K-slab mirror Wk[K/32][N][32] (csmae_weights_kslab): the forward products on the two-workgroups-per-CU
// kernel (k2_tile).  `B_plain` ([N][K], ldb_plain) is the same weight in torch's layout: shapes the kernel does not take (K % 64, small M / N,
// fp32 parity mode) go through csmae_gemm with it, so the caller never branches.
extern "C" int csmae_gemm_ks(int dtype, long long M, long long N, long long K, const void* A, long long lda, const void* Bk, long long slab_rows,
                             const void* B_plain, long long ldb_plain, void* C, long long ldc, int c_dtype, const float* bias, int epilogue,
                             void* aux, long long ldaux, const void* resid, long long ldr, void* stream) {
  const int epi_kind = epilogue == 6 ? EPI_GELU : (epilogue == 7 ? EPI_DGELU : epilogue);
  const bool ok = dtype == CSMAE_BF16 && Bk && gemm_k2_nt_wanted(epi_kind, K, N, M) && (gemm_force_cfg() < 0 || (gemm_force_cfg() & 7) == 6) && K >= 64 && K % 64 == 0 && M >= 128 && N >= 256 && N % 4 == 0 && ldc % 4 == 0 && ldc >= N &&
                  lda % 8 == 0 && lda >= K && slab_rows >= N && slab_rows % 4 == 0 && (M + 128) * lda * 2 < 0xFFFFFFF0ll && (K / 32) * slab_rows * 64 < 0xFFFFFFF0ll &&
                  (((uintptr_t)A | (uintptr_t)Bk | (uintptr_t)C) & 15) == 0;
  if (!ok) {
    CSMAE_REQUIRE(B_plain != nullptr, "csmae_gemm_ks: shape not taken by the K-slab kernel and no plain weight given (M=%lld N=%lld K=%lld)", M, N, K);
    return gemm_core(dtype, 0, 0, M, N, K, A, lda, B_plain, ldb_plain, C, ldc, c_dtype, bias, epilogue, aux, ldaux, resid, ldr, 1, stream);
  }
  const int q8 = (epilogue == 6 || epilogue == 7);
  if (q8) epilogue = epilogue == 6 ? EPI_GELU : EPI_DGELU;
  CSMAE_REQUIRE(!q8 || c_dtype == CSMAE_BF16, "csmae_gemm_ks: the 8-bit gelu' epilogues write bf16");
  CSMAE_REQUIRE(epilogue >= EPI_NONE && epilogue <= EPI_DGELU, "csmae_gemm_ks: bad epilogue %d", epilogue);
  CSMAE_REQUIRE(!(epilogue == EPI_GELU || epilogue == EPI_DGELU) || (aux && ldaux % 4 == 0), "csmae_gemm_ks: gelu epilogues need aux");
  CSMAE_REQUIRE(epilogue != EPI_RESID || (resid && ldr % 4 == 0), "csmae_gemm_ks: residual epilogue needs resid");
  GemmArgs p;
  p.force_cfg = 0; p.split_stride = 0; p.colsum = nullptr; p.dq_a = p.dq_b = nullptr; p.a_fmt = 0; p.aux_q8 = q8; p.q_out = nullptr;
  p.A = A; p.B = Bk; p.C = C; p.bias = (epilogue >= EPI_DGELU) ? nullptr : bias; p.aux = aux; p.resid = resid;
  p.lda = lda; p.ldb = slab_rows; p.ldc = ldc; p.ldaux = ldaux; p.ldr = ldr;
  p.M = (int)M; p.N = (int)N; p.K = (int)K; p.c_dtype = c_dtype; p.epi = epilogue; p.splitk = 1;
  p.a_bytes = (unsigned)(M * lda * 2); p.b_bytes = (unsigned)((K / 32) * slab_rows * 64);
  p.ktiles = (int)(K / 64); p.ktiles_per_split = p.ktiles;
  p.tiles_m = cdiv(M, 128); p.tiles_n = cdiv(N, 256);
  if (c_dtype == CSMAE_BF16 && N % 8 == 0 && ldc % 8 == 0 && (M + 256) * ldc * 2 < 0xFFFFFFF0ll && (epilogue != EPI_RESID || (M + 256) * ldr * 2 < 0xFFFFFFF0ll) &&
      ((epilogue != EPI_GELU && epilogue != EPI_DGELU) || (M + 256) * ldaux * 2 < 0xFFFFFFF0ll) && !csmae_debug_opt("epi_pointers"))
    p.a_fmt |= 256;
  p.split_stride = k2_stagger_sleeps(p.ktiles, p.epi);
  CSMAE_LAUNCH((gemm_bf16_k2_kernel<false>), dim3(p.tiles_m * p.tiles_n), dim3(256), 0, (hipStream_t)stream, p);
  return csmae_check_launch("csmae_gemm_ks");
}
// K-slab mirrors of `count` weights in one launch: desc[i] = {offset (elements) of weight i in `src` AND in `dst`, out-features N, in-features K}
// (K % 32 == 0): dst[off + (k / 32) * N * 32 + n * 32 + k % 32] = src[off + n * K + k].  A workgroup moves 32 rows x 64 k through LDS: 128-B
// row segments in, 2 x (32 rows x 64 B = 2 KiB contiguous) out.  blockIdx.y = weight.
__global__ __launch_bounds__(256) void weights_kslab_kernel(const long long* __restrict__ desc, const bf16_t* __restrict__ src, bf16_t* __restrict__ dst) {
  const long long off = desc[blockIdx.y * 3 + 0];
  const int N = (int)desc[blockIdx.y * 3 + 1], K = (int)desc[blockIdx.y * 3 + 2];
  const int kb2 = K / 64, nblk = (N + 31) / 32;
  __shared__ uint4 tile[32][9];   // 32 rows x 128 B (+ pad)
  for (int b = blockIdx.x; b < nblk * kb2; b += gridDim.x) {
    const int nb = b / kb2, k2 = b - nb * kb2;
    const int r = threadIdx.x >> 3, c = threadIdx.x & 7, n = nb * 32 + r;
    __syncthreads();
    if (n < N) tile[r][c] = *reinterpret_cast<const uint4*>(src + off + (long long)n * K + k2 * 64 + c * 8);
    __syncthreads();
    // out: slab 2 k2 + h, rows nb * 32 .. + 31, 64 B each: thread -> (h = tid >> 7, row = (tid >> 2) & 31, 16-B chunk tid & 3)
    const int h = threadIdx.x >> 7, ro = (threadIdx.x >> 2) & 31, co = threadIdx.x & 3, no = nb * 32 + ro;
    if (no < N) *reinterpret_cast<uint4*>(dst + off + ((long long)(2 * k2 + h) * N + no) * 32 + co * 8) = tile[ro][h * 4 + co];
  }
}
extern "C" int csmae_weights_kslab(int count, const long long* desc, int max_blocks, const void* src_bf16, void* dst_bf16, void* stream) {
  CSMAE_REQUIRE(count > 0 && desc && src_bf16 && dst_bf16 && max_blocks > 0, "csmae_weights_kslab: bad arguments");
  hipLaunchKernelGGL(weights_kslab_kernel, dim3((unsigned)max_blocks, (unsigned)count), dim3(256), 0, (hipStream_t)stream, desc,
                     reinterpret_cast<const bf16_t*>(src_bf16), reinterpret_cast<bf16_t*>(dst_bf16));
  return csmae_check_launch("csmae_weights_kslab");
}


// (round 5 option, measured +0.55 ms in the step: compiled only with -DCSMAE_K2_DW — tools/k2_variants.sh; the product library answers
// csmae_gemm_dw_mode(1) with CSMAE_ERR_UNSUPPORTED)
#ifdef CSMAE_K2_DW
// ------------------------------------------------------------------------------------ weight gradients on the two-workgroups-per-CU structure
// dW[M = out][N = in] += sum_k dY[k][m] X[k][n]: both operands K-strided (token-major), i.e. every fragment is two transposing reads
// (ds_read_b64_tr_b16).  The one-workgroup kernel (k64_tile<true, true>) leaves those reads to the compiler and runs 3 000 - 3 500 clocks per K
// step against 2 300 for the K-contiguous kernels (tools/dw_cycles.py); it also owns its CU for the 200 - 500 us of a launch, so that the
// main chain's LayerNorm / attention kernels run on the CUs that are left.  Here: the k2 geometry (128 x 256 tile, four waves side by side
// along N, wave-private B ring, 80 KiB) with every LDS read issued from inline asm and counted by hand.  lgkmcnt is a 4-bit counter and a K step
// has 48 reads per wave, so an A fragment is re-read FOUR rows after the row that consumed it (for the row four rows ahead) instead of at
// once: at most 3 fragment pairs and one 8-read B burst are younger than the read a row waits for (<= 14).  The step's one barrier sits behind
// row 13 (the last read of A_j was issued behind row 11); A_{j+2} is fetched behind it.
//   A image [64 k][128 m]: 256-B rows, piece = 4 k-rows x 256 B, 32-B granule swizzle g ^ tr_key(k-row) (8 granules: the 8 k-rows one LDS
//   cycle of a transposing read touches fall on 8 distinct slots of the 256-B bank row).  B halves as in k2_tile<TB = true>.
struct DwFold2 { float* slab; float* cs_slab; int nsplit; int slot; int nslots; };
#define K2_LDS_BYTES (2 * 64 * 128 * 2 + 4 * 3 * 32 * 64 * 2)   // 80 KiB
template <bool CS>   // CS: this tile also sums dY over the tokens (bias gradient; the tn == 0 tiles) — an instantiation of its own keeps the other tiles' loop free of branches
__device__ __forceinline__ void k2_tile_tn(char* smem, const GemmArgs& p, const int tm, const int tn, const int split, const int kt_begin, const int kt_end, const DwFold2 fold) {
  constexpr int WM = 128, WN = 64, NW = 4, FM = 8, FN = 4;
  constexpr int ASLOT = 64 * 128 * 2, BSLOT = 32 * 64 * 2, BWAVE = 3 * BSLOT, BBASE = 2 * ASLOT;
  static_assert(BBASE + NW * BWAVE == K2_LDS_BYTES, "LDS layout");
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int t = lane & 15, g = lane >> 4;
  const int m0 = tm * 128, n0 = tn * 256, wn = w * WN;
  GTS(0);
  const i4_t rsA = make_rsrc(p.A, p.a_bytes), rsB = make_rsrc(p.B, p.b_bytes);
  // A: k-row = 16 w + 4 q + (lane >> 4), 16-B chunk (lane & 15); key = (k-row & 3) | (((k-row >> 3) & 1) << 2) = (lane >> 4) | ((q >> 1) << 2)
  unsigned avo[2], bvo[2];
  {
    const int kr = lane >> 4, ch = lane & 15;
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) {
      const int key = kr | (qq << 2);
      avo[qq] = (unsigned)(((long long)(w * 16 + kr) * p.lda + m0 + ((((ch >> 1) ^ key) << 1) | (ch & 1)) * 8) * 2);
    }
  }
  {
    const int kr = lane >> 3, ch = lane & 7;
#pragma unroll
    for (int qo = 0; qo < 2; ++qo) {
      const int key = ((kr >> 1) & 1) | (qo << 1);
      bvo[qo] = (unsigned)(((long long)kr * p.ldb + n0 + wn + ((((ch >> 1) ^ key) << 1) | (ch & 1)) * 8) * 2);
    }
  }
  const unsigned aqs = (unsigned)(4 * p.lda * 2), ajs = (unsigned)(64 * p.lda * 2), bqs = (unsigned)(8 * p.ldb * 2), bhs = (unsigned)(32 * p.ldb * 2);
  const unsigned lds0 = (unsigned)(size_t)LDS_PTR(char, smem);
  const unsigned ldsA = lds0 + (unsigned)(w * 4) * 1024u, ldsB = lds0 + (unsigned)(BBASE + w * BWAVE);
  auto dma_a = [&](int slot, int j, int q) {
    lds_dma16u(rsA, avo[q >> 1] + ((unsigned)j * ajs + (unsigned)q * aqs), ldsA + (unsigned)(slot * ASLOT + q * 1024));
  };
  auto dma_b = [&](int slot, int u, int q) {
    lds_dma16u(rsB, bvo[q & 1] + ((unsigned)u * bhs + (unsigned)q * bqs), ldsB + (unsigned)(slot * BSLOT + q * 1024));
  };
  f4_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int jj = 0; jj < FN; ++jj) acc[i][jj] = f4_t{0.f, 0.f, 0.f, 0.f};
  const int ra0 = (8 * g + (t >> 2)) * 256 + (((t >> 2) | ((g & 1) << 2)) << 5) + (t & 3) * 8;
  const int rb0 = (8 * g + (t >> 2)) * 128 + ((((t >> 3) & 1) | ((g & 1) << 1)) << 5) + (t & 3) * 8;
  s4_t alo[FM], ahi[FM], b0lo[FN], b0hi[FN], b1lo[FN], b1hi[FN];
#pragma unroll
  for (int i = 0; i < FM; ++i) { alo[i] = s4_t{0, 0, 0, 0}; ahi[i] = s4_t{0, 0, 0, 0}; }
  // fragment i of K half h of the A image in `slot`: two transposing reads into the registers the fragment's last row has consumed
  auto read_a2 = [&](int slot, int h, int i, s4_t& lo, s4_t& hi) {
    const unsigned addr = lds0 + (unsigned)(slot * ASLOT) + (unsigned)((ra0 ^ (i << 5)) + h * 8192);
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "+v"(lo) : "v"(addr));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "+v"(hi) : "v"(addr));
  };
  auto read_a = [&](int slot, int h, auto ic) { constexpr int i = decltype(ic)::value; read_a2(slot, h, i, alo[i], ahi[i]); };
  auto read_b = [&](int slot, s4_t (&lo)[FN], s4_t (&hi)[FN]) {
#pragma unroll
    for (int jn = 0; jn < FN; ++jn) {
      const unsigned a2 = ldsB + (unsigned)(slot * BSLOT) + (unsigned)(rb0 ^ (jn << 5));
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo[jn]) : "v"(a2));
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:512" : "=v"(hi[jn]) : "v"(a2));
    }
  };
  // hand-counted wait tied to the registers the next MFMAs read
  auto wait_a2 = [&](auto n, s4_t& lo, s4_t& hi) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(lo), "+v"(hi) : "n"(decltype(n)::value)); };
  auto wait_a = [&](auto n, auto ic) { constexpr int i = decltype(ic)::value; wait_a2(n, alo[i], ahi[i]); };
  auto wait_b = [&](auto n, s4_t (&lo)[FN], s4_t (&hi)[FN]) {
    asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(lo[0]), "+v"(hi[0]), "+v"(lo[1]), "+v"(hi[1]), "+v"(lo[2]), "+v"(hi[2]), "+v"(lo[3]), "+v"(hi[3]) : "n"(decltype(n)::value));
  };
  // bias gradient (sum over tokens of dY) on the tn == 0 tiles: wave w sums fragments 2 w and 2 w + 1 (every wave holds all eight)
  constexpr bool do_cs = CS;
  float cs[2] = {0.f, 0.f};
  auto mma_row2 = [&](int i, const s4_t& al, const s4_t& ah, f4_t (&ac)[FN], s4_t (&blo)[FN], s4_t (&bhi)[FN]) {
    const s8_t fa = join_s4(al, ah);
#pragma unroll
    for (int jj = 0; jj < FN; ++jj)
      ac[jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8_t, join_s4(blo[jj], bhi[jj])), __builtin_bit_cast(bf8_t, fa), ac[jj], 0, 0, 0);
    if (do_cs && w == (i >> 1)) {
      const u4_t d = __builtin_bit_cast(u4_t, fa);
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) { s0 += __uint_as_float(d[e] << 16); s1 += __uint_as_float(d[e] & 0xFFFF0000u); }
      cs[i & 1] += s0 + s1;
    }
  };
  auto mma_row = [&](auto ic, s4_t (&blo)[FN], s4_t (&bhi)[FN]) { constexpr int i = decltype(ic)::value; mma_row2(i, alo[i], ahi[i], acc[i], blo, bhi); };
  // ---- prologue (issue order chosen so that every wait of the first step has the steady state's 8 younger pieces behind what it needs)
  const int nsteps = kt_end - kt_begin;
  const int jb = kt_begin;   // absolute K step of the slice's first step; B half u = 2 (jb + j) + h
#pragma unroll
  for (int q = 0; q < 4; ++q) dma_a(0, jb, q);
#pragma unroll
  for (int q = 0; q < 4; ++q) dma_b(0, 2 * jb, q);
#pragma unroll
  for (int q = 0; q < 4; ++q) dma_b(1, 2 * jb + 1, q);
  if (nsteps >= 2) {
#pragma unroll
    for (int q = 0; q < 4; ++q) dma_b(2, 2 * jb + 2, q);
#pragma unroll
    for (int q = 0; q < 4; ++q) dma_a(1, jb + 1, q);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  GTS(1);
  static_for<4>([&](auto ic) { read_a(0, 0, ic); });   // fragments 0 .. 3 of the first half; 4 .. 7 follow behind rows 0 .. 3
  read_b(0, b0lo, b0hi);
  auto nx3 = [](int s, int k) { s += k; return s >= 3 ? s - 3 : s; };
  // Reads in program order (steady state):  behind row 0: A(4), B_j^1 x 8 | rows 1 .. 3: A(5 .. 7) | rows 4 .. 7: A(8 .. 11) (second half, fragments 0 .. 3)
  // | row 8: A(12), B_{j+1}^0 x 8 | rows 9 .. 11: A(13 .. 15) | row 12: - | row 13: barrier, A(16), A(17) (next step) | rows 14, 15: A(18), A(19).
  // Younger reads at the wait in front of row g (each A = 2 reads): see the table in `wcnt`.
  // DMA: B_{j+1}^1 behind rows 1, 3, 5, 7 | B_{j+2}^0 behind rows 9 .. 12 | A_{j+2} behind rows 14, 15 (two pieces each): every wait leaves 8 pieces in flight.
  // MODE 0 steady | 2 second-to-last step (only B_{j+1}^1 left to fetch) | 3 last step
  auto step = [&](int j, int sa, int sb, auto mode_tag) {
    constexpr int MODE = decltype(mode_tag)::value;
    constexpr bool more = MODE < 3;
    const int sb1 = nx3(sb, 1), sb2 = nx3(sb, 2);
    const int ja = jb + j;
    if (MODE == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // B_j^1 has landed
    __builtin_amdgcn_sched_barrier(0);
    static_for<16>([&](auto rc) {
      constexpr int r = decltype(rc)::value, i = r & 7;
      // ---- wait for the row's fragment (and, at the head of a half, for its B fragments)
      constexpr int wc = (r == 0) ? 6 : (r <= 4) ? 14 : (r <= 8) ? 6 : (r <= 12) ? (more ? 14 : 6) : (r == 13) ? 4 : -1;
      if (r == 0) { wait_a(std::integral_constant<int, wc>{}, std::integral_constant<int, 0>{}); wait_b(std::integral_constant<int, wc>{}, b0lo, b0hi); }
      else if (r == 8) { wait_a(std::integral_constant<int, wc>{}, std::integral_constant<int, 0>{}); wait_b(std::integral_constant<int, wc>{}, b1lo, b1hi); }
      else if (wc >= 0) wait_a(std::integral_constant<int, wc>{}, std::integral_constant<int, i>{});
      if (r < 8) mma_row(std::integral_constant<int, i>{}, b0lo, b0hi); else mma_row(std::integral_constant<int, i>{}, b1lo, b1hi);
      // ---- the fragment for the row four rows ahead, into the registers consumed four rows ago
      if (r < 4) read_a(sa, 0, std::integral_constant<int, r + 4>{});                 // rows 4 .. 7: first half, fragments 4 .. 7
      else if (r < 8) read_a(sa, 1, std::integral_constant<int, r - 4>{});            // rows 8 .. 11: second half, fragments 0 .. 3
      else if (r < 12) read_a(sa, 1, std::integral_constant<int, r - 4>{});           // rows 12 .. 15: second half, fragments 4 .. 7
      if (r == 0) read_b(sb1, b1lo, b1hi);
      if (r == 8 && more) {
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                              // B_{j+1}^0 has landed (wave-private: no barrier)
        read_b(sb2, b0lo, b0hi);
      }
      if (r == 13) {
        if (MODE == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else if (MODE == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // own pieces of A_{j+1}
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                            // every read of A_j is done (the last was issued behind row 11)
        if (more) {
          __builtin_amdgcn_s_barrier();
          __builtin_amdgcn_sched_barrier(0);
          read_a(sa ^ 1, 0, std::integral_constant<int, 0>{}); read_a(sa ^ 1, 0, std::integral_constant<int, 1>{});
        }
      }
      if (r == 14 && more) read_a(sa ^ 1, 0, std::integral_constant<int, 2>{});
      if (r == 15 && more) read_a(sa ^ 1, 0, std::integral_constant<int, 3>{});
      // ---- DMA pieces
      if (more && r < 8 && (r & 1)) dma_b(sb, 2 * ja + 3, r >> 1);
      if (MODE == 0 && r >= 9 && r <= 12) dma_b(sb1, 2 * ja + 4, r - 9);
      if (MODE == 0 && r >= 14) { dma_a(sa, ja + 2, 2 * (r - 14)); dma_a(sa, ja + 2, 2 * (r - 14) + 1); }
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  int j = 0, sa = 0, sb = 0;
  for (; j < nsteps - 2; ++j, sa ^= 1, sb = nx3(sb, 2)) step(j, sa, sb, std::integral_constant<int, 0>{});
  if (nsteps >= 2) { step(j, sa, sb, std::integral_constant<int, 2>{}); ++j; sa ^= 1; sb = nx3(sb, 2); }
  step(j, sa, sb, std::integral_constant<int, 3>{});
  GTS(2);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (do_cs) {
#pragma unroll
    for (int e = 0; e < 2; ++e) { float v = cs[e]; v += __shfl_xor(v, 16); v += __shfl_xor(v, 32); cs[e] = v; }
  }
  // ---- epilogue: one slice -> dW += acc, db += column sums; several -> dense 128 x 256 fp32 slab + column-sum partials for the ordered fold
  constexpr int ESTR = WN + 4, LPR = WN / 4, RPP = 64 / LPR, EROWS = 32;
  float* ew = reinterpret_cast<float*>(smem + BBASE + w * BWAVE);
  if (fold.nsplit > 1) {
    if (do_cs && g == 0) {
      float* csm = fold.cs_slab + ((long long)split * fold.nslots + fold.slot) * 128;
      csm[(2 * w) * 16 + t] = cs[0];
      csm[(2 * w + 1) * 16 + t] = cs[1];
    }
    GemmArgs q = p;
    q.M = 128; q.N = 256; q.ldc = 256; q.bias = nullptr;
    float* mine = fold.slab + ((long long)split * fold.nslots + fold.slot) * (128 * 256);
    epilogue_rows<float, EPI_NONE, FM, FN, WM, EROWS, ESTR, LPR, RPP>(q, mine, acc, ew, 0, wn, lane, t, g);
    GTS(3);
    return;
  }
  if (do_cs && g == 0) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int m = m0 + (2 * w + e) * 16 + t;
      if (m < p.M) p.colsum[m] += cs[e];
    }
  }
  epilogue_rows<float, EPI_RESID, FM, FN, WM, EROWS, ESTR, LPR, RPP>(p, p.C, acc, ew, m0, n0 + wn, lane, t, g);   // dW = dW + acc (resid = C)
  GTS(3);
}
__global__ __launch_bounds__(256, 2) void gemm_dw_group_k2_kernel(DwGroupArgs ga) {
  const int wg = xcd_remap(blockIdx.x, ga.total_tiles * ga.nsplit);   // slice-major: the workgroups of a K slice are neighbours on an XCD
  const int split = wg / ga.total_tiles, tile_id = wg - split * ga.total_tiles;
  DwDesc d = ga.d[0];
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
  p.force_cfg = 0; p.split_stride = 0; p.colsum = d.db; p.dq_a = p.dq_b = nullptr; p.a_fmt = 0; p.aux_q8 = 0; p.q_out = nullptr;
  __shared__ __attribute__((aligned(16))) char smem[K2_LDS_BYTES];
  const DwFold2 fold{ga.slab, ga.cs_slab, ga.nsplit, tile_id, ga.total_tiles};
  if (d.db != nullptr && tn == 0) k2_tile_tn<true>(smem, p, tm, tn, split, kt_begin, kt_end, fold);
  else k2_tile_tn<false>(smem, p, tm, tn, split, kt_begin, kt_end, fold);
}
// ordered fold of the K slices (128 x 256 slabs): workgroup (tile, part) adds 128 / DWR_PARTS rows in slice order into dW; part 0 of the tn == 0 tiles the bias gradient
__global__ __launch_bounds__(256) void dw_group_reduce_k2_kernel(DwGroupArgs ga) {
  const int tile_id = blockIdx.x / DWR_PARTS, part = blockIdx.x % DWR_PARTS;
  DwDesc d = ga.d[0];
#pragma unroll
  for (int i = 1; i < DW_GROUP_MAX; ++i) if (i < ga.n && tile_id >= ga.d[i].tile0) d = ga.d[i];
  const int tile = tile_id - d.tile0;
  const int tm = tile / d.tiles_n, tn = tile - tm * d.tiles_n;
  const int m0 = tm * 128, n0 = tn * 256;
  const int c4 = threadIdx.x & 63, r0 = threadIdx.x >> 6;
  const long long sstride = (long long)ga.total_tiles * 128 * 256;
  const float* base = ga.slab + (long long)tile_id * 128 * 256;
  const int n = n0 + c4 * 4;
  constexpr int RPB = 128 / DWR_PARTS;
#pragma unroll
  for (int r = part * RPB + r0; r < part * RPB + RPB; r += 4) {
    const int m = m0 + r;
    if (m >= d.M || n >= d.N) continue;
    float* dst = d.dW + (long long)m * d.N + n;
    f4_t a = *reinterpret_cast<f4_t*>(dst);
    for (int sl = 0; sl < ga.nsplit; ++sl) a += *reinterpret_cast<const f4_t*>(base + sl * sstride + r * 256 + c4 * 4);
    *reinterpret_cast<f4_t*>(dst) = a;
  }
  if (part == 0 && tn == 0 && d.db != nullptr && threadIdx.x < 128) {
    const int m = m0 + threadIdx.x;
    if (m < d.M) {
      float a = d.db[m];
      for (int sl = 0; sl < ga.nsplit; ++sl) a += ga.cs_slab[((long long)sl * ga.total_tiles + tile_id) * 128 + threadIdx.x];
      d.db[m] = a;
    }
  }
}
// ga.d[i] filled by the caller except tiles_n / tile0; `slots` counts whole-CU workgroups (the one-workgroup kernel's unit): twice as many here
int gemm_k2_launch_dw(DwGroupArgs& ga, int count, int slots, float* workspace, long long ws_elems, hipStream_t st) {
  int tiles = 0;
  for (int i = 0; i < count; ++i) {
    DwDesc& d = ga.d[i];
    d.tiles_n = cdiv(d.N, 256); d.tile0 = tiles;
    tiles += cdiv(d.M, 128) * d.tiles_n;
  }
  for (int i = count; i < DW_GROUP_MAX; ++i) ga.d[i] = ga.d[0];
  ga.total_tiles = tiles;
  long long S = (2LL * slots) / tiles;
  if (S > ga.ktiles / 8) S = ga.ktiles / 8;
  const long long per_slice = (long long)tiles * (128 * 256 + 128);
  if (S > ws_elems / per_slice) S = ws_elems / per_slice;
  if (S < 1) S = 1;
  ga.ktiles_per_split = cdiv(ga.ktiles, S);
  ga.nsplit = cdiv(ga.ktiles, ga.ktiles_per_split);
  ga.slab = workspace; ga.cs_slab = workspace + (long long)ga.nsplit * tiles * 128 * 256;
  if (ga.nsplit > 1 && ws_elems < ga.nsplit * per_slice) { csmae_set_error("csmae_gemm_dw_group: workspace too small"); return CSMAE_ERR_ARG; }
  hipLaunchKernelGGL(gemm_dw_group_k2_kernel, dim3(tiles * ga.nsplit), dim3(256), 0, st, ga);
  if (ga.nsplit > 1) hipLaunchKernelGGL(dw_group_reduce_k2_kernel, dim3(tiles * DWR_PARTS), dim3(256), 0, st, ga);
  return csmae_check_launch("csmae_gemm_dw_group");
}
#else
int gemm_k2_launch_dw(DwGroupArgs&, int, int, float*, long long, hipStream_t) {
  csmae_set_error("csmae_gemm_dw_group: the two-workgroups-per-CU weight-gradient kernel is not in this build (-DCSMAE_K2_DW)");
  return CSMAE_ERR_UNSUPPORTED;
}
#endif
bool gemm_k2_dw_built() {
#ifdef CSMAE_K2_DW
  return true;
#else
  return false;
#endif
}
