// Archived prototype (round 1, negative result — see DESIGN.md §5): the register-staged GEMM kernels that were measured against the
// pipelined LDS-DMA kernel of csrc/gemm.hip and lost (one wave per SIMD with 128x128 wave tiles: -10 % per kernel; the same staging
// with eight waves: -4 %).  Not part of libcsmae_hip.so.  Builds on its own against the product source for re-measurement:
//     hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared tools/gemm_w4_prototype.hip cross-scale-mae_amd/csrc/api.hip -o build/libgemm_w4_proto.so
// (entry point csmae_proto_gemm_w4: K-contiguous A [M][K] and B [N][K], K % 64 == 0, the epilogues of csmae_gemm).
#include "../cross-scale-mae_amd/csrc/gemm.hip"

// ------------------------------------------------------------------------------------ bf16 MFMA, one wave per SIMD (prototype, cfg 6)
// 256x256 tile, FOUR waves with 128x128 wave tiles: 256 accumulator registers per lane, which only a lone wave per SIMD may hold (512
// registers).  One wave per SIMD reaches the full MFMA rate (tools/mfma_rate_probe.hip), and the larger wave tile needs 128 instead of
// 192 fragment reads per 64-wide K step.  A lone wave cannot hide the ~100-cycle issue of an LDS-DMA piece, so the operands come by
// `global_load_dwordx4` into staging registers (issued one K step ahead) and `ds_write_b128` into a double-buffered LDS image with
// the same swizzled K-contiguous layout as the 8-wave kernel.  K-contiguous A and B only.
// NW = 4: one wave per SIMD, 128x128 wave tiles (above).  NW = 8: the 8-wave / 128x64 shape of the pipelined kernel with the same
// register staging instead of LDS-DMA (8 loads + 8 `ds_write_b128` per wave and K step instead of 8 DMA pieces of ~100 issue cycles).
template <int NW>
__global__ __launch_bounds__(NW * 64, 1) void gemm_bf16_w4_kernel(GemmArgs p) {
  constexpr int BM = 256, BN = 256, WM = 128, NWN = NW / 2, WN = BN / NWN, FM = WM / 16, FN = WN / 16, IMG = 256 * 64 * 2;
  constexpr int NCH = 2048 / (NW * 64);     // 16-byte chunks per thread, operand and K step (8 or 4); chunk i = rows r0 + RS i
  constexpr int RS = 256 / NCH, NH = FN / 4;
  __shared__ __attribute__((aligned(16))) char smem[2 * 2 * IMG];   // [buffer][A | B], 128 KiB
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int t = lane & 15, g = lane >> 4;
  const int tiles = p.tiles_m * p.tiles_n;
  const int tile = xcd_remap(blockIdx.x, tiles);
  const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int wm = (w / NWN) * WM, wn = (w % NWN) * WN;
  const int nk = p.ktiles;
  void* Cptr = p.C;
  // staging: thread -> 16-byte chunk c of rows r0 + RS i (eight lanes cover one 128-byte row segment)
  const int c = threadIdx.x & 7, r0 = threadIdx.x >> 3;
  // Loads are unconditional and issued from inline assembly (no divergent control flow, and no compiler-inserted `vmcnt(0)` in the
  // pipelined loop: the waits are counted by hand).  Rows beyond M / N are clamped to the last row (their products are never stored).
  unsigned ao[NCH], bo[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    ao[i] = (unsigned)(((long long)min(m0 + r0 + RS * i, p.M - 1) * p.lda) * 2);
    bo[i] = (unsigned)(((long long)min(n0 + r0 + RS * i, p.N - 1) * p.ldb) * 2);
  }
  u4_t sa[NCH], sb[NCH];
  // (K % 64 == 0 on this path; steps past the last one re-read the last chunk: scalar clamp, no VALU select)
  auto load_a = [&](int kt, int i) {
    const unsigned ko = (unsigned)(min(kt, nk - 1) * 128 + c * 16);
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(sa[i]) : "v"(ao[i] + ko), "s"(p.A) : "memory");
  };
  auto load_b = [&](int kt, int i) {
    const unsigned ko = (unsigned)(min(kt, nk - 1) * 128 + c * 16);
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(sb[i]) : "v"(bo[i] + ko), "s"(p.B) : "memory");
  };
  auto gload1 = [&](int kt, int i) { load_a(kt, i); load_b(kt, i); };
  const int soff = r0 * 128 + ((c ^ (r0 & 7)) << 4);   // (rows r0 + RS i share r0's swizzle key: RS is a multiple of 8)
  const int ra0 = (wm + t) * 128 + ((g ^ (t & 7)) << 4), rb0 = (wn + t) * 128 + ((g ^ (t & 7)) << 4);
  auto read_a = [&](int buf, int h, int i) { return *reinterpret_cast<const s8_t*>(smem + buf * 2 * IMG + (ra0 ^ (h << 6)) + i * 2048); };
  auto read_b = [&](int buf, int h, int j) { return *reinterpret_cast<const s8_t*>(smem + buf * 2 * IMG + IMG + (rb0 ^ (h << 6)) + j * 2048); };
  f4_t acc[NH][FM][4];   // [64-column group][row fragment][column fragment]: each group is what the 64-column epilogues take
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[j / 4][i][j % 4] = f4_t{0.f, 0.f, 0.f, 0.f};
  // A K step is 16 "rows" (8 A fragments x 2 K halves) of FN MFMAs.  The A fragment of row r + 3 is read while row r issues (ring of
  // four 4-register buffers), the B fragments are double-buffered per half, the staging registers are written to the other LDS image
  // during the first rows of a step and refilled from global memory right away (two K steps ahead).  Every memory instruction sits
  // BETWEEN two MFMAs of its row: a lone wave has nobody else to fill the 16-cycle MFMA slots.  `sched_barrier` pins this order.
  constexpr int AD = 3;
  s8_t fa[4], fb[2][FN];
#pragma unroll
  for (int i = 0; i < NCH; ++i) gload1(0, i);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    *reinterpret_cast<u4_t*>(smem + soff + i * RS * 128) = sa[i];
    *reinterpret_cast<u4_t*>(smem + soff + IMG + i * RS * 128) = sb[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NCH; ++i) gload1(1, i);
#pragma unroll
  for (int j = 0; j < FN; ++j) fb[0][j] = read_b(0, 0, j);
#pragma unroll
  for (int r = 0; r < AD; ++r) fa[r] = read_a(0, 0, r);
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    char* sta = smem + (buf ^ 1) * 2 * IMG + soff;
    // (nothing below depends on whether a next step exists: past the last one the staged chunks repeat the last image and the
    // prefetched fragments are never used — no branches in the loop body)
    static_for<2 * FM>([&](auto rc) {
      constexpr int r = decltype(rc)::value, h = r / FM, i = r % FM;
      constexpr bool stg = h == 0 && i < NCH;   // this row stages chunk i
#define W4_MFMA(j) do { asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[(j) / 4][i][(j) % 4]) : "v"(fb[h][j]), "v"(fa[r & 3])); \
                        __builtin_amdgcn_sched_barrier(0); } while (0)
      W4_MFMA(0);
      // the A fragment of row r + AD (the next step's first rows come from the other image: r + AD >= 16 only after the barrier of row 9)
      if (r + AD < 2 * FM) fa[(r + AD) & 3] = read_a(buf, (r + AD) / FM, (r + AD) % FM);
      else fa[(r + AD) & 3] = read_a(buf ^ 1, 0, r + AD - 2 * FM);
      __builtin_amdgcn_sched_barrier(0);
      W4_MFMA(1);
      // B fragments of the other half: this step's second half during the first, the next step's first half after the barrier
      if (h == 0) { if (i < FN) fb[1][i] = read_b(buf, 1, i); }
      else if (r >= FM + 2) {
        constexpr int q = r - (FM + 2);                      // FN = 8: rows 10..15 fetch fragments 0..7 as 2, 2, 1, 1, 1, 1; FN = 4: rows 10..13 one each
        if (FN == 8) { if (q == 0) fb[0][0] = read_b(buf ^ 1, 0, 0); else if (q == 1) fb[0][2] = read_b(buf ^ 1, 0, 2); else fb[0][q + 2] = read_b(buf ^ 1, 0, q + 2); }
        else if (q < FN) fb[0][q] = read_b(buf ^ 1, 0, q);
      }
      __builtin_amdgcn_sched_barrier(0);
      W4_MFMA(2);
      if (FN == 8 && h == 1 && r == FM + 2) fb[0][1] = read_b(buf ^ 1, 0, 1);
      if (FN == 8 && h == 1 && r == FM + 3) fb[0][3] = read_b(buf ^ 1, 0, 3);
      if (stg) {   // chunk i of the next image: its two loads were issued one step ago (2 NCH - 2 younger loads are in flight)
        constexpr int cs = i < NCH ? i : 0;
        asm volatile("s_waitcnt vmcnt(%2)" : "+v"(sa[cs]), "+v"(sb[cs]) : "n"(2 * NCH - 2) : "memory");
        *reinterpret_cast<u4_t*>(sta + cs * RS * 128) = sa[cs];
      }
      __builtin_amdgcn_sched_barrier(0);
      W4_MFMA(3);
      if (stg) *reinterpret_cast<u4_t*>(sta + IMG + (i < NCH ? i : 0) * RS * 128) = sb[i < NCH ? i : 0];
      __builtin_amdgcn_sched_barrier(0);
      constexpr int ci = i < NCH ? i : 0;   // (the two staging registers of chunk i are refilled for step kt + 2)
      if (FN == 8) {
        W4_MFMA(4);
        if (stg) load_a(kt + 2, ci);
        __builtin_amdgcn_sched_barrier(0);
        W4_MFMA(5);
        if (stg) load_b(kt + 2, ci);
        __builtin_amdgcn_sched_barrier(0);
        W4_MFMA(6);
        W4_MFMA(7);
      } else if (stg) { load_a(kt + 2, ci); load_b(kt + 2, ci); __builtin_amdgcn_sched_barrier(0); }
#undef W4_MFMA
      if (r == FM + 1) __syncthreads();                      // next image complete and visible; nobody still reads the image before this one
      __builtin_amdgcn_sched_barrier(0);
    });
  }
  // The loads issued for the two steps past the end are still in flight and will write the staging registers: drain them while the
  // compiler still regards those registers as live (the loads come from inline assembly: it cannot know), before it reuses them.
#pragma unroll
  for (int i = 0; i < NCH; ++i) asm volatile("s_waitcnt vmcnt(0)" : "+v"(sa[i]), "+v"(sb[i]) :: "memory");
  if ((p.force_cfg & 16) && acc[0][0][0][0] != 123456.0f) return;  // tuning aid: main loop only
  __syncthreads();
  // the wave tile goes out in 64-column groups through the epilogues of the pipelined kernel
  constexpr int HN = 64, ESTR = HN + 4, LPR = HN / 4, RPP = 64 / LPR, EROWS = 32;
  static_assert(NW * EROWS * ESTR * 4 <= 4 * IMG, "epilogue strip must fit the LDS images");
  float* ew = reinterpret_cast<float*>(smem) + w * (EROWS * ESTR);
  // (two explicit calls, not a loop over the groups: a loop this large is not unrolled and the accumulators would move to scratch)
#define EPI_CALL(TC_, E_) do { epilogue_rows<TC_, E_, FM, 4, WM, EROWS, ESTR, LPR, RPP>(p, Cptr, acc[0], ew, m0 + wm, n0 + wn, lane, t, g); \
    if (NH == 2) epilogue_rows<TC_, E_, FM, 4, WM, EROWS, ESTR, LPR, RPP>(p, Cptr, acc[NH - 1], ew, m0 + wm, n0 + wn + HN, lane, t, g); } while (0)
#define EPI_CALL8(E_) do { epilogue_rows_bf16x8<E_, FM, 4, WM, EROWS, ESTR>(p, Cptr, acc[0], ew, m0 + wm, n0 + wn, lane, t, g); \
    if (NH == 2) epilogue_rows_bf16x8<E_, FM, 4, WM, EROWS, ESTR>(p, Cptr, acc[NH - 1], ew, m0 + wm, n0 + wn + HN, lane, t, g); } while (0)
  const bool wide = (p.ldc % 8 == 0) && (p.epi == EPI_NONE || p.ldaux % 8 == 0) && ((uintptr_t)p.aux % 16 == 0);
  if (p.c_dtype == CSMAE_BF16) {
    if (p.epi == EPI_RESID) EPI_CALL(bf16_t, EPI_RESID);
    else if (wide) { if (p.epi == EPI_GELU) EPI_CALL8(EPI_GELU); else if (p.epi == EPI_DGELU) EPI_CALL8(EPI_DGELU); else EPI_CALL8(EPI_NONE); }
    else if (p.epi == EPI_GELU) EPI_CALL(bf16_t, EPI_GELU); else if (p.epi == EPI_DGELU) EPI_CALL(bf16_t, EPI_DGELU);
    else EPI_CALL(bf16_t, EPI_NONE);
  } else {
    if (p.epi == EPI_GELU) EPI_CALL(float, EPI_GELU); else if (p.epi == EPI_DGELU) EPI_CALL(float, EPI_DGELU);
    else if (p.epi == EPI_RESID) EPI_CALL(float, EPI_RESID); else EPI_CALL(float, EPI_NONE);
  }
#undef EPI_CALL
#undef EPI_CALL8
}


extern "C" int csmae_proto_gemm_w4(int waves, long long M, long long N, long long K, const void* A, long long lda, const void* B, long long ldb,
                                   void* C, long long ldc, int c_dtype, const float* bias, int epilogue, void* aux, long long ldaux,
                                   const float* resid, long long ldr, void* stream) {
  CSMAE_REQUIRE((waves == 4 || waves == 8) && M > 0 && N > 0 && K > 0 && K % 64 == 0 && epilogue >= EPI_NONE && epilogue <= EPI_DGELU,
                "csmae_proto_gemm_w4: K-contiguous operands, whole 64-wide K steps, waves 4 or 8");
  GemmArgs p;
  p.force_cfg = 0; p.split_stride = 0; p.colsum = nullptr;
  p.A = A; p.B = B; p.C = C; p.bias = (epilogue >= EPI_DGELU) ? nullptr : bias; p.aux = aux; p.resid = resid;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldaux = ldaux; p.ldr = ldr;
  p.M = (int)M; p.N = (int)N; p.K = (int)K; p.c_dtype = c_dtype; p.epi = epilogue; p.splitk = 1;
  p.a_bytes = (unsigned)(M * lda * 2); p.b_bytes = (unsigned)(N * ldb * 2);
  p.ktiles = (int)(K / 64); p.ktiles_per_split = p.ktiles;
  p.tiles_m = cdiv(M, 256); p.tiles_n = cdiv(N, 256);
  dim3 grid(p.tiles_m * p.tiles_n);
  if (waves == 4) hipLaunchKernelGGL(gemm_bf16_w4_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(gemm_bf16_w4_kernel<8>, grid, dim3(512), 0, (hipStream_t)stream, p);
  return csmae_check_launch("csmae_proto_gemm_w4");
}
