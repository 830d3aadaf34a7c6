// Shared pieces of the GEMM family (gemm.hip: 256 x 256 one-workgroup-per-CU kernels, fp8, fp32; gemm_k2.hip: 128 x 256 two-workgroups-per-CU
// kernel): argument block, epilogue codes, LDS-DMA helpers and the row-segment epilogues.  Included by both translation units.
#pragma once
// GEMM family for the ViT encoder/decoder blocks (QKV / proj / fc1 / fc2 and their
// backward products).  C[M,N] = sum_k A(m,k) * B(k,n) with fused epilogues.
//
//   transA = 0 : A stored [M][K]  (K contiguous)      transA = 1 : A stored [K][M]
//   transB = 0 : B stored [N][K]  (K contiguous)      transB = 1 : B stored [K][N]
//
//   forward   y = x W^T        : transA=0, transB=0  (torch Linear weight is [out,in] = [N][K])
//   dX = dY W                  : transA=0, transB=1  (W is [K=out][N=in])
//   dW = dY^T X (reduce tokens): transA=1, transB=1  (dY is [K=tok][M=out], X is [K=tok][N=in])
//
// bf16 path (gfx950): 128x128x64 block tile, 4 waves (2x2), each wave 64x64 = 4x4 MFMA 16x16x32 tiles,
// operands staged HBM->LDS by `buffer_load ... lds` (16 B/lane, no VGPR round trip, OOB -> 0 gives free
// edge handling), double-buffered LDS, one barrier per K-tile.  K-contiguous operands are read with
// ds_read_b128 from an XOR-swizzled image (conflict free); K-strided operands are read with the gfx950
// transpose read ds_read_b64_tr_b16 from a 32-B-granule swizzled image, so no transposed copies of
// weights/activations are ever materialised.  MFMA operands are swapped (D = B_frag x A_frag) so each lane
// ends with 4 consecutive output columns -> 8/16-byte epilogue accesses.
// fp32 path: exact-fp32 FMA tile kernel used for the 1e-4 parity mode (not the throughput path).
#include "common.h"
#include <type_traits>
#include <cstdlib>
#include <utility>

template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

#define EPI_NONE 0    // C = acc + bias
#define EPI_GELU 1    // x = acc + bias ; C = gelu(x) ; aux = gelu'(x)
#define EPI_RESID 2   // C = acc + bias + resid (resid in C's dtype: the fp32 residual stream, or the bf16 one of throughput mode)
#define EPI_DGELU 3   // C = acc * aux   (aux = gelu'(x) saved by the forward epilogue)
#define EPI_ATOMIC 4  // C(fp32) += acc   (split-K, atomics)
#define EPI_SPLIT 5   // C(fp32)[split] = acc  (split-K partial slabs, reduced by dw_reduce_kernel: deterministic, no atomics)

struct GemmArgs {
  const void* A; const void* B; void* C; const float* bias; void* aux; const void* resid;   // resid has C's dtype
  long long lda, ldb, ldc, ldaux, ldr;
  int M, N, K;
  int c_dtype, epi, splitk, tiles_m, tiles_n, ktiles, ktiles_per_split;
  unsigned a_bytes, b_bytes;
  int force_cfg;
  long long split_stride;
  float* colsum;  // K-strided-A kernels only: slab [splitk][M] receiving sum_k A(m,k) (the bias gradient of nn.Linear)
  const float* dq_a; const float* dq_b;  // fp8 kernel only: device scalars, the operands' de-quantisation factors (acc *= dq_a * dq_b)
  int a_fmt;                             // fp8 kernel only: format of A (0 = e4m3, 1 = e5m2); B is e4m3
  int aux_q8;                            // GELU / DGELU epilogues: aux (gelu') is one byte per element (GP_Q8 code below) instead of C's dtype
  // fp8 mode, delayed scaling: the epilogue also emits its bf16 output as fp8 bytes for the GEMM that consumes it next (q_out [M][ldq]),
  // scaled with the amax this tensor had one step earlier (q_amax_prev: 64 partial maxima), records the new amax (q_amax_next: 64 slots)
  // and leaves the de-quantisation factor in q_dq — the separate quantisation pass (read 2 B + write 1 B per element) disappears.
  unsigned char* q_out; long long ldq; const float* q_amax_prev; float* q_amax_next; float* q_dq; int q_fmt;
};

// gelu'(x) lies in [-0.129, 1.129]: stored as the 8-bit code q = round(200 g + 26) (range [-0.13, 1.145], step 5e-3, |error| <= 2.5e-3 — the
// size of a bf16 rounding step at 1) it costs one byte instead of two in the two epilogues that are bound by their HBM bytes (fc1 writes h
// and gelu', fc2-backward reads gelu' and writes dpre: -25 % each).  Scale and offset put g = 0 and g = 1 — the saturated units, i.e. most of
// the MLP — exactly on codes 26 and 226 (one fma, exact for both), so they decode to exactly 0 and 1: no systematic bias in dpre.
#define GP_Q8_SCALE 200.0f
#define GP_Q8_OFF 26.0f
__device__ __forceinline__ unsigned gp_q8_pack4(f4_t g) {
  unsigned p = 0;   // (v_cvt_pk_u8_f32 rounds to nearest; + 0.5 / floor would be needed if it truncated — tests/test_ops_gpu.py pins the codes)
  p = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(g[0], GP_Q8_SCALE, GP_Q8_OFF), 0, p);
  p = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(g[1], GP_Q8_SCALE, GP_Q8_OFF), 1, p);
  p = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(g[2], GP_Q8_SCALE, GP_Q8_OFF), 2, p);
  p = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(g[3], GP_Q8_SCALE, GP_Q8_OFF), 3, p);
  return p;
}
__device__ __forceinline__ f4_t gp_q8_unpack4(unsigned p) {
  const f4_t q = {(float)(p & 0xffu), (float)((p >> 8) & 0xffu), (float)((p >> 16) & 0xffu), (float)(p >> 24)};
  return (q - GP_Q8_OFF) * (1.0f / GP_Q8_SCALE);
}

// the neighbouring lane's value (lane ^ 1) as ONE DPP move (quad_perm [1,0,3,2]); `__shfl_xor(x, 1)` is a ds_bpermute_b32 round trip through the LDS crossbar
__device__ __forceinline__ unsigned lane_xor1(unsigned x) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, true); }

// ------------------------------------------------------------------------------------ epilogue
template <typename TC>
__device__ __forceinline__ void epi_store4(const GemmArgs& p, int m, int n, f4_t v) {
  if (p.bias) { f4_t b = *reinterpret_cast<const f4_t*>(p.bias + n); v += b; }
  TC* c = reinterpret_cast<TC*>(p.C) + (long long)m * p.ldc + n;
  if (p.epi == EPI_GELU) {  // aux <- gelu'(pre) (what backward multiplies by), C <- gelu(pre)
    f4_t gp;
    const f4_t x = v;
    gelu_both4<TC>(x, v, gp);
    st4<TC>(reinterpret_cast<TC*>(p.aux) + (long long)m * p.ldaux + n, gp);
  } else if (p.epi == EPI_RESID) {
    v += ld4<TC>(reinterpret_cast<const TC*>(p.resid) + (long long)m * p.ldr + n);
  } else if (p.epi == EPI_DGELU) {
    v *= ld4<TC>(reinterpret_cast<const TC*>(p.aux) + (long long)m * p.ldaux + n);
  }
  st4<TC>(c, v);
}

__device__ __forceinline__ void epi_dispatch(const GemmArgs& p, int m, int n, f4_t v) {
  if (m >= p.M || n >= p.N) return;
  if (p.epi == EPI_ATOMIC) {
    float* c = reinterpret_cast<float*>(p.C) + (long long)m * p.ldc + n;
    unsafeAtomicAdd(c + 0, v[0]); unsafeAtomicAdd(c + 1, v[1]); unsafeAtomicAdd(c + 2, v[2]); unsafeAtomicAdd(c + 3, v[3]);
  } else if (p.c_dtype == CSMAE_BF16) {
    epi_store4<bf16_t>(p, m, n, v);
  } else {
    epi_store4<float>(p, m, n, v);
  }
}

// XCD-aware, bijective block remap: blocks that share the A row panel land on the same XCD/L2.
__device__ __forceinline__ int xcd_remap(int b, int nwg) {
  int xcd = b & 7, idx = b >> 3, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// ------------------------------------------------------------------------------------ bf16 MFMA
// Block tile BM x BN x 32, NW = (BM/WM)*(BN/WN) waves, each wave WM x WN (16x16x32 MFMA fragments).  Measured on MI355X the
// first (128x128x64, 2-stage) version of this kernel was bound by the HBM/L2 -> LDS path at ~6 TB/s (64 flop per staged byte,
// one K-tile of prefetch): so tiles are as large as the problem allows (256x256: 128 flop/B) and the LDS ring is 4 stages deep
// with counted `s_waitcnt vmcnt(N)` + raw s_barrier, keeping three K-tiles of `buffer_load ... lds` in flight across barriers.
#define OOB_OFF 0xFFFFFFF0u
#define GEMM_BK 32
__device__ __forceinline__ int tr_key(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }

// `buffer_load_dwordx4 ... lds` as inline assembly.  Through the builtin, LLVM's waitcnt pass knows that LDS is being written by
// VMEM and puts `s_waitcnt vmcnt(0)` in front of every later LDS read it cannot prove disjoint — which includes every
// ds_read_b64_tr_b16 (an intrinsic without memory operands): the K-strided kernels then waited for the K-tiles they had just
// prefetched.  The kernels below order DMA against LDS reads themselves (counted vmcnt + s_barrier), so the DMA is hidden from
// the compiler.  LDS destination = M0 + lane * 16; an out-of-range voffset returns zeros.
typedef int i4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i4_t make_rsrc(const void* ptr, unsigned bytes) {
  const unsigned long long a = (unsigned long long)ptr;
  return i4_t{(int)(a & 0xffffffffu), (int)((a >> 32) & 0xffffu), (int)bytes, 0x00020000};
}
__device__ __forceinline__ void lds_dma16(i4_t rsrc, unsigned voffset, const void* lds_dst) {
  const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(size_t)LDS_PTR(const char, lds_dst));
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(m0v), "v"(voffset), "s"(rsrc) : "memory");
}
// the same with the LDS destination as a (wave-uniform) LDS byte address: no generic -> LDS pointer cast (four scalar instructions per piece)
__device__ __forceinline__ void lds_dma16u(i4_t rsrc, unsigned voffset, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voffset), "s"(rsrc) : "memory");
}

template <bool TR, int W>   // staging descriptor of one 1-KiB DMA piece of an operand image
struct PieceDesc {
  unsigned off; int kidx; bool ok;
  __device__ __forceinline__ void init(int piece, int lane, long long base0, long long extent, long long ld) {
    if (!TR) {  // image [W rows][32 k] : 64-B rows, 16-B chunk swizzle c ^ ((row >> 1) & 3)
      int row = piece * 16 + (lane >> 2), c = (lane & 3) ^ ((row >> 1) & 3);
      long long gr = base0 + row;
      ok = gr < extent; kidx = c * 8;
      off = (unsigned)((gr * ld + c * 8) * 2);
    } else {    // image [32 rows k][W cols] : 2W-B rows, 32-B granule swizzle q ^ tr_key(row)
      constexpr int LPR = W / 8, RPP = 64 / LPR;
      int row = piece * RPP + lane / LPR, s16 = lane % LPR;
      int ch = ((((s16 >> 1) ^ tr_key(row)) << 1) | (s16 & 1));
      ok = true; kidx = row;
      off = (unsigned)(((long long)row * ld + base0 + ch * 8) * 2);
    }
  }
};

// buffer resources for the epilogues that address their tensors by 32-bit byte offsets (rows beyond the end are dropped by the range check)
typedef __amdgpu_buffer_rsrc_t brsrc_t;
typedef unsigned int u2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ brsrc_t buf_rsrc(const void* ptr, long long bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), 0, (int)(unsigned)bytes, 0x00020000);
}
// Row-segment epilogue.  vmcnt is one in-order counter for loads AND stores on gfx9/CDNA: a residual / aux load issued after a
// store cannot be consumed before that store has been acknowledged by memory (~2 us under load).  The first version of this
// epilogue interleaved "load, add, store" per fragment and spent 45 % of the GEMM time in those waits.  So every global load of
// a phase is issued before the phase's first store, and the stores then go out back to back.
template <typename TC, int EPI, int FM, int FN, int WM, int EROWS, int ESTR, int LPR, int RPP>
__device__ __forceinline__ void epilogue_rows(const GemmArgs& p, void* Cptr, f4_t (&acc)[FM][FN], float* ew, int mbase, int nbase, int lane, int t, int g) {
  constexpr int NPASS = EROWS / RPP, NPART = WM / EROWS;
  constexpr bool NEEDS_LOAD = EPI == EPI_RESID || EPI == EPI_DGELU;
  constexpr bool PRELOAD_ALL = false;  // (WM <= 64) would fit all residual/aux rows in registers but costs the occupancy of the 2-blocks/CU tile
  const int col = (lane % LPR) * 4, rsub = lane / LPR;
  const int gn = nbase + col;
  const bool colok = gn < p.N;
  const int gnc = colok ? gn : 0;           // clamped addresses: loads are unconditional (no divergent control flow), stores are predicated
  f4_t bias4 = {0.f, 0.f, 0.f, 0.f};
  if (p.bias) bias4 = *reinterpret_cast<const f4_t*>(p.bias + gnc);
  f4_t ld[NEEDS_LOAD ? (PRELOAD_ALL ? NPART * NPASS : NPASS) : 1];
  auto load_part = [&](int part, int slot0) {
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const int gm = min(mbase + part * EROWS + ps * RPP + rsub, p.M - 1);
      if (EPI == EPI_RESID) ld[slot0 + ps] = ld4<TC>(reinterpret_cast<const TC*>(p.resid) + (long long)gm * p.ldr + gnc);
      else if (p.aux_q8) ld[slot0 + ps] = gp_q8_unpack4(*reinterpret_cast<const unsigned*>(reinterpret_cast<const unsigned char*>(p.aux) + (long long)gm * p.ldaux + gnc));
      else ld[slot0 + ps] = ld4<TC>(reinterpret_cast<const TC*>(p.aux) + (long long)gm * p.ldaux + gnc);
    }
  };
  if (PRELOAD_ALL && NEEDS_LOAD) {
#pragma unroll
    for (int part = 0; part < NPART; ++part) load_part(part, part * NPASS);
  }
#pragma unroll
  for (int part = 0; part < NPART; ++part) {
    if (!PRELOAD_ALL && NEEDS_LOAD) load_part(part, 0);
#pragma unroll
    for (int ii = 0; ii < EROWS / 16; ++ii)
#pragma unroll
      for (int j = 0; j < FN; ++j)
        *reinterpret_cast<f4_t*>(ew + (ii * 16 + t) * ESTR + j * 16 + 4 * g) = acc[part * (EROWS / 16) + ii][j];
    // every load of this phase has been issued: retire them once, here, so that no wait is needed between the stores below
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) only
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const int gm = mbase + part * EROWS + ps * RPP + rsub;
      f4_t v = *reinterpret_cast<const f4_t*>(ew + (ps * RPP + rsub) * ESTR + col) + bias4;
      f4_t o = v;
      if (EPI == EPI_GELU) {  // o = gelu(v); v <- gelu'(v), the factor the backward epilogue multiplies by
        const f4_t x = v;
        gelu_both4<TC>(x, o, v);
      }
      if (EPI == EPI_RESID) o = v + ld[(PRELOAD_ALL ? part * NPASS : 0) + ps];
      if (EPI == EPI_DGELU) o = v * ld[(PRELOAD_ALL ? part * NPASS : 0) + ps];
      if (colok && gm < p.M) {
        if (EPI == EPI_GELU) {
          if (p.aux_q8) *reinterpret_cast<unsigned*>(reinterpret_cast<unsigned char*>(p.aux) + (long long)gm * p.ldaux + gn) = gp_q8_pack4(v);
          else st4<TC>(reinterpret_cast<TC*>(p.aux) + (long long)gm * p.ldaux + gn, v);
        }
        st4<TC>(reinterpret_cast<TC*>(Cptr) + (long long)gm * p.ldc + gn, o);
      }
    }
  }
}

// bf16 outputs, 8 columns per lane: 16-byte stores (one instruction covers 8 rows x 128 B instead of 4).  The epilogue of a
// 256x256 tile is bound by the store ISSUE rate of its CU, not by HBM: halving the number of store instructions is what counts.
template <int EPI, int FM, int FN, int WM, int EROWS, int ESTR>
__device__ __forceinline__ void epilogue_rows_bf16x8(const GemmArgs& p, void* Cptr, f4_t (&acc)[FM][FN], float* ew, int mbase, int nbase, int lane, int t, int g) {
  constexpr int LPR = FN * 16 / 8, RPP = 64 / LPR, NPASS = EROWS / RPP, NPART = WM / EROWS;
  constexpr bool NEEDS_LOAD = EPI == EPI_DGELU || EPI == EPI_RESID;
  const int col = (lane % LPR) * 8, rsub = lane / LPR;
  const int gn = nbase + col;
  const bool ok0 = gn < p.N, ok1 = gn + 4 < p.N;   // N % 4 == 0: a lane's 8 columns are valid as two groups of 4
  f4_t b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
  if (p.bias) { if (ok0) b0 = *reinterpret_cast<const f4_t*>(p.bias + gn); if (ok1) b1 = *reinterpret_cast<const f4_t*>(p.bias + gn + 4); }
  bf16_t* C = reinterpret_cast<bf16_t*>(Cptr);
  bf16_t* X = reinterpret_cast<bf16_t*>(p.aux);
  const bool emit = p.q_out != nullptr;
  const float qmax = p.q_fmt == 0 ? 448.0f : 57344.0f;
  float qscale = 1.f, qseen = 0.f;
  unsigned qbits = 0u;   // (the running amax as a bit pattern, see the emit block)
  if (emit) {
    const float am = wave_max64(p.q_amax_prev[lane]);
    qscale = am > 0.f ? qmax / am : 1.f;
    if (blockIdx.x == 0 && threadIdx.x == 0) p.q_dq[0] = am > 0.f ? am / qmax : 1.f;
  }
  const bf16_t* LS = EPI == EPI_RESID ? reinterpret_cast<const bf16_t*>(p.resid) : X;   // what the phase loads: bf16 residual stream / gelu'
  const long long lds_ = EPI == EPI_RESID ? p.ldr : p.ldaux;
  // Loads of part p + 1 are issued BEFORE the stores of part p and nothing waits for a store: vmcnt retires loads and stores in issue
  // order, so a `vmcnt(0)` in front of every part (the first version) made each part wait for the previous part's stores to be
  // acknowledged by memory — three store round trips, over half of the epilogue's 7 k clocks per tile.
  // (8-bit gelu' stores in 16-byte pieces, see the store loop: rows of aux 16-byte aligned, passes in pairs, all 16 columns of a lane pair inside N)
  const bool gp16 = EPI == EPI_GELU && p.aux_q8 && (NPASS % 2 == 0) && p.ldaux % 16 == 0 && ((uintptr_t)p.aux & 15) == 0;
  const bool pair_ok = nbase + ((lane % LPR) & ~1) * 8 + 16 <= p.N;
  uint2 gp_prev = make_uint2(0u, 0u);
  uint4 ld[2][NPASS];   // (dead code unless NEEDS_LOAD)
  auto load_part = [&](int part, uint4 (&dst)[NPASS]) {
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const int gm = min(mbase + part * EROWS + ps * RPP + rsub, p.M - 1);
      const bf16_t* src = LS + (long long)gm * lds_ + gn;
      dst[ps] = make_uint4(0, 0, 0, 0);
      if (EPI == EPI_DGELU && p.aux_q8) {   // one byte per element: 8 (4) bytes per lane
        const unsigned char* s8 = reinterpret_cast<const unsigned char*>(p.aux) + (long long)gm * p.ldaux + gn;
        if (ok1) { const uint2 h = *reinterpret_cast<const uint2*>(s8); dst[ps].x = h.x; dst[ps].y = h.y; }
        else if (ok0) dst[ps].x = *reinterpret_cast<const unsigned*>(s8);
      }
      else if (ok1) dst[ps] = *reinterpret_cast<const uint4*>(src);
      else if (ok0) { const uint2 h = *reinterpret_cast<const uint2*>(src); dst[ps].x = h.x; dst[ps].y = h.y; }
    }
  };
  if (NEEDS_LOAD) load_part(0, ld[0]);
#pragma unroll
  for (int part = 0; part < NPART; ++part) {
#pragma unroll
    for (int ii = 0; ii < EROWS / 16; ++ii)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        *reinterpret_cast<f4_t*>(ew + (ii * 16 + t) * ESTR + j * 16 + 4 * g) = acc[part * (EROWS / 16) + ii][j];
      }
    if (NEEDS_LOAD && part + 1 < NPART) load_part(part + 1, ld[(part + 1) & 1]);
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const int gm = mbase + part * EROWS + ps * RPP + rsub;
      const float* src = ew + (ps * RPP + rsub) * ESTR + col;
      f4_t v0 = *reinterpret_cast<const f4_t*>(src) + b0, v1 = *reinterpret_cast<const f4_t*>(src + 4) + b1;
      f4_t o0 = v0, o1 = v1;
#ifndef GEMM_EPI_ABL
#define GEMM_EPI_ABL 0   // epilogue ablations, compile-time only (tools/gemm_variants.sh): 1 no GELU arithmetic | 2 no gelu' store | 4 no C store
#endif
      if (EPI == EPI_GELU && !(GEMM_EPI_ABL & 1)) { const f4_t x0 = v0, x1 = v1; gelu_both4<bf16_t>(x0, o0, v0); gelu_both4<bf16_t>(x1, o1, v1); }
      if (EPI == EPI_DGELU || EPI == EPI_RESID) {
        const uint4 a = ld[part & 1][ps];
        f4_t a0 = f4_t{__uint_as_float(a.x << 16), __uint_as_float(a.x & 0xffff0000u), __uint_as_float(a.y << 16), __uint_as_float(a.y & 0xffff0000u)};
        f4_t a1 = f4_t{__uint_as_float(a.z << 16), __uint_as_float(a.z & 0xffff0000u), __uint_as_float(a.w << 16), __uint_as_float(a.w & 0xffff0000u)};
        if (EPI == EPI_DGELU && p.aux_q8) { a0 = gp_q8_unpack4(a.x); a1 = gp_q8_unpack4(a.y); }
        if (EPI == EPI_DGELU) { o0 = v0 * a0; o1 = v1 * a1; } else { o0 = v0 + a0; o1 = v1 + a1; }
      }
      if (emit && gm < p.M && ok0) {   // the fp8 copy of this row segment (8 or 4 columns)
        f4_t q0 = o0 * qscale, q1 = o1 * qscale;
        // the running maximum on bit patterns (common.h abs_bits: a NaN / Inf makes the recorded amax +Inf — the next step's scale is 0 x Inf = NaN, the loss
        // gate trips); the clamp as one v_med3 per element, NaN put back on the rare path (NaN is not clamped away)
        unsigned mb = umax3(abs_bits(o0[0]), abs_bits(o0[1]), umax3(abs_bits(o0[2]), abs_bits(o0[3]), 0u));
        if (ok1) mb = umax3(mb, umax3(abs_bits(o1[0]), abs_bits(o1[1]), abs_bits(o1[2])), abs_bits(o1[3]));
        qbits = max(qbits, mb);
        const f4_t u0 = q0, u1 = q1;
#pragma unroll
        for (int k = 0; k < 4; ++k) { q0[k] = __builtin_amdgcn_fmed3f(u0[k], -qmax, qmax); q1[k] = __builtin_amdgcn_fmed3f(u1[k], -qmax, qmax); }
        if (__builtin_amdgcn_ballot_w64(mb >= 0x7f800000u) != 0ull) {   // (wave-uniform: skipped as a whole when every value is finite)
#pragma unroll
          for (int k = 0; k < 4; ++k) { q0[k] = u0[k] != u0[k] ? u0[k] : q0[k]; q1[k] = u1[k] != u1[k] ? u1[k] : q1[k]; }
        }
        int w0 = 0, w1 = 0;
        if (p.q_fmt == 0) {
          w0 = __builtin_amdgcn_cvt_pk_fp8_f32(q0[0], q0[1], w0, false); w0 = __builtin_amdgcn_cvt_pk_fp8_f32(q0[2], q0[3], w0, true);
          w1 = __builtin_amdgcn_cvt_pk_fp8_f32(q1[0], q1[1], w1, false); w1 = __builtin_amdgcn_cvt_pk_fp8_f32(q1[2], q1[3], w1, true);
        } else {
          w0 = __builtin_amdgcn_cvt_pk_bf8_f32(q0[0], q0[1], w0, false); w0 = __builtin_amdgcn_cvt_pk_bf8_f32(q0[2], q0[3], w0, true);
          w1 = __builtin_amdgcn_cvt_pk_bf8_f32(q1[0], q1[1], w1, false); w1 = __builtin_amdgcn_cvt_pk_bf8_f32(q1[2], q1[3], w1, true);
        }
        unsigned char* qd = p.q_out + (long long)gm * p.ldq + gn;
        if (ok1) *reinterpret_cast<uint2*>(qd) = make_uint2((unsigned)w0, (unsigned)w1);
        else *reinterpret_cast<unsigned*>(qd) = (unsigned)w0;
      }
      if (EPI == EPI_GELU && p.aux_q8 && !(GEMM_EPI_ABL & 2)) {
        // 8-bit gelu': a lane's 8 columns are 8 bytes, and this epilogue is bound by the NUMBER of store instructions its CU issues.
        // Neighbouring lanes (columns 16k.. and 16k+8.. of one row) trade the codes of two consecutive passes, so that the even
        // lane stores 16 bytes of the first pass's row and the odd lane 16 bytes of the second's: one gelu' store per two passes.
        const uint2 mine = make_uint2(gp_q8_pack4(v0), gp_q8_pack4(v1));
        unsigned char* a8 = reinterpret_cast<unsigned char*>(p.aux);
        auto st8 = [&](int row, uint2 v) {
          if (row >= p.M) return;
          unsigned char* d8 = a8 + (long long)row * p.ldaux + gn;
          if (ok1) *reinterpret_cast<uint2*>(d8) = v;
          else if (ok0) *reinterpret_cast<unsigned*>(d8) = v.x;
        };
        if (!gp16) st8(gm, mine);
        else if ((ps & 1) == 0) gp_prev = mine;
        else {
          const bool odd = lane & 1;
          const uint2 send = odd ? gp_prev : mine;
          uint2 recv;
          recv.x = lane_xor1(send.x); recv.y = lane_xor1(send.y);
          if (!pair_ok) { st8(gm - RPP, gp_prev); st8(gm, mine); }
          else {   // ONE store instruction for the lane pair's two rows (an even-lane store and an odd-lane store, each half empty, cost what two 8-byte stores cost)
            const int row = odd ? gm : gm - RPP;
            const uint4 val = odd ? make_uint4(recv.x, recv.y, mine.x, mine.y) : make_uint4(gp_prev.x, gp_prev.y, recv.x, recv.y);
            if (row < p.M) *reinterpret_cast<uint4*>(a8 + (long long)row * p.ldaux + (odd ? gn - 8 : gn)) = val;
          }
        }
      }
      if (gm < p.M && C != nullptr && !(GEMM_EPI_ABL & 4)) {   // (C == null: the fused fp8 copy is the only output — csmae_gemm_fp8 in fp8 mode, where nothing reads the bf16 tensor)
        if (ok1) {
          if (EPI == EPI_GELU && !p.aux_q8) *reinterpret_cast<uint4*>(X + (long long)gm * p.ldaux + gn) = make_uint4(pack2bf(v0[0], v0[1]), pack2bf(v0[2], v0[3]), pack2bf(v1[0], v1[1]), pack2bf(v1[2], v1[3]));
          *reinterpret_cast<uint4*>(C + (long long)gm * p.ldc + gn) = make_uint4(pack2bf(o0[0], o0[1]), pack2bf(o0[2], o0[3]), pack2bf(o1[0], o1[1]), pack2bf(o1[2], o1[3]));
        } else if (ok0) {
          if (EPI == EPI_GELU && !p.aux_q8) st4<bf16_t>(X + (long long)gm * p.ldaux + gn, v0);
          st4<bf16_t>(C + (long long)gm * p.ldc + gn, o0);
        }
      }
    }
  }
  if (emit) {   // this wave's share of the new amax: one atomic per wave, spread over the 64 slots
    qseen = wave_max64(amax_of_bits(qbits));
    if (lane == 0) amax_publish(p.q_amax_next + ((blockIdx.x * 8 + (threadIdx.x >> 6)) & 63), qseen);
  }
}

// The same epilogue with buffer addressing (round 3).  The epilogues are bound by their VALU instruction count (tools/epi_abl.py: the
// fc1 epilogue's ~18 k clocks are its ~18 operations per element x 128 elements per lane x two waves per SIMD x 4 clocks), and a good part
// of what is not arithmetic was addressing: a 64-bit multiply-add chain per load / store and per pass, a row-bound compare with an
// exec-mask dance around every store.  Here every tensor is a buffer resource whose size ends at row M: a lane's byte offset advances by ONE
// 32-bit add per pass, rows beyond M are dropped by the hardware's range check, columns beyond N are marked out of range once per tile.
// Requirements (checked by the caller): N % 8 == 0 (a lane's 8 columns are valid or not as a whole), (M + 256) rows of every tensor < 4 GiB,
// no fused fp8 copy.  Same arithmetic, same bytes as epilogue_rows_bf16x8.
template <int EPI, int FM, int FN, int WM, int EROWS, int ESTR>
__device__ __forceinline__ void epilogue_rows_bf16x8b(const GemmArgs& p, f4_t (&acc)[FM][FN], float* ew, int mbase, int nbase, int lane, int t, int g) {
  constexpr int LPR = FN * 16 / 8, RPP = 64 / LPR, NPASS = EROWS / RPP, NPART = WM / EROWS;
  constexpr bool NEEDS_LOAD = EPI == EPI_DGELU || EPI == EPI_RESID;
  const int col = (lane % LPR) * 8, rsub = lane / LPR;
  const int gn = nbase + col;
  const bool colok = gn < p.N;
  f4_t b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
  if (p.bias && colok) { b0 = *reinterpret_cast<const f4_t*>(p.bias + gn); b1 = *reinterpret_cast<const f4_t*>(p.bias + gn + 4); }
  const int row0 = mbase + rsub;
  // byte offsets of the lane's 8 columns in row `row0` of C / the tensor a part loads (residual stream or gelu') / the 8-bit gelu' codes; one add per pass
  const brsrc_t rsC = buf_rsrc(p.C, (long long)p.M * p.ldc * 2);
  unsigned offC = colok ? (unsigned)(((long long)row0 * p.ldc + gn) * 2) : OOB_OFF;
  const unsigned stepC = colok ? (unsigned)(RPP * p.ldc * 2) : 0u;   // (a lane beyond N stays out of range: its offset must not wrap back into the tensor)
  const bool q8 = p.aux_q8 != 0;
  const long long lds_ = EPI == EPI_RESID ? p.ldr : p.ldaux;
  const int esz = (EPI != EPI_RESID && q8) ? 1 : 2;                         // bytes per element of the loaded / aux tensor
  const void* lptr = EPI == EPI_RESID ? p.resid : p.aux;
  const brsrc_t rsL = buf_rsrc((EPI == EPI_NONE) ? p.C : lptr, (EPI == EPI_NONE) ? 0 : (long long)p.M * lds_ * esz);
  unsigned offL = colok ? (unsigned)(((long long)row0 * lds_ + gn) * esz) : OOB_OFF;
  const unsigned stepL = colok ? (unsigned)(RPP * lds_ * esz) : 0u;
  // (8-bit gelu' stores in 16-byte pieces, see the store loop: rows of aux 16-byte aligned, passes in pairs, all 16 columns of a lane pair inside N)
  const bool gp16 = EPI == EPI_GELU && q8 && (NPASS % 2 == 0) && p.ldaux % 16 == 0 && ((uintptr_t)p.aux & 15) == 0;
  const bool pair_ok = nbase + ((lane % LPR) & ~1) * 8 + 16 <= p.N;
  const bool odd = lane & 1;
  uint2 gp_prev = make_uint2(0u, 0u);
  unsigned offL_prev = offL;
  uint4 ld[2][NPASS];   // (dead code unless NEEDS_LOAD)
  auto load_part = [&](uint4 (&dst)[NPASS]) {   // the loads of the NEXT part: offL runs ahead of the stores by one part
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      if (EPI == EPI_DGELU && q8) { const uint2 h = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rsL, offL, 0, 0)); dst[ps] = make_uint4(h.x, h.y, 0u, 0u); }
      else dst[ps] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsL, offL, 0, 0));
      offL += stepL;
    }
  };
  if (NEEDS_LOAD) load_part(ld[0]);
#pragma unroll
  for (int part = 0; part < NPART; ++part) {
#pragma unroll
    for (int ii = 0; ii < EROWS / 16; ++ii)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        *reinterpret_cast<f4_t*>(ew + (ii * 16 + t) * ESTR + j * 16 + 4 * g) = acc[part * (EROWS / 16) + ii][j];
      }
    if (NEEDS_LOAD && part + 1 < NPART) load_part(ld[(part + 1) & 1]);
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const float* src = ew + (ps * RPP + rsub) * ESTR + col;
      f4_t v0 = *reinterpret_cast<const f4_t*>(src) + b0, v1 = *reinterpret_cast<const f4_t*>(src + 4) + b1;
      f4_t o0 = v0, o1 = v1;
      if (EPI == EPI_GELU && !(GEMM_EPI_ABL & 1)) { const f4_t x0 = v0, x1 = v1; gelu_both4<bf16_t>(x0, o0, v0); gelu_both4<bf16_t>(x1, o1, v1); }
      if (EPI == EPI_DGELU || EPI == EPI_RESID) {
        const uint4 a = ld[part & 1][ps];
        f4_t a0 = f4_t{__uint_as_float(a.x << 16), __uint_as_float(a.x & 0xffff0000u), __uint_as_float(a.y << 16), __uint_as_float(a.y & 0xffff0000u)};
        f4_t a1 = f4_t{__uint_as_float(a.z << 16), __uint_as_float(a.z & 0xffff0000u), __uint_as_float(a.w << 16), __uint_as_float(a.w & 0xffff0000u)};
        if (EPI == EPI_DGELU && q8) { a0 = gp_q8_unpack4(a.x); a1 = gp_q8_unpack4(a.y); }
        if (EPI == EPI_DGELU) { o0 = v0 * a0; o1 = v1 * a1; } else { o0 = v0 + a0; o1 = v1 + a1; }
      }
      if (EPI == EPI_GELU && !(GEMM_EPI_ABL & 2)) {
        if (!q8) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4_t, make_uint4(pack2bf(v0[0], v0[1]), pack2bf(v0[2], v0[3]), pack2bf(v1[0], v1[1]), pack2bf(v1[2], v1[3]))), rsL, offL, 0, 0);
        else {
          // a lane's 8 codes are 8 bytes; neighbouring lanes trade the codes of two consecutive passes so that each stores 16 bytes of ONE row
          const uint2 mine = make_uint2(gp_q8_pack4(v0), gp_q8_pack4(v1));
          if (!gp16) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2_t, mine), rsL, offL, 0, 0);
          else if ((ps & 1) == 0) { gp_prev = mine; offL_prev = offL; }
          else {
            const uint2 send = odd ? gp_prev : mine;
            uint2 recv;
            recv.x = lane_xor1(send.x); recv.y = lane_xor1(send.y);
            if (!pair_ok) { __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2_t, gp_prev), rsL, offL_prev, 0, 0); __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2_t, mine), rsL, offL, 0, 0); }
            else {
              const uint4 val = odd ? make_uint4(recv.x, recv.y, mine.x, mine.y) : make_uint4(gp_prev.x, gp_prev.y, recv.x, recv.y);
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4_t, val), rsL, odd ? offL - 8u : offL_prev, 0, 0);
            }
          }
        }
        offL += stepL;
      }
      if (!(GEMM_EPI_ABL & 4))
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4_t, make_uint4(pack2bf(o0[0], o0[1]), pack2bf(o0[2], o0[3]), pack2bf(o1[0], o1[1]), pack2bf(o1[2], o1[3]))), rsC, offC, 0, 0);
      offC += stepC;
    }
  }
}

// bf16 output with nothing but the bias to add (qkv forward, the dX products of fc1 / qkv / proj): the tile crosses the strip as bf16,
// not as fp32.  A 256 x 256 tile's epilogue (7 k clocks at K = 512 .. 768, a quarter of the tile) spent 3.3 k of them pushing 256 KiB
// of fp32 accumulators through ds_write_b128 (13 clocks per KiB); rounding first halves the bytes written and read and leaves the
// row-segment pass with nothing to compute.  The arithmetic is unchanged (acc + bias in fp32, one rounding): bit-identical outputs.
template <int FM, int FN, int WM>
__device__ __forceinline__ void epilogue_rows_bf16_plain(const GemmArgs& p, void* Cptr, f4_t (&acc)[FM][FN], char* ew, int mbase, int nbase, int lane, int t, int g) {
  constexpr int WN = FN * 16, ROWB = WN * 2 + 8, EROWS = (WM % 64 == 0) ? 64 : 32, NPART = WM / EROWS, LPR = WN / 8, RPP = 64 / LPR, NPASS = EROWS / RPP;
  static_assert(WM % EROWS == 0 && EROWS * ROWB <= 32 * (WN + 4) * 4, "the bf16 strip reuses the fp32 strip's bytes");
  f4_t bj[FN];
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int n = nbase + j * 16 + 4 * g;
    bj[j] = (p.bias && n < p.N) ? *reinterpret_cast<const f4_t*>(p.bias + n) : f4_t{0.f, 0.f, 0.f, 0.f};
  }
  const int col = (lane % LPR) * 8, rsub = lane / LPR;
  const int gn = nbase + col;
  const bool ok0 = gn < p.N, ok1 = gn + 4 < p.N;
  bf16_t* C = reinterpret_cast<bf16_t*>(Cptr);
  // (buffer addressing — one add per pass, no row compare — measured 0.17 ms per step SLOWER for this epilogue, which has no arithmetic to
  // speak of and is bound by store issue: it pays in the VALU-bound row-segment epilogues only, epilogue_rows_bf16x8b)
#pragma unroll
  for (int part = 0; part < NPART; ++part) {
#pragma unroll
    for (int ii = 0; ii < EROWS / 16; ++ii)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        f4_t v = acc[part * (EROWS / 16) + ii][j];
        v += bj[j];
        *reinterpret_cast<uint2*>(ew + (ii * 16 + t) * ROWB + (j * 16 + 4 * g) * 2) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
      }
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const int r = ps * RPP + rsub, gm = mbase + part * EROWS + r;
      const uint4 v = *reinterpret_cast<const uint4*>(ew + r * ROWB + col * 2);
      if (gm < p.M) {
        if (ok1) *reinterpret_cast<uint4*>(C + (long long)gm * p.ldc + gn) = v;
        else if (ok0) *reinterpret_cast<uint2*>(C + (long long)gm * p.ldc + gn) = make_uint2(v.x, v.y);
      }
    }
  }
}

// phase timestamps of one workgroup (tuning aid, only in -DGEMM_TIMING builds: tools/gemm_phase_timing.py)
#ifdef GEMM_TIMING
static __device__ unsigned long long g_gemm_ts[8];   // one per translation unit (no relocatable device code): csmae_debug_gemm_ts / csmae_debug_k2_ts
#ifndef GTS_BLOCK
#define GTS_BLOCK 300
#endif
#define GTS(i) do { if (blockIdx.x == GTS_BLOCK && threadIdx.x == 0) g_gemm_ts[i] = __builtin_readcyclecounter(); } while (0)
#else
#define GTS(i)
#endif
// grouped weight-gradient launches (csmae_gemm_dw_group): descriptor table passed by value
#define DW_GROUP_MAX 8
struct DwDesc { const void* dY; const void* X; float* dW; float* db; int M, N; long long ldy, ldx; int tiles_n, tile0; };
struct DwGroupArgs {
  DwDesc d[DW_GROUP_MAX];
  int n, K, ktiles, ktiles_per_split, nsplit, total_tiles, force_cfg;
  float* slab; float* cs_slab;
};
#define DWR_PARTS 16
int gemm_k2_launch_dw(DwGroupArgs& ga, int count, int slots, float* workspace, long long ws_elems, hipStream_t st);   // gemm_k2.hip: 128 x 256 tiles, two workgroups per CU
// host-side hooks shared by the two translation units
int gemm_force_cfg();                 // csmae_gemm_force_tile's value (-1 = heuristic)
bool gemm_k2_nn_wanted(int epilogue, long long K, long long N, long long M);   // policy of csmae_gemm_k2_mode (gemm.hip)
bool gemm_k2_nt_wanted(int epilogue, long long K, long long N, long long M);
int gemm_k2_launch_nn(const GemmArgs& p, hipStream_t st);   // gemm_bf16_k2_kernel<true> (gemm_k2.hip)
int gemm_core(int dtype, int transA, int transB, long long M, long long N, long long K, const void* A, long long lda, const void* B, long long ldb,
              void* C, long long ldc, int c_dtype, const float* bias, int epilogue, void* aux, long long ldaux, const void* resid, long long ldr,
              int splitk, void* stream);
