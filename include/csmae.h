/* libcsmae_hip — C ABI of the MI355X (gfx950) Cross-Scale MAE pre-training hot path.
 *
 * The reference (aicip/Cross-Scale-MAE) is pure Python and has NO native/FFI boundary of its own
 * (SURVEY §8b): its "operators" are ATen/timm calls inside `models_mae/*.py` and `engine_pretrain.py`.
 * Each entry point below therefore cites the reference expression it replaces (paths relative to the
 * reference root) rather than a pre-existing FFI symbol.  INTEGRATION.md shows the ctypes binding a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous memory owned by the caller (PyTorch allocates);
 *     kernels never allocate, free or retain pointers;
 *   - `stream` is a hipStream_t; calls only enqueue work and return immediately (no implicit sync);
 *   - return 0 on success, negative csmae_status otherwise; csmae_last_error() gives a thread-local text;
 *   - dtype: CSMAE_F32 = 0 (exact-fp32 parity mode), CSMAE_BF16 = 1 (MFMA throughput mode; raw bf16 bits);
 *   - row "views": storage_row(r) = (r / group) * gstride + off + r % group  (the `[:, 1:, :]` slices);
 *   - scalars that change every step (crop box, upstream gradient, AdamW hyper-parameters) are read from
 *     device memory so a captured hipGraph of the step stays valid.
 */
#ifndef CSMAE_H
#define CSMAE_H
#ifdef __cplusplus
extern "C" {
#endif

#define CSMAE_F32 0
#define CSMAE_BF16 1

enum csmae_status { CSMAE_OK = 0, CSMAE_ERR_ARG = -1, CSMAE_ERR_LAUNCH = -2, CSMAE_ERR_UNSUPPORTED = -3 };
enum csmae_epilogue { CSMAE_EPI_NONE = 0, CSMAE_EPI_GELU = 1, CSMAE_EPI_RESID = 2, CSMAE_EPI_DGELU = 3, CSMAE_EPI_ATOMIC = 4, CSMAE_EPI_SPLIT = 5,
                      CSMAE_EPI_GELU_Q8 = 6, CSMAE_EPI_DGELU_Q8 = 7 /* bf16 path: aux = gelu'(x) as one byte per element, q = round(200 g + 26): g = 0 and 1 are codes 26 and 226 exactly */ };
enum csmae_loss { CSMAE_LOSS_MSE = 0, CSMAE_LOSS_L2 = 1, CSMAE_LOSS_MAE = 2, CSMAE_LOSS_L1 = 3, CSMAE_LOSS_BCE = 4 };

const char* csmae_last_error(void);
int csmae_abi_version(void);
const char* csmae_source_hash(void);   /* sha256 of the kernel sources the library was built from (tools/csrc_hash.py): profiles are keyed to it */

/* ---- completion event of the NEXT kernel launch of the calling host thread (ABI version 4).  The reference hands work between streams through
 * events that torch records BEHIND a kernel (autograd's stream hand-offs; DistributedDataParallel's bucket hooks, main_pretrain.py:417-421);
 * a recorded event is a marker packet that costs the recording stream 3-5 us.  csmae_next_launch_event(ev) makes the next csmae_gemm /
 * csmae_gemm_fp8 (pipelined kernels) / csmae_attn_bwd (bf16) launch carry `ev` (a hipEvent_t) as its dispatch packet's completion signal
 * instead; csmae_flush_launch_event(stream) records it the plain way when that launch did not take it (other kernels), nullptr clears it.  */
int csmae_next_launch_event(void* hip_event);
int csmae_flush_launch_event(void* stream);

/* ---- dense contractions: nn.Linear of timm Block / decoder_embed / decoder_pred / predictor and their backward
 * (timm 0.4.12 Attention.qkv/.proj, Mlp.fc1/.fc2 — call sites models_mae/MAE_ViT_Baseline.py:160-188,270,295;
 *  models_mae/MLP.py:6,9).  C[M,N] = sum_k A(m,k) B(k,n); transX = 0: K contiguous ([M,K] / [N,K]); 1: K strided ([K,M] / [K,N]).
 *  epilogue: NONE (+bias) | GELU (x = acc + bias: C = gelu(x), aux = gelu'(x)) | RESID (C = acc + bias + resid; both in c_dtype:
 *            the fp32 residual stream, or the bf16 one of throughput mode) |
 *            DGELU (C = acc * aux) | ATOMIC (fp32 C += acc) | SPLIT (fp32 split-K slabs, see csmae_gemm_dw). */
int csmae_gemm(int dtype, int transA, int transB, long long M, long long N, long long K,
               const void* A, long long lda, const void* B, long long ldb, void* C, long long ldc, int c_dtype,
               const float* bias, int epilogue, void* aux, long long ldaux, const void* resid /* in C's dtype */, long long ldr,
               int splitk, void* stream);

/* ---- the same forward product y = x W^T (+ epilogue) with the weight given as its K-SLAB mirror (ABI version 6): Wk[K/32][slab_rows][32] bf16,
 * Wk[(k / 32) * slab_rows * 32 + n * 32 + k % 32] = W[n][k] — written by csmae_weights_kslab from the bf16 weight mirror after every optimizer
 * step.  Rows of a slab are 64 B, so 16 consecutive rows of one 32-wide K half are 1 KiB contiguous: the 128 x 256-tile kernel that runs TWO
 * 4-wave workgroups per CU (one's C-tile stores run under the other's MFMA loop) stages its weight operand per wave and per K half in whole
 * cache lines (csrc/gemm_k2.hip).  `B_plain` ([N][K], torch's layout) is used for the shapes that kernel does not take (K % 64 != 0, M < 128,
 * N < 256, fp32): same results either way (same MFMA shape and K order: bit-identical to csmae_gemm).  The dX products (transB = 1) reach the
 * same kernel through csmae_gemm: W[K][N] is K-strided with 128-B row segments per wave by nature.  csmae_gemm_k2_mode(nn, nt) switches the
 * two uses off / on (A/B aid, default on).  timm Block call sites: MAE_ViT_Baseline.py:160-188. */
int csmae_gemm_ks(int dtype, long long M, long long N, long long K, const void* A, long long lda, const void* Bk, long long slab_rows,
                  const void* B_plain, long long ldb_plain, void* C, long long ldc, int c_dtype, const float* bias, int epilogue,
                  void* aux, long long ldaux, const void* resid, long long ldr, void* stream);
/* desc: device int64 [count][3] = {offset (elements) of weight i in src AND dst, out-features N, in-features K (K % 64 == 0)}; one launch of
 * (max_blocks, count) workgroups */
int csmae_weights_kslab(int count, const long long* desc, int max_blocks, const void* src_bf16, void* dst_bf16, void* stream);
/* ---- a Linear product of timm's Block WITH the LayerNorm next to it in its epilogue (ABI version 7; csrc/gemm_ln.hip): a 128 x 512 tile — eight
 * waves side by side along N, one workgroup per CU — holds whole output rows (N <= 512, N % 64 == 0: the decoders), so the row statistics are made
 * inside the workgroup.  All tensors bf16 (the throughput mode's residual stream); bias / gamma / beta / mean / rstd fp32; K % 64 == 0.
 *   fwd  X = A Wk^T + bias + resid ; Y = LayerNorm(X; gamma, beta, eps) ; mean / rstd of X          [Wk: the K-slab mirror, as csmae_gemm_ks]
 *        = `x = x + attn.proj(..)` -> `norm2(x)`, and `x = x + mlp.fc2(..)` -> the NEXT block's `norm1(x)`     (MAE_ViT_Baseline.py:160-188)
 *   bwd  dx_out = LayerNorm'(dY W; x, mean, rstd, gamma) + dres_in ; partial_ws[ceil(M / 128)][2][N] = the tiles' dgamma / dbeta rows
 *        (W [K][N] as torch stores the layer's weight, K = out-features) = the backward of `norm2 -> mlp.fc1` and `norm1 -> attn.qkv`; the
 *        product itself never reaches HBM.  Same arithmetic as csmae_gemm + csmae_layernorm_bwd on the bf16-rounded product; fixed summation
 *        order (bit-reproducible).  csmae_ln_param_reduce_rows folds the partial rows (rows = ceil(M / 128)). */
int csmae_gemm_ln_supported(long long M, long long N, long long K);
int csmae_gemm_ln_fwd(long long M, long long N, long long K, const void* A, long long lda, const void* Bk, long long slab_rows,
                      const float* bias, const void* resid, long long ldr, void* X, long long ldx, const float* gamma, const float* beta,
                      float eps, void* Y, long long ldy, float* mean, float* rstd, void* stream);
int csmae_gemm_ln_bwd(long long M, long long N, long long K, const void* dY, long long lda, const void* W, long long ldb,
                      const void* x, long long ldx, const float* mean, const float* rstd, const float* gamma, const void* dres_in, long long ldd,
                      void* dx_out, long long ldo, float* partial_ws, long long partial_elems, void* stream);
int csmae_gemm_k2_mode(int nn, int nt);
int csmae_gemm_dw_mode(int k2);   /* csmae_gemm_dw_group on the 256 x 256 one-workgroup kernel (0, default) or the two-workgroups-per-CU one (1): measured slower in the step, kept as an option */

/* weight gradient of nn.Linear: dW[M=out,N=in] (fp32, contiguous) += dY[K,M]^T X[K,N]; token axis split over the chip into fp32 slabs in
 * `workspace` (>= M*N+M floats; more = more slices), folded by a deterministic reduce; db[M] (nullable) += column sums of dY, computed
 * inside the same kernel by an all-ones MFMA operand (util/misc.py:314 backward products). */
int csmae_gemm_dw(int dtype, long long M, long long N, long long K, const void* dY, long long ldy, const void* X, long long ldx,
                  float* dW, float* db, float* workspace, long long ws_elems, void* stream);
/* The weight gradients of several nn.Linear layers that reduce over the same K tokens (a transformer block's qkv / proj / fc1 / fc2:
 * MAE_ViT_Baseline.py:160-188 backward) in ONE launch: host arrays of `count` (<= 8) device pointers / sizes.  Together the products
 * fill `slots` workgroups (<= 0: 128) with no or few K slices.  One slice: the tile is accumulated into dW / db directly; several: dense
 * fp32 slabs + one ordered fold for the whole group (bit-reproducible either way).  workspace: >= tiles * slices * 65 792 floats (less
 * = fewer slices).  fp32 parity mode and products with M or N < 256 are run one by one through csmae_gemm_dw. */
int csmae_gemm_dw_group(int dtype, int count, long long K, const void* const* dY, const long long* ldy, const void* const* X,
                        const long long* ldx, float* const* dW, float* const* db, const long long* M, const long long* N, int slots,
                        float* workspace, long long ws_elems, void* stream);
/* The same grouped weight gradients with fp8 operands (ABI version 7; BASELINE.json configs[4]; csrc/gemm_fp8_dw.hip): dW[i] += dq_y[i] dq_x[i] dY8[i]^T X8[i],
 * db[i] (nullable) += dq_y[i] colsum(dY8[i]); dY8 e5m2 bytes [K][ldy] (the gradient copies the dX products read), X8 e4m3 bytes [K][ldx] (the activation
 * copies the forward products read), dq_* device scalars (the de-quantisation factors their producers left), fp32 accumulation on
 * v_mfma_scale_f32_16x16x128_f8f6f4 with both operands K-strided (ds_read_b64_tr_b8).  M, N >= 256, N % 4 == 0, ld % 16 == 0.  Workspace as csmae_gemm_dw_group
 * plus count * 64 * roundup(max M, 256) floats for the bias-gradient partial sums.  Fixed summation order.  timm Block backward: MAE_ViT_Baseline.py:160-188. */
int csmae_gemm_dw_group_fp8(int count, long long K, const void* const* dY, const long long* ldy, const float* const* dq_y, const void* const* X,
                            const long long* ldx, const float* const* dq_x, float* const* dW, float* const* db, const long long* M, const long long* N,
                            int slots, float* workspace, long long ws_elems, void* stream);
/* ---- fp8 MFMA path (BASELINE.json configs[4]: "fp8 MFMA GEMMs"; the same nn.Linear call sites, MAE_ViT_Baseline.py:160-188).
 * OCP fp8, per-tensor scales.  An amax is 64 partial maxima (float[64]; its value is their maximum: thousands of same-address atomics per
 * launch serialise in L2): csmae_fp8_amax folds max|x| into slots the caller zeroed; csmae_fp8_quantize writes
 * q = fp8(x * FMAX / amax) (fmt 0 = e4m3, 1 = e5m2; transpose = 1 writes dst[c][r], the mirror of a weight the dX products read) and
 * the de-quantisation factor dq = amax / FMAX; with amax_next the amax passed in is the previous step's (one pass over the tensor,
 * out-of-range values saturate) and this step's is recorded for the next.  csmae_gemm_fp8: C[M,N] = dq_a * dq_b * sum_k A8(m,k) B8(n,k) with both operands
 * K-contiguous fp8 bytes (A in a_fmt, B e4m3), fp32 accumulation (v_mfma_scale_f32_16x16x128_f8f6f4), epilogues of csmae_gemm.  With the fused fp8 copy
 * (q_out) and an epilogue without a residual, C may be NULL: the copy (and the gelu' codes) is then the product's only output — fp8 mode's steady state,
 * where the next product, the weight gradients and the bias gradients all read the fp8 bytes. */
int csmae_fp8_amax(int in_dtype, long long rows, int cols, const void* src, long long ld, float* amax, void* stream);
int csmae_fp8_quantize(int in_dtype, int fmt, int transpose, long long rows, int cols, const void* src, long long ld, void* dst,
                       long long ldd, const float* amax, float* dq, float* amax_next /* nullable: delayed scaling, += max|src| */, void* stream);
/* every fp8 weight mirror of a model in three launches (amax, W8, W8^T; blockIdx.y = weight): desc (device, 3 int64 per weight) = {offset of the
 * fp32 master in p — the same offset is used in the byte mirrors w8 ([out][in]) and w8t ([in][out]) —, out, in}; amax [count][64] zeroed by the
 * caller, dq [count] receives the de-quantisation factors (current scaling, e4m3). */
int csmae_fp8_weights(int count, const long long* desc, const float* p, void* w8, void* w8t, float* amax, float* dq, void* stream);
int csmae_gemm_fp8(int a_fmt, long long M, long long N, long long K, const void* A, long long lda, const void* B, long long ldb,
                   void* C, long long ldc, int c_dtype, const float* bias, int epilogue, void* aux, long long ldaux,
                   const void* resid, long long ldr, const float* dq_a, const float* dq_b,
                   void* q_out /* nullable: also emit C as fp8 bytes [M][ldq] in q_fmt for the GEMM that reads it next (delayed scaling) */,
                   long long ldq, int q_fmt, const float* q_amax_prev /* [64] */, float* q_amax_next /* [64] */, float* q_dq, void* stream);
/* tuning hook for tools/gemm_bench.py: force the bf16 block tile (0: 128x128, 1: 256x128, 2: 256x256, -1: heuristic) */
int csmae_gemm_force_tile(int cfg);

/* ---- softmax attention of timm Block (Attention.forward: softmax(q k^T * hd^-0.5) v), qkv is [B*T, 3*H*hd]
 * in timm's (3, H, hd) column order; out [B*T, H*hd]; lse [B, H, T] (natural log). */
int csmae_attn_fwd(int dtype, long long B, int T, int H, int hd, const void* qkv, void* out, float* lse, void* stream);
int csmae_attn_bwd(int dtype, long long B, int T, int H, int hd, const void* qkv, const void* out, const void* dout,
                   const float* lse, void* dqkv, void* stream);
/* fp8 mode (BASELINE configs[4]): the same two kernels also leave their output as OCP fp8 bytes for the nn.Linear that consumes it —
 * attn.proj forward reads `out` (q_fmt 0: e4m3), attn.qkv's backward reads dqkv (q_fmt 1: e5m2) — quantised from the rounded bf16 values
 * with FMAX / max(q_amax_prev[64]) (the tensor's amax one step earlier: delayed scaling, conventions of csmae_gemm_fp8); q_amax_next[64]
 * receives partial maxima of |x|, q_dq[0] the de-quantisation factor.  bf16 shapes of the LDS-resident kernels only: csmae_attn_resident().
 * Same reference expression (timm Attention inside Block, models_mae/MAE_ViT_Baseline.py:160-188); replaces a csmae_fp8_quantize pass.
 * csmae_attn_bwd_q: dqkv may be NULL — the fp8 copy is then the only output (no reader of the bf16 gradient in fp8 mode's steady state). */
int csmae_attn_resident(int dtype, int T, int hd);
int csmae_attn_fwd_q(int dtype, long long B, int T, int H, int hd, const void* qkv, void* out, float* lse, void* q_out, int q_fmt,
                     const float* q_amax_prev, float* q_amax_next, float* q_dq, void* stream);
int csmae_attn_bwd_q(int dtype, long long B, int T, int H, int hd, const void* qkv, const void* out, const void* dout, const float* lse,
                     void* dqkv, void* q_out, int q_fmt, const float* q_amax_prev, float* q_amax_next, float* q_dq, void* stream);

/* ---- nn.LayerNorm(eps=1e-6) (MAE_ViT_Baseline.py:43-45): x [M,D] in x_dtype (the residual stream: fp32, or bf16 in throughput
 * mode); y in out_dtype (+ optional fp32 copy y32); statistics in fp32.
 * bwd: dx_out = dres_in + LN'(dy), both in x_dtype (the residual-gradient stream); dx_lp = optional low-precision copy for the next
 * GEMM (with a bf16 stream dx_out is that operand); dgamma/dbeta += via per-block partial rows in partial_ws (>= 2*D floats per
 * block; null -> atomics).  dgamma == NULL with a workspace: the partial rows (min(ceil(M/4), 1024, partial_elems / 2D) of them) are
 * left there for csmae_ln_param_reduce, which folds a whole batch of LayerNorms in one launch off the critical path.
 * q_out (nullable; fp8 mode, delayed scaling): the output that feeds the next GEMM (y / dx_out) also leaves as fp8 bytes [M][D] in q_fmt,
 * scaled with q_amax_prev (64 partial maxima), the new amax folded into q_amax_next (64 slots), the de-quantisation factor in q_dq.
 * fwd with q_out: y may be NULL — only the fp8 copy (and mean / rstd) is written (fp8 mode's steady state: every reader of y takes the fp8 bytes). */
int csmae_layernorm_fwd(int x_dtype, int out_dtype, long long M, int D, const void* x, const float* gamma, const float* beta, float eps,
                        void* y, float* y32, float* mean, float* rstd, void* q_out, int q_fmt, const float* q_amax_prev,
                        float* q_amax_next, float* q_dq, void* stream);
int csmae_layernorm_bwd(int dy_dtype, int x_dtype, int lp_dtype, long long M, int D, const void* dy, const void* x, const float* mean,
                        const float* rstd, const float* gamma, const void* dres_in, void* dx_out, void* dx_lp,
                        float* dgamma, float* dbeta, float* partial_ws, long long partial_elems, void* q_out, int q_fmt,
                        const float* q_amax_prev, float* q_amax_next, float* q_dq, void* stream);
/* LayerNorm k of the batch: partial rows at partials + k * stride (slice_elems floats each, written by csmae_layernorm_bwd with the same
 * M and D), dgamma at gbase + goff[2k], dbeta at gbase + goff[2k+1] (device array).  Fixed summation order, no atomics. */
int csmae_ln_param_reduce(int count, long long M, int D, const float* partials, long long stride, long long slice_elems,
                          float* gbase, const long long* goff, void* stream);
/* the same fold with the number of partial rows per LayerNorm given by the caller (csmae_gemm_ln_bwd leaves ceil(M / 128) of them); stride >= rows * 2 * D */
int csmae_ln_param_reduce_rows(int count, int rows, int D, const float* partials, long long stride, float* gbase, const long long* goff, void* stream);

/* ---- predictor BatchNorm1d(num_patches) + ReLU (models_mae/MLP.py:7-8): channel = token position, batch statistics
 * over (sample, feature); updates running stats (momentum, unbiased var) and num_batches_tracked in place. */
int csmae_bnrelu_fwd(int dtype, int N, int L, int Hp, const void* u, const float* gamma, const float* beta, float eps,
                     float momentum, void* r, float* mean, float* rstd, float* running_mean, float* running_var,
                     long long* num_batches_tracked, int training, void* stream);
int csmae_bnrelu_bwd(int dtype, int N, int L, int Hp, const void* u, const void* dr, const float* gamma, const float* beta,
                     const float* mean, const float* rstd, void* du, float* dgamma, float* dbeta, void* stream);

/* ---- MAE_ViT_MsLd.py:29-35,52: crop box (device int[4] i,j,h,w) + bilinear anti-aliased resize back to SxS */
int csmae_crop_resize(long long planes, int S, const float* src, float* dst, const int* box, void* stream);
/* ---- MAE_ViT_Shared.random_masking (:57-84): stable ascending argsort of noise; int64 ids_restore, f32 mask, kept ids */
int csmae_mask_sort(long long rows, int L, int keep, const float* noise, long long* ids_restore, float* mask, int* ids_keep,
                    int* ids_shuffle, void* stream);
/* ---- timm PatchEmbed on the kept patches (MAE_ViT_Baseline.py:245,251): im2col rows for the patch-embed GEMM */
int csmae_patch_gather(int dtype, long long rows, int keep, int N, int C, int S, int p, const float* img0, const float* img1,
                       const int* ids_keep, void* out, long long ld, void* stream);
/* ---- MAE_ViT_Baseline.py:248,253-256: + encoder_pos_embed, cls prepend (and its backward) */
int csmae_embed_assemble(int x_dtype, long long B2, int keep, int D, const float* tok, const float* pos, const float* cls, const int* ids_keep,
                         void* x, void* stream);
int csmae_embed_assemble_bwd(int in_dtype, int dtype, long long B2, int keep, int D, const void* dx, void* dtok, float* dcls, void* stream);
/* ---- MAE_ViT_Baseline.forward_decoder (:273-283): mask-token fill, gather(ids_restore), + decoder_pos_embed */
int csmae_unshuffle_fwd(int x_dtype, long long B2, int L, int keep, int Dd, const float* z, const float* mask_token, const float* dpos,
                        const long long* ids_restore, void* xd, void* stream);
int csmae_unshuffle_bwd(int in_dtype, int dtype, long long B2, int L, int keep, int Dd, const void* dxd, const long long* ids_restore, void* dz,
                        float* dmask_token, void* stream);
/* ---- `[:, 1:, :]` views around the predictor (MAE_ViT_MsLdCeCd.py:57-58) */
int csmae_rows_gather(int dtype, long long rows, int D, const float* src, long long group, long long gstride, long long off,
                      void* dst, void* stream);
int csmae_rows_scatter_add(int dtype, long long rows, int D, const void* src, float scale, long long group, long long gstride,
                           long long off, float* dst, void* stream);
/* two sources into two row views of the same buffer in one pass: dst[view(r) + off_a] += scale_a a[r], dst[view(r) + off_b] += scale_b b[r]
 * (the cross-decoder loss' gradient w.r.t. its target — the original's decoder embedding — and the predictor's input gradient — the
 * crop's —, MAE_ViT_MsLdCeCd.py:56-59: `loss_cd(x_embed_orig[:, 1:], predictor(x_embed_crop[:, 1:]))`, target not detached). */
int csmae_rows_scatter_add2(int dtype, long long rows, int D, const void* a, float scale_a, long long off_a, const void* b, float scale_b, long long off_b,
                            long long group, long long gstride, float* dst, void* stream);
/* The cross-decoder head's backward chain run AHEAD of the upstream gradient (ABI version 7): pair loss -> predictor Linear -> BatchNorm / ReLU -> Linear
 * depends on d(loss) only through a scalar factor, so csmae_hip runs it with unit gradient at the end of the forward pass (it is the longest chain of the
 * forward / backward junction); when the real factor g (device scalar) arrives, this call adds g x (tmp_dgamma, tmp_dbeta) — BatchNorm's parameter
 * gradients of the unit run — into dgamma / dbeta and scales the chain's three bf16 gradient tensors (n0 / n1 / n2 elements) by g in place: nothing to
 * scale when g == 1.  MAE_ViT_MsLdCeCd.py:56-59 backward. */
int csmae_spec_fixup(const float* g, void* b0, long long n0, void* b1, long long n1, void* b2, long long n2, const float* tmp_dgamma, const float* tmp_dbeta,
                     float* dgamma, float* dbeta, int L, void* stream);
/* MAE_ViT_Shared.py:77 (`torch.gather(x, dim=1, index=ids_keep.unsqueeze(-1).repeat(1, 1, D))` of the stand-alone random_masking):
 * out[n, k, :] = x[n, ids[n * ids_ld + k], :], x [N, L, D] fp32, ids int32 (csmae_mask_sort's ids_keep), out [N, keep, D] fp32. */
int csmae_rows_gather_idx(long long N, int L, int keep, int D, const float* x, const int* ids, long long ids_ld, float* out, void* stream);

/* ---- MAE_ViT_Shared.forward_loss (:269-290) with process_target/patchify fused (:24-39,97-111).
 * pred is [B2*(L+1), ldp] (row 0 of every sample = cls, ignored) in `pred_dtype` (fp32, or the bf16 the throughput path's decoder_pred
 * product writes — what autocast hands the reference's loss); rowloss [B2*L].  fwd: `mask` (nullable, [B2*L]) lets the kernel skip the
 * patches the loss does not weigh (:113-120): their rowloss is 0. */
int csmae_target_minmax(int norm_pix, long long B2, int N, int C, int S, int p, const float* img0, const float* img1,
                        float* scratch, float* out, void* stream);
int csmae_recon_loss_fwd(int kind, int norm_pix, int pred_dtype, long long B2, int N, int C, int S, int p, const float* img0, const float* img1,
                         const void* pred, long long ldp, const float* minmax, const float* mask, float* rowloss, void* stream);
int csmae_recon_loss_bwd(int kind, int norm_pix, int out_dtype, int pred_dtype, long long B2, int N, int C, int S, int p, const float* img0,
                         const float* img1, const void* pred, long long ldp, const float* minmax, const float* mask,
                         const float* losses, const float* gout, float vscale, const float* extra, void* dpred, long long ldd,
                         void* stream);
/* ---- the ssim family of reconstruction losses: MAE_ViT_Shared.forward_loss_{ssim,ms_ssim} (:165-247) around
 * pytorch-msssim 0.2.1 ssim / ms_ssim (:204,:247).  levels = 1 (ssim, nonnegative) or 5 (ms_ssim).  `ws` holds the image planes,
 * partial sums and statistics between the calls (csmae_ssim_workspace_floats floats, caller-allocated, untouched between fwd and bwd).
 * fwd: terms[v] = 1 - score of view v.   apply (after csmae_loss_finalize): losses[1+v] (+)= weight * terms[v], total fixed up
 * (`pure` = no per-patch term, the plain ssim / ms_ssim kinds; otherwise the 0.1-weighted mse_* kinds :249-267).
 * flags serve util/metrics.py (ssim / ms_ssim of two image batches in [0, 1], :4-10,:41-46): 1 = no min-max scaling, 2 = no relu.
 * bwd: extra[B2*L][P] (fp32) = gout * scale * d terms / d pred, handed to csmae_recon_loss_bwd (kind 5 = no per-patch term). */
int csmae_ssim_workspace_floats(long long B2, int C, int S, int p, int levels, long long* floats /* host pointer, out */);
int csmae_ssim_fwd(int levels, int flags /* 1: operands as they are (no scale_01) | 2: signed single-scale score */, int norm_pix,
                   long long B2, int N, int C, int S, int p, const float* img0, const float* img1, const float* pred, long long ldp,
                   const float* mask, float* ws, float* terms, void* stream);
int csmae_ssim_apply(int pure, int views, float weight, float recon_scale, const float* terms, float* losses, void* stream);
int csmae_ssim_bwd(int levels, long long B2, int N, int C, int S, int p, const float* pred, long long ldp, const float* mask,
                   const float* gout, float scale, float* ws, float* extra, void* stream);
/* ---- un-masked `.mean()` losses: cross-decoder (MAE_ViT_MsLdCeCd.py:56-59, target NOT detached) and latent (MsLdLe.py:44) */
int csmae_pair_loss_fwd(int kind, long long rows, int D, const float* a, long long a_group, long long a_gstride, long long a_off,
                        const float* t, long long t_group, long long t_gstride, long long t_off, float* partial, void* stream);
int csmae_pair_loss_bwd(int kind, int lp_dtype, long long rows, int D, const float* a, long long a_group, long long a_gstride,
                        long long a_off, const float* t, long long t_group, long long t_gstride, long long t_off,
                        const float* gout, float coef, void* da_lp, float* da_acc, float* dt_acc, void* stream);
/* ---- util/contrast_loss.NTXentLoss(bs, 0.5, cos_sim=True) on mean-pooled latents (MAE_ViT_MsLdCeCd.py:61-69) */
int csmae_ntxent_fwd(int N, int Te, int keep, int D, const float* latent, float tau, float eps, float* z, float* inv_norm,
                     float* E, float* neg, float* rowloss, void* stream);
int csmae_ntxent_bwd(int N, int D, const float* z, const float* inv_norm, const float* E, const float* neg, float tau, float eps,
                     const float* gout, float* dpool, void* stream);
int csmae_latent_grad_finish(int lp_dtype, long long B2, int Te, int D, float* dlat, const float* dpool, float inv_keep,
                             void* dlat_lp, void* stream);
/* ---- scalar assembly: losses[0]=total [1]=recon orig [2]=recon crop [3]=cross-decoder [4]=contrastive [5]=latent
 *      [6],[7]=sum(mask) per view (MAE_ViT_MsLd.py:64-66, MAE_ViT_MsLdCeCd.py:72) */
int csmae_loss_finalize(long long per_view, int views, const float* rowloss, const float* mask, float recon_scale,
                        const float* cd_partial, float cd_scale, const float* e_partial, float e_scale,
                        const float* ce_rowloss, int ce_rows, float* losses, void* stream);

/* ---- input step before the path (util/datasets.py:120-136 training transform): decoded uint8 HWC images ->
 * ToTensor, Normalize(mean, std), horizontal / vertical flip, RandomResizedCrop(S, bicubic, antialias) in one kernel.
 * src [N, Hmax, Wmax, C] u8 (image n in the top-left H x W corner), meta [N, 8] int32 = {H, W, i, j, h, w, hflip, vflip}
 * (box in the coordinates of the flipped image; drawn on the host in torchvision's RNG order), dst [N, C, S, S] fp32. */
int csmae_augment_u8(long long N, int C, int Hmax, int Wmax, int S, const unsigned char* src, const int* meta, const float* mean,
                     const float* inv_std, float* dst, void* stream);

/* ---- optimizer side (main_pretrain.py:426-427 torch.optim.AdamW; util/misc.py:314 backward products) */
/* gate (nullable device scalar): the update is skipped as a whole when it is not finite — engine_pretrain.py:56-58 raises on a
 * non-finite loss BEFORE backward / step; here the loss stays on the device and the kernel keeps the poisoned step out of the weights. */
/* tile_ks / p_ks (nullable, ABI 6): the launch also writes the K-slab mirrors (csmae_gemm_ks) of the weights it steps — tile_ks[tile] =
 * {flat offset of the tile's weight, N, K} (device int64 [ntiles][3]; K = 0: none) — so that no csmae_weights_kslab launch is needed per step. */
int csmae_adamw(long long ntiles, const long long* tile_off, const int* tile_cnt, const float* tile_wd, float* p, const float* g,
                float* m, float* v, float lr, float beta1, float beta2, float eps, float bias_correction1, float bias_correction2,
                void* p_lp, const float* gate, const long long* tile_ks, void* p_ks, void* stream);
/* The same step for the block Linear weights of an fp8-mode model (ABI version 7), writing their fp8 mirrors on the way: W8 [out][in] (e4m3) for the forward
 * products, W8^T [in][out] for dX — instead of csmae_fp8_weights' three launches per step.  tile8: device int64 [ntiles][6] = {flat offset of the weight,
 * N = out, K = in (multiples of 64), n0, k0 (the workgroup's 64 x 64 sub-block), weight index}.  Delayed scaling: scale 448 / amax_prev[w] (64 partial
 * maxima per weight, one step old), the new maximum into amax_next[w] (zeroed by the caller), dq[w] = amax_prev[w] / 448.  One weight-decay value per launch. */
int csmae_adamw_fp8(long long ntiles, const long long* tile8, float weight_decay, float* p, const float* g, float* m, float* v, float lr, float beta1,
                    float beta2, float eps, float bias_correction1, float bias_correction2, void* p_lp, const float* gate, void* w8, void* w8t,
                    const float* amax_prev, float* amax_next, float* dq, void* stream);
int csmae_gate_accumulate(const float* loss, float* slot, int accumulate, void* stream);
/* util/misc.py:310-318 (`torch.nn.utils.clip_grad_norm_(parameters, clip_grad)`) / :338-355 (`get_grad_norm_`) on the flat gradient
 * buffer: out[0] = total 2-norm, out[1] = min(1, max_norm / (norm + 1e-6)); g *= out[1] in place when max_norm > 0 (<= 0: norm only).
 * scratch: >= 1024 floats.  No host synchronisation. */
int csmae_clip_grad_norm(long long n, float* g, float max_norm, float* scratch, float* out, void* stream);
int csmae_cast_f32_to_bf16(long long n, const float* src, void* dst, void* stream);
int csmae_cast_bf16_to_f32(long long n, const void* src, float* dst, void* stream);
int csmae_colsum(int dtype, long long M, int N, const void* x, long long ld, float* out, void* stream);


/* ---- a stream confined to a subset of the compute units (ABI version 5).  The reference overlaps DDP's bucket all-reduces and autograd's
 * weight-gradient work with the main chain on CUDA streams that share every SM (main_pretrain.py:417-421); on MI355X a GEMM workgroup owns a
 * whole CU (160 KiB of LDS, 512 threads), so two streams time-slice CUs unless each is given its own.  `mask`: `words` x 32 bits; on this
 * 8-XCD part bit i is CU (i / 8) of XCD (i % 8) (the driver deals the bits round-robin over XCDs, then over shader engines), so the
 * first 8 n bits are n CUs of every XCD.  `*out` receives a hipStream_t (wrap it with torch.cuda.ExternalStream); destroy it with
 * csmae_stream_destroy once no work is pending on it. */
int csmae_stream_create_cu_mask(int words, const unsigned* mask, void** out);
int csmae_stream_destroy(void* stream);

#ifdef __cplusplus
}
#endif
#endif
