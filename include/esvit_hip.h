/* libesvit_hip.so -- C ABI of the MI355X-native EsViT pre-training hot path.
 *
 * Every entry point replaces a stretch of the reference's PyTorch code (cited per
 * function as reference file:line, paths relative to microsoft/esvit).  Conventions
 * (SURVEY.md 8b):
 *   - all buffers are caller-allocated DEVICE memory passed as raw pointers; the
 *     library never allocates, frees or retains device memory;
 *   - all work is enqueued on the caller's hipStream_t; no internal synchronisation;
 *   - return 0 on success, negative on failure (ESVIT_ERR_*); the message is
 *     available through esvit_last_error(); no C++ exception crosses the ABI;
 *   - `dtype` selects the storage type of *activations* (ESVIT_F32 / ESVIT_BF16);
 *     parameters, the residual stream, statistics and gradients of parameters are
 *     always fp32.
 */
#ifndef ESVIT_HIP_H
#define ESVIT_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default) /* the library is built with -fvisibility=hidden: this header IS the export list */

#define ESVIT_F32 0
#define ESVIT_BF16 1

#define ESVIT_OK 0
#define ESVIT_ERR_ARG (-1)
#define ESVIT_ERR_HIP (-2)
#define ESVIT_ERR_UNSUPPORTED (-3)

typedef void* esvit_stream_t; /* hipStream_t */

/* ---- library ---------------------------------------------------------- */
int esvit_version(void);
const char* esvit_last_error(void);
/* Sizes of caller-allocated scratch / capability questions, one entry point (no device work, no state):
 *   ESVIT_Q_ATTN_FRAG_ELEMS (N)                floats per N x N matrix in MFMA fragment order (4096 for N <= 64), -1 unsupported
 *   ESVIT_Q_ATTN_LSE_ELEMS (N)                 per-(window, head) log-sum-exp floats the forward writes (0 for N <= 64)
 *   ESVIT_Q_ATTN_BWD_PARTS (N, Bw, nH)         leading dimension of dbias_ws of esvit_window_attn_bwd
 *   ESVIT_Q_ATTN_BWD_PAD_ROWS (N, Bw, nH | dtype << 32)  rows of dpad_ws of esvit_window_attn_bwd
 *   ESVIT_Q_LN_BWD_BLOCKS (rows, C)            nblk of the [nblk, 2, C] LayerNorm-backward scratch
 *   ESVIT_Q_COLSUM_BLOCKS (rows)               blocks of the esvit_colsum scratch
 *   ESVIT_Q_COL_REDUCE_BLOCKS (rows)           blocks of the esvit_dwconv3x3_wgrad / esvit_col_sums2 scratch
 *   ESVIT_Q_UPDATE_CHUNK_ELEMS ()              elements per chunk of the fused update's chunk table
 *   ESVIT_Q_MLP_FUSED (dtype, C)               bit 0: esvit_mlp_fused_fwd exists (bf16, C in {96, 128, 192, 256, 384}), bit 1: esvit_mlp_fused_bwd exists (bf16, C in {96, 128, 192, 256})
 *   ESVIT_Q_AUG_MAX_BOX (S)                    largest crop-box side esvit_aug_crops resizes to S x S
 * Unknown `what` returns ESVIT_ERR_ARG. */
#define ESVIT_Q_ATTN_FRAG_ELEMS 1
#define ESVIT_Q_ATTN_LSE_ELEMS 2
#define ESVIT_Q_ATTN_BWD_PARTS 3
#define ESVIT_Q_ATTN_BWD_PAD_ROWS 4
#define ESVIT_Q_LN_BWD_BLOCKS 5
#define ESVIT_Q_COLSUM_BLOCKS 6
#define ESVIT_Q_COL_REDUCE_BLOCKS 7
#define ESVIT_Q_UPDATE_CHUNK_ELEMS 8
#define ESVIT_Q_MLP_FUSED 9
#define ESVIT_Q_AUG_MAX_BOX 10
int64_t esvit_query(int what, int64_t a, int64_t b, int64_t c);

/* ---- host-side integer index maps (bit-exact vs reference) -------------
 * swin_transformer.py:100-110 (relative_position_index), :40-69 + :286-325
 * (pad -> roll -> window_partition and its inverse), :249-272 (shift mask).  */
/* out[N*N], N = ws*ws */
int esvit_relative_position_index(int ws, int64_t* out);
/* win2tok[nW*N]: source token (i*W+j) for every window slot, -1 for zero-pad slots.
 * tok2win[H*W]: window slot of every real token.
 * region_ids[nW*N] (shift > 0 only): region label of every window slot; mask[w,p,q] = (ids[w,p] == ids[w,q]) ? 0 : -100
 * (what the attention kernels consume: the 49x49 mask is rebuilt in registers instead of being loaded).
 * Any of the three pointers may be NULL. */
int esvit_window_maps(int H, int W, int ws, int shift, int32_t* win2tok, int32_t* tok2win, int32_t* region_ids);
/* mask[nW*N*N] in {0,-100}; returns nW through *n_windows */
int esvit_shift_mask(int H, int W, int ws, int shift, float* mask, int* n_windows);

/* ---- MFMA GEMM family -------------------------------------------------- */
#define ESVIT_EPI_NONE 0
#define ESVIT_EPI_GELU 1     /* out = gelu(acc+bias); aux (if set) receives acc+bias */
#define ESVIT_EPI_GELU_BWD 2 /* out = acc * gelu'(aux) */
#define ESVIT_EPI_QGELU 3     /* QuickGELU (cvt_v4_transformer.py:44-46): out = v*sigmoid(1.702 v), v = acc+bias; aux receives v */
#define ESVIT_EPI_QGELU_BWD 4 /* out = acc * d/dv[v*sigmoid(1.702 v)] at v = aux */

typedef struct {
    const void* A;
    const void* B;
    void* C;
    int32_t M, N, K;
    int64_t lda, ldb, ldc;
    int32_t a_kstrided; /* 0: A is M x K row-major (K contiguous); 1: A is K x M row-major */
    int32_t b_kstrided; /* 0: B is N x K row-major (K contiguous); 1: B is K x N row-major */
    int32_t batch;      /* >=1; strides in elements */
    int64_t strideA, strideB, strideC;
    const float* bias;     /* [N] or NULL */
    const float* residual; /* fp32, indexed [dst_row*ldr + n] or NULL */
    int64_t ldr;
    const int32_t* rowmap; /* NULL or [rowmap_period]: dst_row = (m/period)*rowmap_tokens + rowmap[m%period]; <0 => row dropped */
    int32_t rowmap_period, rowmap_tokens;
    const float* rowscale; /* NULL or per-sample scale, index dst_row / rows_per_sample */
    int32_t rows_per_sample;
    void* aux; /* activation dtype, ld = ldaux */
    int64_t ldaux;
    int32_t epilogue; /* ESVIT_EPI_* */
    int32_t out_f32;  /* 1: C is fp32, 0: C has the activation dtype */
    int32_t splitk;   /* >1: partial sums in `partial` then reduced into C (fp32, out_f32 must be 1) */
    float* partial;   /* workspace >= splitk*M*N floats */
    int32_t accumulate; /* split-K reduce: C += sum (1) or C = sum (0) */
    float alpha;
    float* colsum;         /* optional, fp32 [M]: row sums of op(A) over K (= bias gradient when A = dY^T); fused via an all-ones B fragment */
    float* colsum_partial; /* workspace >= splitk*M floats when splitk > 1 */
    int32_t kernel;        /* ESVIT_GEMM_AUTO (0): chosen from the shape; otherwise force one main loop (tests / tuning) */
    /* optional softmax statistics of the OUTPUT rows (the logits of the DINO head, vision_transformer.py:418): for every row and
     * every 64-column block j the pair (m, s) = (max_k z_k, sum_k 2^(z_k - m)) over k in [64 j, 64 j + 64) of
     * z_k = (C[row][k] as stored, i.e. rounded to the activation dtype, - rowstat_center[k]) * rowstat_scale
     * (scale = log2(e) / temperature; center NULL = 0).  rowstat: fp32 [M, N / 64, 2] -- or [M, N / 32, 2], blocks of 32 columns,
     * when the descriptor asks for ESVIT_GEMM_P8 (M and N whole 256-tiles; esvit_gemm_select reports what a descriptor resolves to -- AUTO keeps
     * the 128 x 128 tile for statistics, measured faster inside the step).  Only for the plain bf16 epilogue of a
     * dense forward GEMM with M % 128 == 0 and N % 128 == 0 (rejected otherwise); esvit_rowstat_combine folds the blocks of a row,
     * esvit_dino_ce_fwd_bwd takes them in place of its first pass over the logits (main_esvit.py:728-742). */
    float* rowstat;
    const float* rowstat_center;
    float rowstat_scale;
    /* optional, with rowstat on the 128 x 128 tile only (not ESVIT_GEMM_P8): fp32 [M / 64, N], the sums of the STORED logits over
     * each 64-row wave tile per column -- esvit_partial_reduce folds them into the batch sum the centre update needs
     * (main_esvit.py:752-770) without another pass over the logits. */
    float* colstat;
} esvit_gemm_desc;

/* main loops of the family (esvit_gemm_desc.kernel; what esvit_gemm_select returns) */
#define ESVIT_GEMM_AUTO 0
#define ESVIT_GEMM_REGSTAGE 1 /* register-staged 128-row tiles: the exact-fp32 mode, and a bf16 fallback */
#define ESVIT_GEMM_DMA4 2     /* bf16, LDS-DMA, 128 x {64,96,128} tiles, 4 waves, two workgroups per CU */
#define ESVIT_GEMM_DMA8 3     /* bf16, LDS-DMA, 256 x 256 tiles, 8 waves, one workgroup per CU: very long reductions */
#define ESVIT_GEMM_DMA4W 4    /* bf16, LDS-DMA, 4 waves, 128 x 192 / 128 x 96 tiles with whole-width wave rows (N % 96 == 0) */
#define ESVIT_GEMM_P8 5       /* bf16, LDS-DMA, 256 x 256 tiles, 8 waves, eight-phase schedule with counted DMA waits (K % 64 == 0, no rowmap; rowstat only over whole 256 x 256 tiles, in 32-column blocks, without colstat) */

/* C = alpha * op(A) op(B) (+ epilogue).  Replaces every nn.Linear / conv-as-GEMM on the
 * path: swin_transformer.py:31-37,127,150,418,531; vision_transformer.py:414-418;
 * and their autograd backward (dgrad: b_kstrided or cached W^T; wgrad: a_kstrided && b_kstrided). */
int esvit_gemm(int dtype, const esvit_gemm_desc* d, esvit_stream_t stream);
/* The main loop esvit_gemm would run for this descriptor (a pure function of it; > 0) with its output tile and the
 * number of workgroups the chip keeps resident at once -- what a caller needs to size split-K (one workgroup per
 * (tile, slice); a launch of more workgroups than resident slots runs in rounds) before it allocates `partial`. */
int esvit_gemm_select(int dtype, const esvit_gemm_desc* d, int* tile_m, int* tile_n, int* resident_slots);

/* ---- fused Swin MLP branch, forward and backward (swin_transformer.py:331 + 31-37) ---
 * y = x + rowscale[row] * ( GELU( LayerNorm(x) W1^T + b1 ) W2^T + b2 ) for the narrow stages (bf16, C in {96, 192}: Swin-T / -S, {128, 256}: Swin-B; C = 384: see below;
 * esvit_query(ESVIT_Q_MLP_FUSED, dtype, C, 0) bit 0 = forward, bit 1 = backward).  The unfused LayerNorm -> fc1 (+GELU) -> fc2
 * (+residual) sequence is bound there by the HBM round trips of the 4C-wide hidden activation (40 B per token-channel forward,
 * 64 B backward).  x, y fp32 [M, C]; W1 [4C, C], W2 [C, 4C] in the activation dtype; gamma, beta, b1, b2 fp32; rowscale fp32 [M]
 * (the DropPath factor of each row) or NULL.
 *
 * esvit_mlp_fused_fwd: one kernel, 8 B per token-channel, NOTHING hidden-sized is written (inference passes and training alike:
 *   the backward below recomputes).  Optional second output (gamma_next != NULL): LayerNorm(y) with the parameters of the
 *   NEXT block's norm1 (swin_transformer.py:283) -- xw_next [M, C] in the activation dtype, mean_next / rstd_next fp32 [M] --
 *   so that block needs no LayerNorm launch.
 * esvit_mlp_fused_bwd: the data-gradient path of the branch in one kernel.  Inputs: x (the branch input), gy = dL/dy fp32,
 *   rowscale_mlp (the forward's rowscale), W1, and the transposed copies W2T = W2^T [4C, C], W1T = W1^T [C, 4C]
 *   (esvit_mlp_fused_weight).  It recomputes LayerNorm(x) and the pre-activation, forms dA = (rowscale gy W2) o GELU'(A) and
 *   dH = dA W1 with the hidden tile in registers, applies the LayerNorm backward and writes
 *     gx      fp32 [M, C]   dL/dx = gy + LN'(dH)                 gx_act  act [M, C]  cast(rowscale_out[row] * gx) (NULL scale = 1)
 *     xhat    act [M, C]    (x - mean) rstd                      a1g     act [M, 4C] GELU(A)          da1  act [M, 4C]  dA
 *   The weight gradients are two esvit_gemm calls: dW2 = (rowscale gy)^T a1g, and G = da1^T xhat, db1 = colsum(da1) followed by
 * esvit_ln_fold_finish: LayerNorm folded out of a weight gradient.  With LN(x) = xhat o gamma + beta and G = dY^T xhat [J, C],
 *   db = colsum(dY) [J], W the fp32 master [J, C]:  dW = G o gamma + db (x) beta (written over G),
 *   dgamma[c] (+)= sum_j W[j, c] G[j, c],  dbeta[c] (+)= sum_j db[j] W[j, c]   (swin_transformer.py:331 autograd).
 */
int esvit_mlp_fused_fwd(int dtype, const float* x, const float* gamma, const float* beta, float eps, const void* W1,
                        const float* b1, const void* W2, const float* b2, const float* rowscale, int64_t M, int C, float* y,
                        const float* gamma_next, const float* beta_next, void* xw_next, float* mean_next, float* rstd_next,
                        esvit_stream_t stream);
/* The wide stage (C = 384, bf16: stage 2 of Swin-T / -S; esvit_query bit 0): esvit_mlp_fused_fwd is the inference pass (the EMA teacher:
 * x in, y out, 8 B per token-channel; gamma_next must be NULL).  esvit_mlp_fused_fwd_train is the same kernel for a pass whose backward is
 * the unfused sequence (swin_transformer.py:31-37 autograd through esvit_gemm): besides y it writes what that backward reads --
 *   a1 act [M, 4C] the pre-activation LN(x) W1^T + b1,  a1g act [M, 4C] GELU(a1),  h act [M, C] LayerNorm(x),  mean / rstd fp32 [M] --
 * in place of the LayerNorm launch and the two GEMM launches that would produce them.  W1 = ESVIT_MLP_W1_FWD (the plain cast at this width). */
int esvit_mlp_fused_fwd_train(int dtype, const float* x, const float* gamma, const float* beta, float eps, const void* W1, const float* b1,
                              const void* W2, const float* b2, const float* rowscale, int64_t M, int C, float* y, void* a1, void* a1g,
                              void* h, float* mean, float* rstd, esvit_stream_t stream);
int esvit_mlp_fused_bwd(int dtype, const float* x, const float* gy, const float* rowscale_mlp, const float* rowscale_out,
                        const float* gamma, const float* beta, float eps, const void* W1, const void* W2T, const void* W1T,
                        const float* b1, int64_t M, int C, float* gx, void* gx_act, void* xhat, void* a1g, void* da1,
                        esvit_stream_t stream);
int esvit_ln_fold_finish(float* G, const float* db, const float* W, const float* gamma, const float* beta, int J, int C,
                         float* dgamma, float* dbeta, int accumulate, esvit_stream_t stream);
/* The weight copies the fused kernels stream, produced from the fp32 masters (fc1.weight [4C, C], fc2.weight [C, 4C]); fc2.weight
 * itself is consumed as its plain activation-dtype cast.  esvit_mlp_fused_fwd takes W1 = ESVIT_MLP_W1_FWD; esvit_mlp_fused_bwd
 * takes W1 = ESVIT_MLP_W1_BWD, W2T = ESVIT_MLP_W2T_BWD, W1T = ESVIT_MLP_W1T_BWD (the channel order of a copy follows the kernel
 * generation that consumes it and is private to the library). */
#define ESVIT_MLP_W1_FWD 0
#define ESVIT_MLP_W1_BWD 1
#define ESVIT_MLP_W1T_BWD 2
#define ESVIT_MLP_W2T_BWD 3
int esvit_mlp_fused_weight(int kind, const float* src, void* dst_bf16, int C, esvit_stream_t stream);
/* dst (bf16) = cast(src fp32 [R, S]), transposed to [S, R] if `transpose`, and with perm32 != 0 every aligned 32-block of a dst row
 * reordered so that position 8g + e holds column 4g + e (e < 4) / 16 + 4g + (e - 4): the channel order of the 16-token kernels. */
int esvit_cast_weight(const float* src, void* dst_bf16, int R, int S, int transpose, int perm32, esvit_stream_t stream);

/* ---- normalisation ----------------------------------------------------- */
/* LayerNorm forward over rows of C channels (swin_transformer.py:283,331,417,546,687;
 * eps 1e-6 at :963).  x: fp32 [rows, C] (or gathered, see gather).  y has `dtype`.
 * rowmap (optional): y row = (r/tokens)*period_out + rowmap[r%tokens] (token -> window slot);
 * the caller pre-zeroes y so pad slots stay 0 (swin_transformer.py:286-290).
 * y_f32 (optional) receives an fp32 copy at the un-mapped row. */
int esvit_layernorm_fwd(int dtype, const float* x, const float* gamma, const float* beta, float eps,
                        int64_t rows, int C, void* y, float* y_f32, float* mean, float* rstd,
                        const int32_t* rowmap, int tokens, int period_out, esvit_stream_t stream);
/* LayerNorm backward.  dy (dtype) is read at the mapped row when rowmap is given.
 * dx = g_in (optional fp32 residual gradient) + LN'(dy).  dgamma/dbeta partials are written to ws ([nblk,2,C] floats,
 * nblk = esvit_query(ESVIT_Q_LN_BWD_BLOCKS, rows, C, 0)) and reduced into dgamma/dbeta (fp32, overwritten).
 * dx_act (optional, un-mapped rows only) = rowscale[row / rows_per_sample] * dx: the DropPath-scaled copy the next dgrad / wgrad
 * GEMMs of the backward read (rowscale may be NULL = 1) -- saves one esvit_gather_cast pass.  act_dtype: its dtype -- `dtype`, or
 * ESVIT_BF16 while dy is fp32 (the patch-embedding norm, whose upstream gradient is the fp32 residual-stream gradient).  dx may be NULL
 * when dx_act is given (only the copy is wanted; g_in must then be NULL). */
int esvit_layernorm_bwd(int dtype, const void* dy, const float* x, const float* mean, const float* rstd,
                        const float* gamma, const float* g_in, int64_t rows, int C, float* dx,
                        float* dgamma, float* dbeta, float* ws, const int32_t* rowmap, int tokens,
                        int period_in, void* dx_act, int act_dtype, const float* rowscale, int rows_per_sample,
                        esvit_stream_t stream);

/* ---- element-wise / data-movement helpers ------------------------------ */
/* dst[r,:] = cast(scale[r/rows_per_sample] * src[map(r),:]); src fp32 [*, C]; dst dtype [rows, C].
 * rowmap (optional, [period]): src row = (r/period)*tokens + rowmap[r%period]; <0 => zeros.
 * Used for dY staging of the windowed proj backward (swin_transformer.py:315-330 reversed). */
int esvit_gather_cast(int dtype, const float* src, void* dst, int64_t rows, int C, const int32_t* rowmap,
                      int period, int tokens, const float* rowscale, int rows_per_sample,
                      esvit_stream_t stream);
/* plain cast fp32 -> activation dtype (weight caches) */
int esvit_cast_f32_to(int dtype, const float* src, void* dst, int64_t n, esvit_stream_t stream);
/* out[n] (+)= sum_r x[r,n]  (bias gradients).  ws >= esvit_query(ESVIT_Q_COLSUM_BLOCKS, rows, 0, 0)*N floats */
int esvit_colsum(int dtype, const void* x, int64_t rows, int N, int64_t ld, float* out, float* ws,
                 int accumulate, esvit_stream_t stream);

/* PatchEmbed im2col for the 4x4/4 conv (swin_transformer.py:531,544): img fp32 NCHW [nB,3,S,S]
 * -> cols dtype [nB*(S/P)^2, Kpad], column order (c, ph, pw), zero padded to Kpad. */
int esvit_patch_im2col(int dtype, const float* img, void* cols, int nB, int S, int P, int Kpad,
                       esvit_stream_t stream);
/* PatchMerging gather + LayerNorm(4C) (swin_transformer.py:410-417): x fp32 [nB,H,W,C] ->
 * y dtype [nB*(H/2)*(W/2), 4C]; channel blocks [x(2i,2j), x(2i+1,2j), x(2i,2j+1), x(2i+1,2j+1)]. */
int esvit_merge_ln_fwd(int dtype, const float* x, const float* gamma, const float* beta, float eps,
                       int nB, int H, int W, int C, void* y, float* mean, float* rstd,
                       esvit_stream_t stream);
/* backward of the above: dy dtype [nB*(H/2)*(W/2), 4C] -> dx fp32 [nB,H,W,C] (overwritten); dgamma / dbeta overwritten, or
 * added to when accumulate != 0 (further resolution groups sharing the parameters).  dx_act (optional, `dtype`, [nB,H,W,C]) =
 * rowscale[token row / rows_per_sample] * dx as in esvit_layernorm_bwd: the MLP-branch operand of the stage's LAST block, whose
 * dL/dy this is. */
int esvit_merge_ln_bwd(int dtype, const void* dy, const float* x, const float* mean, const float* rstd,
                       const float* gamma, int nB, int H, int W, int C, float* dx, float* dgamma,
                       float* dbeta, float* ws, void* dx_act, const float* rowscale, int rows_per_sample, int accumulate,
                       esvit_stream_t stream);
/* token mean (swin_transformer.py:688-689): x fp32 [nB,T,C] -> out fp32 [nB,C] (+ act copy) */
int esvit_token_mean_fwd(int dtype, const float* x, int nB, int T, int C, float* out, void* out_act,
                         esvit_stream_t stream);
/* dx[b,t,:] = g_tok[b,t,:] (optional) + g_mean[b,:]/T */
int esvit_token_mean_bwd(const float* g_mean, const float* g_tok, int nB, int T, int C, float* dx,
                         esvit_stream_t stream);

/* ---- fused attention branch of a Swin block (swin_transformer.py:283-330 with 120-152) --------------------
 * y = x + rowscale * (proj(window_attention(qkv(LayerNorm(x)))) + b_proj) in ONE kernel for 7x7 windows (N = ws*ws <= 64),
 * head_dim 32, C = 32 nH in {96, 192}, bf16 activations: LayerNorm output, qkv and the attention output stay on the chip
 * (8 B per token-channel through HBM).  x, y fp32 [nB*L, C] token-ordered rows of one resolution group; win2tok / region_ids /
 * rel_table / bias_frag_ws / scale as for esvit_window_attn_fwd (zero-pad slots carry q / k / v = bias, pad query rows are
 * dropped).  Wqkv_p bf16 [3C, C] and Wproj_p bf16 [C, C]: esvit_cast_weight(W, perm32 = 1) of qkv.weight / proj.weight.
 * rowscale fp32 [nB*L] (DropPath factor of every row) or NULL.  Side outputs for a training pass whose backward runs the
 * unfused kernels -- all five or none: xw bf16 [nB*L, C] = LayerNorm(x), qkv bf16 [nB*L, 3C], ao bf16 [nB*L, C] (attention
 * output before the projection), mean / rstd fp32 [nB*L].  The rows of one call must fit 2 GiB buffer ranges. */
int esvit_attn_branch_fwd(int dtype, const float* x, const float* gamma, const float* beta, float eps, const void* Wqkv_p,
                          const float* bqkv, const void* Wproj_p, const float* bproj, const int32_t* win2tok, int L,
                          const float* rel_table, int ws, float* bias_frag_ws, const int32_t* region_ids, int nW, int nB, int N,
                          int nH, float scale, const float* rowscale, float* y, void* xw, void* qkv, void* ao, float* mean,
                          float* rstd, esvit_stream_t stream);

/* ---- window attention (swin_transformer.py:126-152) ---------------------
 * "frag layout" of an NP x NP matrix X[q][key] (NP = 64 for 7x7 windows): the order in which the
 * MFMA accumulators of the transposed score tile hold it,
 *   X_frag[((ki*4 + qj)*64 + lane)*4 + r] = X[q = 16*qj + c][key = 16*ki + 4*g + r], lane = 16*g + c.
 * esvit_query(ESVIT_Q_ATTN_FRAG_ELEMS, N, 0, 0) = floats per matrix in that layout (4096), -1 if N is unsupported. */
/* Token-ordered window attention.  qkv dtype [nB*L, 3C] (columns [3][nH][hd]) -> out dtype [nB*L, C].
 * win2tok int32 [nW*N] (esvit_window_maps): window slot -> token of the image, -1 for a zero-pad slot; this map
 * IS pad -> roll -> window_partition and its inverse (swin_transformer.py:286-325), applied on the fly.
 * qkv_bias fp32 [3C]: the value of q,k,v at a zero-pad slot (LayerNorm output is zero-padded, so qkv = bias there;
 * pad keys/values take part in every softmax exactly as in the reference, pad query rows are dropped).
 * rel_table fp32 [(2ws-1)^2, nH] is the relative_position_bias_table parameter itself (swin_transformer.py:133-136,
 * index in closed form).  bias_frag_ws (required): fp32 scratch [2, nH, ESVIT_Q_ATTN_FRAG_ELEMS(N)] the library fills with
 * the bias in MFMA fragment order (one 16-byte load per lane per score tile instead of table gathers in the kernel).
 * rel_table = NULL: bias_frag_ws already holds that (an earlier forward / backward call with the same table, ws and nH filled
 * it) -- a block's second resolution group and its backward reuse the forward's fill.
 * region_ids int32 [nW*N] (esvit_window_maps) for shifted blocks or NULL.
 * scale: applied to q before the product (swin_transformer.py:130: hd^-0.5; CvT passes dim^-0.5).  N = ws*ws <= 64 with
 * hd in {32, 64} (7x7 Swin windows; 7x7 / 6x6 / 3x3 CvT windows at hd 64), or N = 196 (14x14) with hd = 32.
 * lse fp32 [nB*nW*nH, ESVIT_Q_ATTN_LSE_ELEMS(N)]: per-query log-sum-exp, written for 14x14 windows (the blocked
 * backward needs it), unused (may be NULL) for 7x7.  Without attn_out, the 16-slot query tiles of a window that hold no live
 * token (win2tok < 0 in all 16 slots: the padding of a 96^2 crop's window) are not computed -- their output rows do not exist and
 * their lse entries read 0 (a placeholder: esvit_window_attn_bwd either skips the same tiles or treats them as P = 0, it never exponentiates
 * against that 0).  One image's qkv rows (L * 3C activations) must fit a 2 GiB buffer
 * descriptor.
 * attn_out (optional, fp32 [nB*nW,nH,N,N]) receives the softmax (swin_transformer.py:146,152). 
 * N <= 64: head_dim 32 or 64.  64 < N <= 224: head_dim 32, or 64 in bf16 -- the head_dim-64 instances (whole ViT crops: one window per
 * image, N < ws * ws allowed, zero table) leave dbias_ws unwritten.
 */
int esvit_window_attn_fwd(int dtype, const void* qkv, const float* qkv_bias, const int32_t* win2tok, int L,
                          const float* rel_table, int ws, float* bias_frag_ws, const int32_t* region_ids, int nW, int nB,
                          int N, int nH, int hd, float scale, void* out, float* lse, float* attn_out,
                          esvit_stream_t stream);
/* dout dtype [nB*L, C] -> dqkv dtype [nB*L, 3C] (every row written).  fwd_out / lse: the forward's outputs (needed for
 * 14x14 windows only).  Partials: dbias_ws fp32 [ESVIT_Q_ATTN_BWD_PARTS(N, nB*nW, nH), nH, frag]
 * (relative-position-bias gradient, reduced by esvit_relpos_bias_bwd) and dpad_ws fp32
 * [ESVIT_Q_ATTN_BWD_PAD_ROWS(N, nB*nW, nH | dtype << 32), 2C], ZERO-INITIALISED by the caller: sums of the dK / dV rows
 * of zero-pad slots, layout [k|v][nH][hd] -- gradients of qkv_bias[C:3C] (column-sum them into the bias gradient). */
int esvit_window_attn_bwd(int dtype, const void* qkv, const float* qkv_bias, const int32_t* win2tok, int L,
                          const void* dout, const void* fwd_out, const float* lse, const float* rel_table, int ws,
                          float* bias_frag_ws, const int32_t* region_ids, int nW, int nB, int N, int nH, int hd,
                          float scale, void* dqkv, float* dbias_ws, float* dpad_ws, esvit_stream_t stream);
/* dtable fp32 [table_rows, nH] (overwritten, or added to when accumulate != 0: the second resolution group of a ragged
 * multi-crop block) = scatter-add over index of sum_parts dbias_ws */
int esvit_relpos_bias_bwd(const float* dbias_ws, int parts, const int64_t* index, int N, int nH,
                          int table_rows, float* dtable, int accumulate, esvit_stream_t stream);

/* ---- DINOHead pieces (vision_transformer.py:414-418) -------------------- */
/* z = x / max(||x||_2, 1e-12) row-wise; x dtype [R, D]; z dtype; inv_norm fp32 [R] */
int esvit_l2norm_fwd(int dtype, const void* x, int64_t R, int D, void* z, float* inv_norm,
                     esvit_stream_t stream);
/* dx = (dz - (dz.z) z) * inv_norm */
int esvit_l2norm_bwd(int dtype, const void* dz, const void* z, const float* inv_norm, int64_t R, int D,
                     void* dx, esvit_stream_t stream);
/* legacy weight_norm(dim=0): w[k,:] = g[k] * v[k,:]/||v[k,:]||; v fp32 [K,D]; g fp32 [K];
 * w, wT in dtype ([K,D] and [D,K], either may be NULL); inv_norm fp32 [K] */
int esvit_weightnorm_fwd(int dtype, const float* v, const float* g, int K, int D, void* w, void* wT,
                         float* inv_norm, esvit_stream_t stream);
/* dv = g*inv*(dw - (dw.vhat) vhat); dg = dw.vhat (dg may be NULL) */
int esvit_weightnorm_bwd(const float* dw, const float* v, const float* g, const float* inv_norm, int K,
                         int D, float* dv, float* dg, esvit_stream_t stream);

/* ---- DINOLoss / DDINOLoss (main_esvit.py:603-770) ----------------------- */
/* per-row stats of the sharpened, centred teacher logits (main_esvit.py:629,694,697):
 * for row r: mx[r] = max_k (t[r,k]-c[k])/temp ; lse[r] = log sum exp(.. - mx) */
int esvit_teacher_row_stats(int dtype, const void* t, const float* center, float inv_temp, int64_t R,
                            int K, float* row_max, float* row_lse, esvit_stream_t stream);
/* region matching (main_esvit.py:735-736), fused argmax (first index on ties) + row assembly for the region loss: sim fp32 [B, S, ld] holds, for image b, the cosine
 * similarities of its S student tokens (all crops, image-major) against the 2*Tt teacher tokens
 * (view 0 then view 1).  For student position s of crop crop_id[s] and teacher view iq:
 *   tmatch[cm_row[b*S+s]*2 + iq] = crop_id[s]==iq ? -1 : iq*B*Tt + b*Tt + argmax_j sim[b,s,iq*Tt+j]
 * cm_row maps the image-major position to the student's crop-major logit row (main_esvit.py:710-715). */
int esvit_region_match(const float* sim, int B, int S, int Tt, int ld, const int32_t* crop_id,
                       const int32_t* cm_row, int32_t* tmatch, esvit_stream_t stream);
/* fused student log-softmax CE + gradient (main_esvit.py:706-746).
 * s dtype [Rs, K] student logits (un-tempered); for student row r the teacher rows it is
 * scored against are tmatch[r*2+0], tmatch[r*2+1] (row index into t, -1 = unused term).
 * row_w[r] = weight of each term of that row (0.5/(n_terms*B) or 0.5/(n_terms*B*Ts), 1/.. for DINOLoss); terms = 2.
 * Mixup targets (DINOLoss, main_esvit.py:639-641): terms = 4, tmatch[r*4+j] and an individual weight term_w[r*4+j] per term
 * (row_w unused): row r's loss is sum_j w_j CE(teacher row j, student row r).  term_w = NULL selects the two-term form.
 * Outputs: row_loss fp32 [Rs] (sum over the row's terms, already weighted), ds dtype [Rs, K]
 * = d loss / d s (includes 1/student_temp).
 * row_order (optional, int32 [Rs], a permutation of the rows): the order in which workgroups take the rows -- the region loss
 * passes the image-major order so that the student rows of one image, which share their <= 98 teacher rows, run together and
 * re-read them from cache instead of HBM.  Results do not depend on it.
 * s_row_max / s_row_lse (optional pair, fp32 [Rs]): statistics of z = s / tau_s already known (esvit_rowstat_combine of the
 * last-layer GEMM's side output): lse(z) = s_row_max + s_row_lse and the kernel's first pass over the row is skipped. */
int esvit_dino_ce_fwd_bwd(int dtype, const void* s, const void* t, const float* center,
                          const float* t_row_max, const float* t_row_lse, const int32_t* tmatch,
                          const float* row_w, int terms, const float* term_w, float inv_student_temp,
                          float inv_teacher_temp, int64_t Rs, int K, float* row_loss, void* ds, const int32_t* row_order,
                          const float* s_row_max, const float* s_row_lse, esvit_stream_t stream);
/* fold esvit_gemm_desc::rowstat (fp32 [R, nblocks, 2], base-2 block statistics) into natural-log row statistics, the outputs of
 * esvit_teacher_row_stats: row_max[r] = max_k z_k, row_lse[r] = log sum_k exp(z_k - row_max[r]) */
int esvit_rowstat_combine(const float* rowstat, int64_t R, int nblocks, float* row_max, float* row_lse, esvit_stream_t stream);
/* deterministic sum of row_loss -> loss[0] */
int esvit_sum_f32(const float* x, int64_t n, float* out, esvit_stream_t stream);
/* ds *= scale[0] (scale on device: grad_output of the scalar loss) */
int esvit_scale_inplace(int dtype, void* x, int64_t n, const float* scale, esvit_stream_t stream);
/* center EMA (main_esvit.py:651-660, 752-770): c = c*m + (1-m) * colsum / denom */
int esvit_center_ema(float* center, const float* colsum, float momentum, float denom, int K,
                     esvit_stream_t stream);

/* ---- fused update: per-parameter clip + optimizer rule + teacher EMA -------------
 * utils.py:106-115 (clip), the optimizers of main_esvit.py:408-415 as driven by :506-510,574, EMA main_esvit.py:587-590.
 *   ESVIT_RULE_ADAMW  torch.optim.AdamW          (beta1, beta2, eps as named)
 *   ESVIT_RULE_SGD    torch.optim.SGD(momentum)  (beta1 = momentum; beta2, eps unused): mu = momentum mu + (c g + wd p)
 *   ESVIT_RULE_LARS   utils.LARS (utils.py:519-557) (beta1 = momentum, beta2 = eta): mu = momentum mu + q (c g + wd p),
 *                     q = eta |p| / |c g + wd p| for tensors of group 0 (ndim != 1), 1 otherwise;  p -= lr mu
 * (c = the clip factor of the tensor).  Tensor table (device, int64[ntensors*12]):
 *   [p, g, exp_avg (SGD: momentum_buffer, LARS: mu), exp_avg_sq (AdamW only), teacher_p (0 = none), numel,
 *    group (0: weight decay, 1: none), flags (bit0: has gradient; otherwise only the EMA is applied),
 *    bits(1-beta1^t) | bits(1-beta2^t) << 32 (AdamW), reserved,
 *    bf16 copy of p (0 = none), bf16 copy of teacher_p (0 = none)]   -- the copies are refreshed in the same pass
 * chunk table (device, int32[nchunks*2]): [tensor_id, chunk_index], chunk =
 * esvit_query(ESVIT_Q_UPDATE_CHUNK_ELEMS, 0, 0, 0) elements.
 * esvit_grad_sqnorm: stats = 1: sqnorms fp32 [ntensors] = sum g^2;  stats = 3 (LARS): fp32 [ntensors*3] =
 * (sum g^2, sum p^2, sum g p) per tensor.  esvit_fused_clip_update_ema reads the layout its rule needs.
 * Non-finite guard: if ANY statistic is NaN / inf (a non-finite loss poisons every gradient) the update launch is a no-op --
 * student, moments, teacher and weight copies keep their values (the reference exits before its update, main_esvit.py:546-551);
 * `skipped` (device int32, may be NULL) is then incremented by one, so that a host that looks at the loss only now and then can still
 * count the updates that did not happen (a gradient overflow with a finite loss) and correct its step counters. */
#define ESVIT_RULE_ADAMW 0
#define ESVIT_RULE_SGD 1
#define ESVIT_RULE_LARS 2
int esvit_grad_sqnorm(const int64_t* tensors, int ntensors, const int32_t* chunks, int nchunks, int stats,
                      float* sqnorms, esvit_stream_t stream);
int esvit_fused_clip_update_ema(int rule, const int64_t* tensors, int ntensors, const int32_t* chunks, int nchunks,
                                const float* sqnorms, float clip, float lr, float wd, float beta1,
                                float beta2, float eps, float ema_m, int32_t* skipped, esvit_stream_t stream);

/* ---- CvT backbone pieces (BASELINE config 5) ------------------------------
 * Token-major NHWC activations.  cvt_v4_transformer.py:349-382 (ConvEmbed), :75-105 (DepthWiseConv2d = dw 3x3 +
 * BatchNorm2d + 1x1), :44-46 (QuickGELU: see ESVIT_EPI_QGELU).
 * esvit_conv_im2col: cols[(b,oy,ox)][(ky*k+kx)*Cin + c] = src(b, oy*stride-pad+ky, ox*stride-pad+kx, c), zero outside
 *   the image and in the tail columns [k*k*Cin, Kpad); src = fp32 NCHW images (nchw=1) or activation-dtype NHWC tokens.
 * esvit_conv_col2im: the adjoint, dsrc fp32 NHWC [nB,H,W,Cin].
 * esvit_dwconv3x3: y[b,y,x,c] = sum_t w[c][t] x[b,y+ky-1,x+kx-1,c] (stride 1, zero pad 1); flip=1 applies the taps
 *   mirrored (the data gradient).  esvit_dwconv3x3_wgrad: dw[c][t]; ws: fp32 [ESVIT_Q_COL_REDUCE_BLOCKS(rows)*9*C].
 * esvit_col_sums2: out[0..C) = sum_r a[r][c], out[C..2C) = sum_r a[r][c]*b[r][c]  (BatchNorm statistics: b = a;
 *   BatchNorm backward: a = dy, b = pre-norm activations); ws: fp32 [ESVIT_Q_COL_REDUCE_BLOCKS(rows)*2*C].
 * esvit_col_affine2: act 0: y = a1[c]*x1 + a2[c]*x2 + a3[c]  (x2 may be null);
 *   act 1: y = GELU(a1[c]*x1 + a3[c]) and act 2: y = x2 * GELU'(a1[c]*x1 + a3[c]) -- BatchNorm1d + GELU of DINOHead(use_bn=True)
 *   (vision_transformer.py:391-402) and its backward, the normalised value rebuilt from the pre-norm activation;
 *   act 3: y = max(a1[c]*x1 + a3[c], 0) and act 4: y = x2 where a1[c]*x1 + a3[c] > 0, else 0 -- BatchNorm2d + ReLU of the
 *   residual stem (cvt_v4_transformer.py:385-430) and its backward. */
int esvit_conv_im2col(int dtype, const void* src, int nchw, int nB, int H, int W, int Cin, int k, int stride, int pad,
                      int Ho, int Wo, int Kpad, void* cols, esvit_stream_t stream);
int esvit_conv_col2im(int dtype, const void* dcols, int nB, int H, int W, int Cin, int k, int stride, int pad, int Ho,
                      int Wo, int Kpad, float* dsrc, esvit_stream_t stream);
int esvit_dwconv3x3(int dtype, const void* x, const float* w, int flip, int nB, int H, int W, int C, void* y,
                    esvit_stream_t stream);
int esvit_dwconv3x3_wgrad(int dtype, const void* x, const void* dy, int nB, int H, int W, int C, float* dw, float* ws,
                          esvit_stream_t stream);
int esvit_col_sums2(int dtype, const void* a, const void* b, int64_t rows, int C, float* out, float* ws,
                    esvit_stream_t stream);
int esvit_col_affine2(int dtype, const void* x1, const void* x2, int64_t rows, int C, const float* a1, const float* a2,
                      const float* a3, int act, void* y, esvit_stream_t stream);
/* token grid [nB,Hs,Ws,C] -> [nB,Hd,Wd,C]: zero-pad at the bottom / right (F.pad of cvt_v4_transformer.py:173) or crop (:216) */
int esvit_pad_crop_tokens(int dtype, const void* src, int nB, int Hs, int Ws, int Hd, int Wd, int C, void* dst,
                          esvit_stream_t stream);
/* BatchNorm2d coefficient vectors (nn.BatchNorm2d of cvt_v4_transformer.py:95; eps 1e-5, momentum 0.1).
 * fwd:  sums = [sum d | sum d^2] over n positions (already summed over the ranks for SyncBatchNorm) ->
 *       coef fp32 [4*C] = [a | shift | mean | rstd], y = a*d + shift; running_mean / running_var (both or neither) are
 *       updated in place with the unbiased variance.   eval: the same coef from the running statistics.
 * bwd:  esvit_bn_bwd_local: sums = [sum dy | sum dy*d] -> red = [sum dy | sum dy*xhat] (= this rank's d beta | d gamma);
 *       esvit_bn_bwd_coeffs: red summed over the ranks (NULL = eval statistics) -> abc = [A | B | C], d(d) = A*dy + B*d + C. */
int esvit_bn_fwd_coeffs(const float* sums, float n, const float* gamma, const float* beta, float eps, float momentum,
                        float* running_mean, float* running_var, int C, float* coef, esvit_stream_t stream);
int esvit_bn_eval_coeffs(const float* running_mean, const float* running_var, const float* gamma, const float* beta,
                         float eps, int C, float* coef, esvit_stream_t stream);
int esvit_bn_bwd_local(const float* sums, const float* coef, int C, float* red, esvit_stream_t stream);
int esvit_bn_bwd_coeffs(const float* red, float n, const float* gamma, const float* coef, int C, float* abc,
                        esvit_stream_t stream);

/* ---- crop producer ------------------------------------------------------ */
/* DataAugmentationDINO (datasets/build.py:203-261; utils.py:43-75 GaussianBlur / Solarization) for n crops of ONE output size S,
 * given their random draws: img.crop(box).resize((S, S), BICUBIC) -> horizontal flip -> ColorJitter (Brightness / Contrast /
 * Color / hue in the drawn order) -> grayscale -> GaussianBlur -> solarize -> ToTensor -> Normalize(ImageNet mean / std).
 * Bit-exact with Pillow's 8-bit arithmetic (Resample.c, Blend.c, Convert.c, BoxBlur.c) at every stage.
 *   src     uint8, decoded RGB images, HWC, packed back to back (the buffer must be readable up to the next 4-byte boundary
 *           past its last pixel);  images  int64 [n_img, 3] = (byte offset into src, H, W)
 *   params  int32 [n, ESVIT_AUG_PARAM_INTS], one row per crop:
 *           0 image row | 1 top | 2 left | 3 h | 4 w  (crop box, inside the image) | 5 flip
 *           6..9  jitter operations in application order: 0 brightness, 1 contrast, 2 saturation, 3 hue, -1 none
 *           10 brightness | 11 contrast | 12 saturation factor (float bits) | 13 hue shift = uint8(hue_factor * 255)
 *           14 grayscale | 15 box radius + 1 of the blur (0 = no blur) | 16 ww | 17 fw (BoxBlur.c fixed-point weights of the
 *           Gaussian radius) | 18 solarize | 19.. reserved (0)
 *   n       crops in this call, <= 65535;  S  output size, a multiple of 4, <= 280;  max_h, max_w  the largest box of the n crops (sizes the LDS of the resize;
 *           <= esvit_query(ESVIT_Q_AUG_MAX_BOX, S, 0, 0))
 *   planes  uint8 scratch of n * (3 S^2 + 4) bytes: on return its first n * 3 S^2 bytes are the resized, flipped crops
 *           [n, 3, S, S] before the jitter;  out  fp32 [n, 3, S, S] */
#define ESVIT_AUG_PARAM_INTS 24
int esvit_aug_crops(const uint8_t* src, const int64_t* images, const int32_t* params, int n, int S, int max_h, int max_w,
                    uint8_t* planes, float* out, esvit_stream_t stream);

/* ---- global attention of the monolithic ViT backbones --------------------- */
/* models/vision_transformer.py:67-94 (Attention.forward of deit_tiny / deit_small / vit_base): the score matrix lives in HBM
 * and the products run on esvit_gemm (batch = B * nH); these are the layout changes and the row softmax around them.
 *   esvit_heads_split   x [B * N, parts * nH * hd] token-major (the qkv GEMM output: parts = 3; a gradient of the merged heads:
 *                       parts = 1)  ->  y [parts, B, nH, Np, hd], rows N..Np-1 zero        (:76, reshape + permute)
 *   esvit_heads_merge   the inverse, dropping the pad rows                                   (:83, transpose + reshape)
 *   esvit_softmax_rows_fwd   in place over s [batch, Np, Np]: softmax(scale * s[:, :N, :N]) along the last axis, zero on pad rows
 *                       and columns (Np <= 256)                                              (:79-80)
 *   esvit_softmax_rows_bwd   in place over dp: scale * p o (dp - sum_j p_j dp_j), zero on pad rows and columns */
int esvit_heads_split(int dtype, const void* x, int B, int N, int Np, int nH, int hd, int parts, void* y, esvit_stream_t stream);
int esvit_heads_merge(int dtype, const void* y, int B, int N, int Np, int nH, int hd, int parts, void* x, esvit_stream_t stream);
int esvit_softmax_rows_fwd(int dtype, void* s, int64_t batch, int N, int Np, float scale, esvit_stream_t stream);
int esvit_softmax_rows_bwd(int dtype, const void* p, void* dp, int64_t batch, int N, int Np, float scale, esvit_stream_t stream);
/* Vision Longformer's sliding-chunk attention (models/vision_longformer.py AttnBlock 'longformerhand' -> layers/longformer2d.py:138-262
 * with layers/slidingchunk_2d.py, mode 0, exact 0, rpe off, shared global weights) on the same route: the row softmax restricted to
 * the keys a query may see.  chunk int32 [N]: -1 for a global token, else (chunk row << 16) | chunk column of the token's w x w
 * chunk; a query sees all global tokens and the local tokens of its own and the eight adjacent chunks (the reference's zero-padded
 * and out-of-range positions are exactly the ones that do not exist here); a global query sees every token (longformer2d.py:310-327).
 * Masked entries of P are zero, so the backward needs no mask for correctness.  nglo / chunk_row_tokens (optional, 0 = unknown)
 * describe the token order -- nglo global tokens, then the local tokens chunk row by chunk row, chunk_row_tokens (= w * grid width)
 * per chunk row: the kernels then read only the global columns and the three chunk rows around the query's own (everything else is
 * written as zero unread). */
int esvit_softmax_rows_chunked_fwd(int dtype, void* s, int64_t batch, int N, int Np, float scale, const int32_t* chunk, int nglo,
                                   int chunk_row_tokens, esvit_stream_t stream);
int esvit_softmax_rows_chunked_bwd(int dtype, const void* p, void* dp, int64_t batch, int N, int Np, float scale, const int32_t* chunk,
                                   int nglo, int chunk_row_tokens, esvit_stream_t stream);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* ESVIT_HIP_H */
