/* libe4t_hip.so — C ABI of the MI355X-native (gfx950) E4T training hot path.
 *
 * The reference (mkshing/e4t-diffusion) is pure Python; its "FFI" for this path is the set of
 * torch ops its modules call.  Each entry point below replaces those call sites (cited as
 * reference file:line, relative to the reference root) with one hand-written HIP kernel family.
 * The Python modules in e4t-diffusion_amd/e4t/ bind these symbols with ctypes and pass
 * tensor.data_ptr() / torch.cuda.current_stream().cuda_stream.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless it is a descriptor struct passed by host pointer;
 *   - the caller owns every buffer, including workspaces; the library keeps no tensor state;
 *   - activations: bf16, NHWC, i.e. row-major (B*H*W, C); statistics / grads of parameters: fp32;
 *   - all launches are asynchronous on the caller's stream (hipStream_t passed as void*);
 *   - return value 0 = OK, negative errno-style code otherwise (-22 bad argument, -12 workspace,
 *     -5 launch failure); e4t_last_error() returns the message; nothing throws across the ABI;
 *   - one host thread per process/rank; re-entrant across streams.
 */
#ifndef E4T_HIP_H
#define E4T_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* e4t_stream; /* hipStream_t */

int e4t_version(void);
/* bit set of how the library was built: E4T_BUILD_EXPERIMENTAL = the measured-and-rejected GEMM tile variants (codes 3xxx / 4xxx, 5064,
 * 5128, 256 with 64-wide K-tiles, 640, 1128 / 1160) are compiled in; the default build answers 0 and maps those codes to product tiles */
#define E4T_BUILD_EXPERIMENTAL 1
int e4t_build_flags(void);
const char* e4t_last_error(void);
/* device sanity: returns 0 and fills arch name (e.g. "gfx950"), CU count */
int e4t_device_info(char* arch, int arch_len, int* cu_count);
/* diagnostics: write one text line per kernel launch (kernel symbol | shape | algorithmic bytes | flops) to `path`
 * (NULL or "" stops).  Also enabled by the environment variable E4T_LAUNCH_LOG=<path>.  Used by tools/roofline_report.py to
 * join a rocprofv3 kernel trace / PMC run of the same process per SHAPE; the reference has no counterpart. */
int e4t_set_launch_log(const char* path);

/* ---------------------------------------------------------------- GEMM / conv (gemm.hip) ---- */
/* epilogue flags */
#define E4T_OUT_F32 1   /* C stored as fp32 (default bf16) */
#define E4T_RES_F32 2   /* residual is fp32 (default bf16) */
#define E4T_ACT_GELU 4  /* exact-erf GELU after bias, before residual */
#define E4T_ACCUM 8     /* C += result (read-modify-write in C's dtype) */
#define E4T_REDUCE_BATCH 16 /* sum the batch entries into ONE C (needs workspace) */

/* C[M,N] = epi(alpha * A[M,K] . B[N,K]^T): F.linear / 1x1 conv / their dX and dW GEMMs.
 * Replaces cross_attention.py:506,516,518,534 ; attention.py:376,419 ; transformer_2d.py:153,205,
 * 258-261 ; encoder.py:101-106,159-168 ; [3P] open_clip ViT linears ; time_emb_proj ; conv_shortcut. */
typedef struct {
  const void* A;       /* bf16 [M][lda] */
  const void* A2;      /* optional second K-source: columns [K1,K) come from A2[M][lda2] (fused torch.cat) */
  const void* B;       /* bf16 [N][ldb] */
  void* C;             /* bf16 or fp32 [M][ldc] */
  const float* bias;   /* optional fp32 [N] */
  const void* residual;/* optional [M][ldr], added after activation */
  const float* rowbias;/* optional fp32 [M/rows_per_batch][ldrb] (ResBlock time-embedding add) */
  void* workspace;     /* fp32 scratch for split-K / batch reduction, or NULL */
  size_t workspace_bytes;
  int M, N, K, K1;
  int lda, lda2, ldb, ldc, ldr;
  int rows_per_batch;
  int flags;
  int tile;            /* 0 = auto; 64, 128, 160 (= 128x160), 256 (= 256x128), 512 (= 256x256 ping-pong), 640 (= 512x128),
                          1128 / 1160 (= persistent streaming 256x128 / 256x160, gemm_ps.hip; needs K % 64 == 0, batch 1, no split-K);
                          + 3000 / 4000 forces 3 / 4 LDS stages on the 64 / 128 / 160 tiles (e.g. 3128); 5064 / 5128 / 5256 = the 64 / 128 /
                          256x128 tiles with 32-wide K-tiles (5256: three 24-KiB stages, two workgroups per CU — the automatic choice
                          for tall outputs with N % 128 == 0); 2320 = 256x320 ping-pong with k-step phases (gemm_pq_kernel: N % 320 == 0,
                          K % 64 == 0, plain bias / row-bias / residual epilogue — the automatic choice where its rounds are full).
                          An unsupported code for the shape falls back to the nearest tile that fits; e4t_gemm_plan reports the choice. */
  int splitk;          /* 0 = auto, >= 1 forced */
  int batch;           /* >= 1; operand base pointers advance by the strides below (elements) */
  long long strideA, strideB, strideC, strideBias;
  float alpha;
  int ldrb;            /* row stride of rowbias (elements); 0 = N */
  float* colstats;     /* optional out: fp32 [M/32][N][2] = per 32-row block (sum, sum of squares) of every output column, for the
                          GroupNorm that consumes C (e4t_groupnorm_fwd_cs).  Produced only by the single-pass bf16 epilogue
                          (M % 32 == 0, batch 1, no split-K): the call returns 1 when it was written, 0 otherwise */
  int panel_rows;      /* 0 = dense.  > 0: the M logical rows are panels of panel_rows rows (a multiple of 256, dividing M) that lie */
  int panel_stride;    /* panel_stride rows apart, the first at row panel_off, in A, C and residual alike: the 16 x 256 patch tokens  */
  int panel_off;       /* of a [16][257] ViT token matrix (panel_rows 256, stride 257, offset 1) run as 16 full 256-row tiles instead of
                          17 ragged ones ([3P] open_clip ViT linears, encoder.py:154).  N % 320 == 0, K % 64 == 0; no batch / row bias /
                          colstats / A2 / accumulate / split-K */
} e4t_gemm_desc;
/* returns 0 (or 1, see colstats) on success, a negative errno-style code on error */
int e4t_gemm_nt(const e4t_gemm_desc* d, e4t_stream stream);
/* C[M,N] = epi(alpha * A^T . B) with A = bf16 [K][lda >= M], B = bf16 [K][ldb >= N]: the contraction runs over the ROWS of both
 * operands — the weight gradient dW = dY^T . X of every linear layer (autograd of cross_attention.py:506-518, attention.py:376,419,
 * ...) without transposing dY and X first.  Same descriptor; A2 / rowbias / batch are not supported; split-K is automatic
 * (workspace >= splitk * M * N * 4 bytes, falls back to one pass when it is smaller). */
int e4t_gemm_tn(const e4t_gemm_desc* d, e4t_stream stream);

/* 3x3 convolution, pad 1, NHWC, implicit GEMM (no im2col buffer).
 * Replaces [3P diffusers 0.14] ResnetBlock2D.conv1/conv2, Downsample2D.conv (stride 2),
 * Upsample2D (nearest x2 + conv) built at unet_2d_blocks.py:481,760,804,881,1732,1774,1855,1872 and
 * conv_in / conv_out (unet_2d_condition.py:106-108,285-287), plus their data gradients. */
#define E4T_CONV_S1 1   /* stride 1 (also dgrad of stride 1 with flipped weights) */
#define E4T_CONV_S2 2   /* stride 2 */
#define E4T_CONV_UP2 3  /* nearest x2 upsample fused into the gather, then stride 1 */
#define E4T_CONV_S2T 4  /* transposed stride 2 (dgrad of E4T_CONV_S2) */
#define E4T_CONV_S2A 5  /* stride 2 with pad (0,1,0,1): [3P] AutoencoderKL encoder Downsample2D(padding=0) */
typedef struct {
  const void* X;       /* bf16 [B][Hin][Win][Cin] */
  const void* W;       /* bf16 [Cout][3][3][Cin]  (e4t_conv_weight_prepare) */
  void* Y;             /* bf16/fp32 [B][Hout][Wout][Cout] */
  const float* bias;   /* fp32 [Cout] or NULL */
  const void* residual;/* [B][Hout][Wout][Cout] or NULL */
  const float* rowbias;/* fp32 [B][ldrb] or NULL */
  void* workspace;
  size_t workspace_bytes;
  int B, Hin, Win, Cin, Hout, Wout, Cout;
  int mode, flags, tile, splitk;
  int ldrb;            /* row stride of rowbias (elements); 0 = Cout.  Lets all ResBlocks' time-embedding projections
                          live in one (B, sum Cout) matrix produced by a single GEMM */
  float* colstats;     /* optional out, as in e4t_gemm_desc (M = B*Hout*Wout, N = Cout) */
} e4t_conv_desc;
int e4t_conv3x3(const e4t_conv_desc* d, e4t_stream stream);

/* What the launcher will do for a descriptor, WITHOUT launching: the tile it picks (codes as e4t_gemm_desc.tile; tile_m x tile_n
 * are its dimensions), the split-K factor, and the fp32 workspace (bytes) the call wants for it — e4t_gemm_nt / e4t_conv3x3 fall
 * back to a single pass when handed less in auto mode, and fail with -12 when split-K or REDUCE_BATCH was requested explicitly.
 * The pointer fields of the descriptor are never dereferenced, but whether A2 / rowbias / residual / colstats are NULL or not is part of the
 * decision (a two-source A and a row bias restrict the tiles; wanted column statistics price the split-K variants, which leave none): leave them NULL or non-NULL exactly as the later launch will have
 * them (any non-NULL value does).  Shapes, strides, flags, tile, splitk and batch are read as given.
 * This is the one statement of the tile / split-K heuristic: callers size workspaces and label timings from it (SURVEY §8b
 * "query size via e4t_<op>_workspace_bytes"). */
typedef struct {
  int tile, tile_m, tile_n;
  int splitk;
  size_t workspace_bytes;
  /* > 0: the last `tail_rows` (<= 32) rows of a dense GEMM whose M is a multiple of 128 plus a few rows (the CLIP-ViT's 16 x 257 = 4112
   * token rows, [3P] open_clip via e4t/encoder.py:154) are computed by a tail stage at the end of the same launch and the tile, split-K and
   * workspace above are those of the first M - tail_rows rows (gemm.hip: plan_gemm_tail) */
  int tail_rows;
  /* LDS stages (2 - 4) of the 64 / 128 / 160 tiles — a template argument of the kernel symbol the launch will show in a trace (round 6; the field was
   * `reserved` before); 0 for every other tile */
  int stages;
} e4t_gemm_plan_t;
int e4t_gemm_plan(const e4t_gemm_desc* d, e4t_gemm_plan_t* out);
int e4t_gemm_tn_plan(const e4t_gemm_desc* d, e4t_gemm_plan_t* out);
int e4t_conv3x3_plan(const e4t_conv_desc* d, e4t_gemm_plan_t* out);

/* ---------------------------------------------------------------- attention (attention.hip) -- */
/* O = softmax(Q K^T * scale) V ; Q/K/V/O are (B, T|S, heads*DH) bf16 matrices with row strides ld*
 * and batch strides b* (elements); lse: fp32 [B][H][T] (log2 units) or NULL.  DH in {32,40,64,80,160}.
 * Replaces cross_attention.py:521-531 (SDPA), :222-251,313-314 (math), :473-481 (xFormers). */
int e4t_attention_fwd(const void* Q, const void* K, const void* V, void* O, float* lse, int B, int H, int T, int S, int DH,
                      int ldq, int ldk, int ldv, int ldo, long long bq, long long bk, long long bv, long long bo,
                      float scale, int causal /* keys > query masked (CLIP text encoder) */, e4t_stream stream);
/* delta_ws: fp32 [B][H][T] scratch.  dQ/dK/dV share the layout (strides) of Q/K/V; dO that of O. */
int e4t_attention_bwd(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* lse,
                      float* delta_ws, void* dQ, void* dK, void* dV, int B, int H, int T, int S, int DH, int ldq, int ldk,
                      int ldv, int ldo, long long bq, long long bk, long long bv, long long bo, float scale, int causal,
                      e4t_stream stream);
/* The same with a workspace of ws_floats >= e4t_attention_bwd_workspace_floats(B, H, T, S, DH) fp32 (16-byte aligned): with few
 * keys (cross-attention over the 77 text tokens, cross_attention.py:516-531) the dK/dV kernel then cuts the query range into chunks
 * that run as separate workgroups and a second kernel sums their fp32 partials in a fixed order (deterministic).  A workspace of
 * only B*H*T floats gives the un-split kernel of e4t_attention_bwd.  For an un-split query range the stated size is 3*B*H*T + 4: Delta plus
 * the {L, Delta} pairs the dQ kernel leaves for the dK/dV kernel's LDS-DMA staging (dh 40; round 6). */
size_t e4t_attention_bwd_workspace_floats(int B, int H, int T, int S, int DH);
int e4t_attention_bwd_ws(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* lse,
                         float* ws, size_t ws_floats, void* dQ, void* dK, void* dV, int B, int H, int T, int S, int DH, int ldq,
                         int ldk, int ldv, int ldo, long long bq, long long bk, long long bv, long long bo, float scale,
                         int causal, e4t_stream stream);

/* ---------------------------------------------------------------- norms (norm.hip) ----------- */
/* GroupNorm over NHWC input given as up to two channel-sources (x1: C1 ch, x2: C2 ch or NULL) —
 * the fused form of torch.cat([h, skip], 1) -> GroupNorm -> SiLU (unet_2d_blocks.py:1795,1883;
 * [3P] ResnetBlock2D.norm1/norm2; transformer_2d.py:149,253; unet_2d_condition.py:275-278,554-556). */
int e4t_groupnorm_num_chunks(int B, int HW);
size_t e4t_groupnorm_workspace_bytes(int B, int HW, int C, int G, int with_param_grads);
int e4t_groupnorm_stats(const void* x1, int C1, const void* x2, int C2, int B, int HW, int G, float eps,
                        float* mean_rstd /* [B][G][2] */, void* workspace, size_t ws_bytes, e4t_stream stream);
int e4t_groupnorm_apply(const void* x1, int C1, const void* x2, int C2, const float* mean_rstd, const float* gamma,
                        const float* beta, void* y /* bf16 [B*HW][C1+C2] */, int B, int HW, int G, int silu,
                        e4t_stream stream);
/* stats + apply in two launches (the apply kernel finalises mean / rstd from the chunk partials itself and writes them to
 * mean_rstd for the backward).  workspace: e4t_groupnorm_workspace_bytes(B, HW, C, G, 0). */
int e4t_groupnorm_fwd(const void* x1, int C1, const void* x2, int C2, const float* gamma, const float* beta, void* y,
                      float* mean_rstd /* out [B][G][2] */, int B, int HW, int G, float eps, int silu, void* workspace,
                      size_t ws_bytes, e4t_stream stream);
/* Same result, with the statistics pass replaced by a reduction of the column statistics cs1 / cs2 ([B*HW/32][C1|C2][2]) that the
 * GEMM / conv which produced x1 / x2 wrote in its epilogue (e4t_gemm_desc.colstats): the activation is read once, not twice. */
int e4t_groupnorm_fwd_cs(const void* x1, int C1, const float* cs1, const void* x2, int C2, const float* cs2, const float* gamma,
                         const float* beta, void* y, float* mean_rstd, int B, int HW, int G, float eps, int silu, void* workspace,
                         size_t ws_bytes /* e4t_groupnorm_workspace_bytes(B, HW, C, G, 0) */, e4t_stream stream);
/* dx1|dx2 = d/dx of act(GN(x)) given dy, plus the optional gradients that reach x1 / x2 through another consumer (the
 * ResBlock shortcut / residual): add1 bf16 [B*HW][C1], add2 bf16 [B*HW][C2], either may be NULL; optional per-chunk
 * channel partials [B][chunks][C][2] = (sum dz, sum dz*xhat) for dbeta/dgamma. */
int e4t_groupnorm_bwd(const void* x1, int C1, const void* x2, int C2, const void* dy, const float* mean_rstd,
                      const float* gamma, const float* beta, const void* add1, const void* add2, void* dx1, void* dx2,
                      float* dgamma_dbeta_partial, int B, int HW, int G, int silu, void* workspace, size_t ws_bytes,
                      e4t_stream stream);
/* LayerNorm over the last dim (attention.py:259,268,273; open_clip ViT ln_*). */
int e4t_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean_rstd /* [M][2] or NULL */,
                      int M, int D, float eps, e4t_stream stream);
/* The same with fp32 input rows (y stays bf16): LayerNorm of an fp32 residual stream — under torch.autocast the CLIP-ViT's
 * ResidualAttentionBlock keeps x = x + attn(ln_1(x)) in fp32 ([3P] open_clip via encoder.py:153-154). */
int e4t_layernorm_fwd_f32(const float* x, const float* gamma, const float* beta, void* y, float* mean_rstd, int M, int D, float eps,
                          e4t_stream stream);
/* dx = LN'(dy) (+ add): `add` (bf16 [M][D] or NULL) is the gradient that reaches x through the residual branch of a
 * pre-LN block, fused here instead of a separate elementwise add. */
int e4t_layernorm_bwd(const void* x, const void* dy, const float* gamma, const float* mean_rstd, const void* add, void* dx,
                      int M, int D, e4t_stream stream);
/* Column reductions over the rows of a (M, C) bf16 matrix (two launches, deterministic).  workspace: fp32
 * [e4t_colreduce_splits(M)][C] (x2 for the LayerNorm variant).  accumulate != 0: out += result (gradient accumulation). */
int e4t_colreduce_splits(int M);
/* out[c] = sum_r x[r][c]: bias gradients of nn.Linear / nn.Conv2d (tuning_e4t.py trains every UNet weight). */
int e4t_colsum(const void* x, int ldx, int M, int C, float* out, int accumulate, void* workspace, size_t ws_bytes, e4t_stream stream);
/* dgamma[c] = sum_r dy * xhat, dbeta[c] = sum_r dy of a LayerNorm over the last dim. */
int e4t_layernorm_param_grad(const void* x, const void* dy, const float* mean_rstd, int M, int D, float* dgamma, float* dbeta,
                             int accumulate, void* workspace, size_t ws_bytes, e4t_stream stream);

/* ---------------------------------------------------------------- weight offsets (wo.hip) ---- */
/* One descriptor per WeightOffsets instance (weightoffsets.py:5-23) + the projection weight it
 * modulates (cross_attention.py:506,516,518).  row = in_features, col = out_features.
 * Descriptors live in DEVICE memory (array of n); wc == NULL marks a plain weight (cast only). */
#define E4T_WO_STORE_F32 1    /* weff is fp32 [col][ld_weff] instead of bf16 */
#define E4T_WO_OFFSETS_ONLY 2 /* weff receives the offsets themselves (WeightOffsets.forward()), not W o (1+offsets) */
typedef struct {
  const float *v, *w1, *b1, *w2, *b2, *wc, *bc, *wr, *br; /* v[1] linear1.{w,b}[row] linear2.{w,b}[col] linear_column[row][row],[row] linear_row[col][col],[col] */
  const float* W;        /* base weight fp32 [col][row] */
  float* vecs;           /* fp32 scratch, e4t_wo_vecs_floats(row, col) */
  float* partial;        /* fp32 scratch, e4t_wo_partial_floats(row, col) */
  void* weff;            /* out: bf16 [col][ld_weff]  W o (1 + offsets)   (may be NULL) */
  void* weffT;           /* out: bf16 [row][ld_weffT] transposed copy      (may be NULL) */
  const float* dweff;    /* in (backward): fp32 [col][ld_dweff] dL/dW_eff */
  float *g_v, *g_w1, *g_b1, *g_w2, *g_b2, *g_wc, *g_bc, *g_wr, *g_br; /* out (backward): parameter grads */
  float* g_W;            /* out (backward, optional): dL/dW = dW_eff o (1 + offsets) */
  int row, col, ld_weff, ld_weffT, ld_dweff;
  int mode;              /* E4T_WO_* bits */
} e4t_wo_desc;
size_t e4t_wo_vecs_floats(int row, int col);
size_t e4t_wo_partial_floats(int row, int col);
int e4t_wo_forward(const e4t_wo_desc* descs_dev, int n, int max_row, int max_col, e4t_stream stream);
int e4t_wo_backward(const e4t_wo_desc* descs_dev, int n, int max_row, int max_col, int accumulate, e4t_stream stream);
int e4t_weight_prepare(const e4t_wo_desc* descs_dev, int n, int max_row, int max_col, e4t_stream stream);
/* OIHW fp32 3x3 weights -> [O][ky][kx][Ipad] (forward) and [I][2-ky][2-kx][Opad] (dgrad), bf16, zero padded */
int e4t_conv_weight_prepare(const float* w_oihw, void* w_fwd, void* w_dgrad, int O, int I, int Ipad, int Opad, e4t_stream stream);

/* ---------------------------------------------------------------- streaming ops (elementwise.hip) */
int e4t_geglu_fwd(const void* u /* [M][2H] */, void* h /* [M][H] */, long long M, int H, e4t_stream stream);  /* attention.py:428-430 */
int e4t_geglu_bwd(const void* u, const void* dh, void* du, long long M, int H, e4t_stream stream);
#define E4T_OP_SILU 0
#define E4T_OP_SILU_BWD 1
#define E4T_OP_GELU 2
#define E4T_OP_GELU_BWD 3
#define E4T_OP_LRELU 4
#define E4T_OP_LRELU_BWD 5
#define E4T_OP_QGELU 6      /* x * sigmoid(1.702 x): CLIPTextModel hidden_act of SD-1.x (modeling_clip.py via transformers) */
#define E4T_OP_QGELU_BWD 7
int e4t_unary(const void* x, const void* dy, void* y, long long n, int op, e4t_stream stream);
int e4t_add(const void* a, const void* b, void* y, long long n, e4t_stream stream);
int e4t_transpose(const void* in, void* out, int batch, int R, int C, int ldi, int ldo, long long bsi, long long bso, e4t_stream stream);
int e4t_sumpool2(const void* in /* [B][2H][2W][C] */, void* out /* [B][H][W][C] */, int B, int H, int W, int C, e4t_stream stream);
int e4t_spatial_mean(const void* x, float* out, int B, int HW, int C, int ldo, int coff, e4t_stream stream);            /* encoder.py:147 */
int e4t_spatial_mean_bwd(const float* g, const void* base, void* dx, int B, int HW, int C, int ldg, int coff, e4t_stream stream);
int e4t_timestep_embedding(const long long* t, void* out /* bf16 [B][dim] = [cos|sin] */, int B, int dim, e4t_stream stream); /* unet_2d_condition.py:461 */
int e4t_clip_preprocess(const float* pixels_nchw, void* patches /* bf16 [B*g*g][Kpad] */, int B, int Hin, int Win, int S, int P, int Kpad, e4t_stream stream); /* encoder.py:131-139 + patchify */
/* Sampling loop glue (pipeline_stable_diffusion_e4t.py:209-214): classifier-free guidance eps = u + g*(c-u) over
 * pred = [uncond | cond] (cfg = 1; cfg = 0: pred is eps) fused with a linear scheduler update
 * out = c_sample*sample + c_pred*eps (+ c_noise*noise, noise may be NULL) — DDIM for epsilon and v prediction is of this
 * form.  coef: DEVICE float[4] = {g, c_sample, c_pred, c_noise} (so a captured graph replays with new values).
 * pred_nhwc = 1: pred is [B(*2)][HW][C] (the UNet's native output); sample/noise/out are [B][C][HW] fp32. */
int e4t_guided_step(const float* pred, const float* sample, const float* noise, float* out, const float* coef,
                    int B, int C, int HW, int cfg, int pred_nhwc, e4t_stream stream);
/* Data path (pretrain_e4t.py:137-144 make_transforms = SmallestMaxSize(interpolation=3: cv2.INTER_AREA) -> RandomCrop ->
 * HorizontalFlip, and :174-177 image/127.5-1, HWC->CHW): a batch of raw decoded uint8 RGB images in one device pool ->
 * out fp32 [B][3][S][S].  table: int64 [B][8] (device) = {byte offset of the image in pool, H, W, newH, newW (the
 * SmallestMaxSize dims), crop y0, crop x0 (in the resized image), flip}.  Byte-exact INTER_AREA (area / area-fast /
 * enlarging fixed-point branches); only the cropped window is computed. */
int e4t_image_prep(const void* pool, const long long* table, float* out, int B, int S, e4t_stream stream);
/* in-place row softmax of a bf16 matrix [rows][ld] over the first L columns (fp32 math); L % 8 == 0, L <= 16384 */
int e4t_softmax_rows(void* x, long long rows, int L, int ld, e4t_stream stream);
/* 3x3/pad-1 im2col of a 3-channel NCHW fp32 image -> bf16 [B*H*W][32], column (ky*3+kx)*3+c, columns 27..31 zero */
int e4t_im2col3_rgb(const float* pixels_nchw, void* out, int B, int H, int W, e4t_stream stream);
/* out[(tap*C + c)][m] = gathered X (3x3 taps, zero outside), m over OUTPUT pixels: B operand of the conv weight-gradient GEMM */
int e4t_im2col_T(const void* x, void* out, int B, int Hin, int Win, int C, int Hout, int Wout, int ldo, int mode, e4t_stream stream);
/* out[m][tap*C + c] (bf16 [B*Hout*Wout][9*C]): with e4t_gemm_tn, dW[co][tap][ci] = dY^T . im2col(X) needs no transposes. */
int e4t_im2col(const void* x, void* out, int B, int Hin, int Win, int C, int Hout, int Wout, int mode, e4t_stream stream);
int e4t_adamw(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps,
              float weight_decay, int step, float grad_scale, e4t_stream stream);                                        /* pretrain_e4t.py:387-392,652 */
/* The same update with the step-dependent scalars in DEVICE memory: hyper_dev = float[4] {lr, 1 - beta1^t, sqrt(1 - beta2^t), grad_scale}
 * (16-byte aligned).  A training step captured into a hipGraph (E4TTrainer step graph) replays this launch unchanged and the host
 * refreshes the four floats before each replay; the arithmetic is that of e4t_adamw with the same values. */
int e4t_adamw_hyper(float* p, const float* g, float* m, float* v, long long n, const float* hyper_dev, float beta1, float beta2, float eps,
                    float weight_decay, e4t_stream stream);
/* AdamW (same arithmetic) of a stack of n [rows][cols] fp32 matrices whose gradient is the rank-K product
 *   dW_i[r][c] = grad_scale * sum_k G[k][r] * Z[k][i*cols + c]        (bf16 factors, fp32 k-ordered fmaf chain)
 * formed in registers and never written: the E4T head's 129 first_linears (encoder.py:108-123,159-162 — dW_i = g^T z_i over the step's
 * images) under pretrain_e4t.py:387-392,652.  G: [K][ldg >= rows], Z: [K][ldz >= n*cols] (row-major bf16; under data parallelism every rank's
 * gathered rows).  hyper_dev != NULL: {lr, 1 - beta1^t, sqrt(1 - beta2^t), grad_scale} are read from device memory as in e4t_adamw_hyper
 * (lr / step / grad_scale arguments ignored).  cols % 4 == 0; p, m, v 16-byte aligned. */
int e4t_adamw_rank(float* p, float* m, float* v, const void* G, const void* Z, int n, int rows, int cols, int K, int ldg, long long ldz,
                   float lr, float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale, const float* hyper_dev,
                   e4t_stream stream);
int e4t_sumsq_partial(const float* g, long long n, float* partial, int nblocks, e4t_stream stream);                      /* tuning_e4t.py:335 grad-norm */

/* ---------------------------------------------------------------- data-parallel collectives (comm.hip) ---------- */
/* An RCCL communicator owned by the library, for a host that does not bring torch.distributed: replaces the DDP reducer accelerate wraps
 * the reference's models with (pretrain_e4t.py:410-412 `accelerator.prepare`, :648 `accelerator.backward`; tuning_e4t.py:197-200, :328).
 * Collectives run IN PLACE on a library-owned high-priority stream: each call first orders that stream after everything queued so far on
 * the caller's `stream` (the kernels that produced the buffer), and returns without waiting; e4t_comm_wait(comm, stream) makes `stream`
 * wait — on the device, no host sync — for every collective issued so far.  One communicator per process (= per GPU); calls on one
 * communicator come from one host thread.  librccl.so is resolved at first use from the copy already mapped into the process (torch's),
 * else loaded by name (E4T_RCCL_LIB overrides); without one every e4t_comm_* call returns -38.  The Python host in this repository
 * uses torch.distributed by default and this path with E4TTrainer(collectives="library"). */
typedef struct e4t_comm* e4t_comm_t;
#define E4T_COMM_F32 0
#define E4T_COMM_BF16 1
#define E4T_COMM_SUM 0
#define E4T_COMM_AVG 1
#define E4T_COMM_MIN 2
#define E4T_COMM_MAX 3
int e4t_comm_unique_id(void* id128);                 /* rank 0: 128 bytes the launcher hands to every rank (ncclGetUniqueId) */
int e4t_comm_init(e4t_comm_t* comm, const void* id128, int rank, int world);   /* collective over all ranks; current HIP device */
int e4t_comm_allreduce(e4t_comm_t comm, void* buf, long long count, int dtype, int op, e4t_stream stream);
int e4t_comm_allgather(e4t_comm_t comm, const void* send, void* recv /* world * count */, long long count, int dtype, e4t_stream stream);
int e4t_comm_wait(e4t_comm_t comm, e4t_stream stream);
int e4t_comm_info(e4t_comm_t comm, int* rank, int* world, e4t_stream* comm_stream);
int e4t_comm_destroy(e4t_comm_t comm);

/* ---------------------------------------------------------------- probe (probe.hip) ---------- */
/* writes, for lane l and register r of v_mfma_f32_32x32x16_bf16 with A[i][k] = i*16+k... see probe.hip */
int e4t_probe_mfma_layout(float* out_rows /* [64][16] */, float* out_cols /* [64][16] */, e4t_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* E4T_HIP_H */
