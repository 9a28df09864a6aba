/*
 * ga_dit.h -- C-ABI of the MI355X-native DiT/SiT denoiser forward (bf16 MFMA, fp32 residual stream).
 *
 * Drop-in boundary for the denoise half of GaussianAnything's render-and-denoise hot path: one call of
 * ga_dit_forward() computes what the reference computes in
 *     DiT_I23D_PCD_PixelArt_noclip.forward              /root/reference/dit/dit_i23d.py:511-567   (stage 1)
 *     DiT_I23D_PCD_PixelArt_noclip_clay_stage2.forward  /root/reference/dit/dit_i23d.py:707-750   (stage 2)
 * i.e. everything torchdiffeq calls per function evaluation through
 *     transport/integrators.py:104-107 -> Transport.velocity_ode (transport/transport.py:209-218) -> model(x, t, **kw).
 * The block arithmetic replaces the third-party kernels the reference reaches through
 *     xformers.ops.memory_efficient_attention   vit/vision_transformer.py:297, ldm/modules/attention.py:538-548
 *     xformers FusedMLP                         dit/dit_models_xformers.py:281-286
 *     RMSNorm (apex or dit/norm.py:29-43), t2i_modulate (dit_models_xformers.py:53-54)
 * The per-op entry points (ga_gemm_bf16, ga_attention_bf16, ...) are exported as well: the parity tests call them
 * one by one, and a maintainer can bind them individually.
 *
 * Conventions: every pointer is a DEVICE pointer unless it says "host"; bf16 tensors are raw uint16 (upper half of the
 * fp32 bit pattern, round-to-nearest-even); row-major; all work is enqueued on `stream` (hipStream_t as void*), no
 * host sync, no allocation; return 0 or a negative GA_DIT_ERR_* code.
 */
#ifndef GA_DIT_H
#define GA_DIT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GA_DIT_OK 0
#define GA_DIT_ERR_NULL_ARG (-1)
#define GA_DIT_ERR_BAD_SHAPE (-2) /* K % 64 != 0, N % 4 != 0, head_dim != 64, ... (see each entry point) */
#define GA_DIT_ERR_LAUNCH (-4)

typedef uint16_t ga_bf16;

/* epilogues of ga_gemm_bf16:  acc[m][n] = sum_k A[m][k] * W[n][k]  (fp32 accumulate) */
#define GA_GEMM_EPI_STORE_BF16 0 /* out_bf16[m][n] = acc + bias[n]                                      */
#define GA_GEMM_EPI_GELU_BF16 1  /* out_bf16[m][n] = gelu_erf(acc + bias[n])            (FusedMLP fc1) */
#define GA_GEMM_EPI_RESIDUAL 2   /* x_f32[m][n]   += gate[m / rows_per_batch][n] * (acc + bias[n])      */
#define GA_GEMM_EPI_STORE_F32 3  /* x_f32[m][n]    = acc + bias[n]                                      */

typedef struct GaGemmArgs {
    int32_t M, N, K;          /* K % 64 == 0, N % 4 == 0                                             */
    int32_t epilogue;         /* GA_GEMM_EPI_*                                                       */
    const ga_bf16 *A;         /* [M, lda] activations                                                */
    int64_t lda;              /* elements; >= K, multiple of 8                                       */
    const ga_bf16 *W;         /* [N, K] weight as torch.nn.Linear stores it (out_features, in_features) */
    const float *bias;        /* [N] or NULL                                                         */
    void *out;                /* ga_bf16[M, ldo] (EPI 0/1) or float[M, ldo] (EPI 2/3)                */
    int64_t ldo;              /* elements, multiple of 4                                             */
    const float *gate;        /* EPI 2: [M / rows_per_batch, gate_stride] or NULL (= 1)              */
    int64_t gate_stride;      /* elements between the gate rows of consecutive batch items           */
    int32_t rows_per_batch;   /* EPI 2 / V^T store: tokens per batch item                            */
    /* EPI 0 only, optional: output columns n >= vt_col0 (the V projection) are NOT written to `out` but TRANSPOSED into
     * vt[(b*heads + h)*64 + d][token], b = m / rows_per_batch, token = m % rows_per_batch, h*64 + d = n - vt_col0, row
     * length vt_ld (>= rows_per_batch, multiple of 64; the pad must stay zero) -- the layout ga_attention_bf16 reads. */
    ga_bf16 *vt;
    int32_t vt_col0;
    int64_t vt_ld;
    /* EPI 0 only, optional: per-head RMSNorm (eps 1e-5, dit/norm.py:29-43) of the 64-wide column groups ("heads") of the
     * result before the bf16 store -- columns [0, qk_cols0) with weight qk_w0[64], [qk_cols0, qk_cols1) with qk_w1[64]
     * (multiples of 64; 0 = none).  This is the q_norm / k_norm of MemEffAttention / MemoryEfficientCrossAttention
     * applied where the projection is produced instead of where it is consumed. */
    const float *qk_w0, *qk_w1;
    int32_t qk_cols0, qk_cols1;
    /* Folding an UN-modulated RMSNorm (y = x * rsqrt(mean(x^2) + eps) * weight, dit/norm.py:29-43) that sits between a
     * residual GEMM and the next projection into the two GEMMs (one launch and one pass over the stream less); the norm
     * weight is folded into the consumer's W offline (W'[n][k] = W[n][k] * weight[k]):
     *   producer, EPI 2 only, optional (N % 64 == 0): besides x += ..., write emit_x[m][n] = bf16(x_new[m][n]) and
     *     emit_ss[m][n / 64] = sum of x_new[m][.]^2 over that 64-column group (fixed order: deterministic);
     *   consumer, EPI 0 only, optional: with A = emit_x and W', every output row is scaled by
     *     rsqrt(sum_t row_ss[m][t] / row_ss_dim + row_ss_eps) (t < row_ss_tiles) before the per-head qk-norm and the store.
     *     Row m of emit_ss / row_ss starts (tiles rounded up to a multiple of 4) floats after row m - 1; the producer writes the pad
     *     entries as zeros (round 6: widths that are no multiple of 256 -- 1152: 18 sums, 20 floats per row).  row_ss_tiles <= 16
     *     when K is a multiple of 256, <= 20 otherwise; 16-byte aligned.
     * The activations are rounded to bf16 before instead of after the row scale and weight: the same relative rounding. */
    ga_bf16 *emit_x;
    float *emit_ss;
    int64_t emit_ld;          /* elements per row of emit_x, multiple of 8                           */
    const float *row_ss;
    int32_t row_ss_tiles, row_ss_dim;
    float row_ss_eps;
    /* Round 3: W may be given TILED -- [N/8][K/64][8][64] bf16, i.e. the 8 x 64 block a single 1 KiB LDS-DMA instruction moves is
     * contiguous in memory (N % 8 == 0).  Weights are packed once at load time; on weights that stream from HBM (a DiT evaluation
     * reads 600 MB of them) whole-KiB requests measured 3-6 % faster than eight 128-byte row pieces (tools/gemm_lab.hip). */
    int32_t w_tiled;
    /* Round 4: folding a MODULATED RMSNorm, y = x * rsqrt(mean(x^2) + eps) * weight * (1 + scale_b) + shift_b (the adaLN pre-norms of
     * the self-attention and the MLP: dit/dit_models_xformers.py:775-785), between a residual GEMM and the next projection:
     *     y W^T = rsqrt(.) * ((x * weight * (1 + scale_b)) W^T) + shift_b W^T
     *   producer (EPI 2 with emit_x / emit_ss), optional: emit_x[m][n] = bf16(x_new[m][n] * emit_w[n] * (1 + emit_scale[b][n])),
     *     b = m / rows_per_batch, emit_scale rows emit_scale_stride elements apart; emit_ss stays the sum of the RAW x_new^2;
     *   consumer (EPI 0 or 1): row_ss as above, applied to the accumulator BEFORE the bias, and the bias may be one row per batch
     *     item -- bias[b * bias_stride + n], bias_stride != 0 -- holding bias[n] + sum_k shift_b[k] W[n][k] (ga_dit_shift_bias).
     * k_rows (EPI 2, 0 = M): only the first k_rows rows of A take part in the product; the rows behind them get the epilogue with
     *   a zero accumulator (x += gate * bias, and the emit) -- the batch items whose cross-attention is skipped (ca_batch) still
     *   receive the output bias and the folded pre-norm of the next projection in the same launch.  A is not read past row k_rows. */
    const float *emit_w, *emit_scale;
    int64_t emit_scale_stride;
    int64_t bias_stride;
    int32_t k_rows;
    /* Round 6: optional scratch for a DETERMINISTIC split-K (EPI 0 / 1 / 2 without k_rows: reductions whose output tiles cannot fill the
     * chip -- FusedMLP's second linear up to 3072 rows, every wide projection at 768).  2 or 4 workgroups share the reduction of one 192 x 128 / 96 x 128 output tile; the partial
     * tiles meet in the scratch and are added in split order by whichever workgroup arrives last -- results do not depend on the
     * arrival order (bit-reproducible), nobody waits for anybody.  Layout: GA_GEMM_SPLITK_COUNTER_BYTES of tile counters, which must
     * be ZERO before the call and are left zero by it, then the partial tiles.  ga_gemm_splitk_workspace_bytes(M, N) bytes always
     * suffice; NULL or a smaller buffer = no split-K (the same kernels as before).  256-byte aligned.  One launch at a time per
     * scratch buffer (launches on one stream are; concurrent streams need a buffer each). */
    void *splitk_ws;
    int64_t splitk_ws_bytes;
} GaGemmArgs;

#define GA_GEMM_SPLITK_COUNTER_BYTES 16384
#define GA_GEMM_SPLITK_MAX_TILES (GA_GEMM_SPLITK_COUNTER_BYTES / 4)
size_t ga_gemm_splitk_workspace_bytes(int32_t M, int32_t N);
/* Which split-K configuration calls with a scratch take.  MEASURED SLOWER than the unsplit kernels on every shape of the denoiser
 * (profiles/r6_splitk.txt), therefore OFF by default: -1 (default; or the value of the environment variable GA_GEMM_SPLITK at first use)
 * and 0: none; 1: 192 x 128 tiles x 4 splits; 2: 96 x 128 x 2; 3: 96 x 128 x 4; 4: 192 x 128 x 2 (the only one the bf16-store epilogues
 * have); 6: chosen by shape (splits x tiles must fill 160 ... 256 workgroups); 5: the same for EPI 2 only.  A configuration still needs
 * K / 64 divisible by 4 x splits and >= 8 x splits.  Returns the previous mode.  Process-wide. */
int ga_gemm_splitk_mode(int mode);

int ga_gemm_bf16(const GaGemmArgs *args, void *stream);

/* out[b][n] = bias[n] + sum_k shift[b * shift_stride + k] * W[n][k]  (fp32; W [N, K] bf16, row-major or the tiled image; N % 8 == 0,
 * K % 64 == 0; bias may be NULL): the per-batch bias row of a consumer GEMM behind a folded modulated RMSNorm (GaGemmArgs.bias_stride).
 * ga_dit_forward computes these rows for every block's qkv and fc1 projection in one launch per evaluation. */
int ga_dit_shift_bias(const ga_bf16 *W, int32_t w_tiled, const float *bias, int32_t N, int32_t K, const float *shift,
                      int64_t shift_stride, int32_t batch, float *out, void *stream);

/* softmax(q k^T / sqrt(64)) v with per-head RMSNorm of q and k fused on load (weights qn/kn, eps 1e-5; NULL = none):
 * what MemEffAttention / MemoryEfficientCrossAttention compute between their projections.  head_dim must be 64.
 * q row of (batch b, token i, head h) starts at q + (b*Lq + i)*q_stride + h*64 (same for k, out).
 * V is read TRANSPOSED: vt[(b*heads + h)*64 + d][key], row length vt_ld >= Lk rounded up to 64, zero beyond Lk
 * (written in that layout by ga_gemm_bf16's V^T store).  q, k, vt 16-byte aligned with strides % 8 == 0 (k and vt are moved
 * by LDS-DMA unless k_norm_weight is given), out 8-byte aligned with out_stride % 4 == 0. */
typedef struct GaAttentionArgs {
    int32_t batch, heads, Lq, Lk;
    const ga_bf16 *q, *k, *vt;
    int64_t q_stride, k_stride, vt_ld;    /* elements                                               */
    const float *q_norm_weight, *k_norm_weight; /* [64] each                                        */
    ga_bf16 *out;
    int64_t out_stride;
    /* Optional (round 5): the q PROJECTION inside the attention workgroup -- q = A W^T per head, instead of a projection launch in
     * front (the denoiser's cross-attention, /root/reference/ldm/modules/attention.py:497-522: to_q has no bias).  qp_a != NULL: `q`
     * is not read; row (b, i) of A starts at qp_a + (b*Lq + i)*qp_lda, qp_k elements (qp_k % 64 == 0); qp_w = the [heads*64, qp_k]
     * weight, row-major or the tiled image of GaGemmArgs.w_tiled; qp_row_ss (optional) = the rows' partial sums of squares of a folded
     * RMSNorm as GaGemmArgs.row_ss reads them (row scale rsqrt(sum / qp_row_ss_dim + qp_row_ss_eps) applied to the product).  The
     * per-head RMSNorm q_norm_weight is applied to the product, which is rounded to bf16 as the projection GEMM would have stored it.
     * k_norm_weight must be NULL (the LDS-DMA path). */
    const ga_bf16 *qp_a, *qp_w;
    int64_t qp_lda;
    int32_t qp_k, qp_w_tiled;
    const float *qp_row_ss;
    int32_t qp_row_ss_tiles, qp_row_ss_dim;
    float qp_row_ss_eps;
} GaAttentionArgs;

int ga_attention_bf16(const GaAttentionArgs *args, void *stream);

/* The same attention for head dims OTHER than 64 (head_dim % 8 == 0, <= 128): DiT-PixArt-PCD-CLAY-XL of the reference registry has 16
 * heads of 72 (/root/reference/dit/dit_i23d.py:1526-1535, 1677).  q, k, v row-major: row (b, i) of q starts at q + (b*Lq + i)*q_stride +
 * h*head_dim (k, v: b*Lk + j); q and k ALREADY carry their per-head RMSNorm (ga_head_rmsnorm_bf16); softmax scale head_dim^-1/2.
 * 16-byte aligned operands, strides % 8 == 0; out 8-byte aligned, out_stride % 4 == 0.  With v row-major: the correctness-first kernel of
 * round 5; with vt (below): the tuned one ga_dit_forward uses. */
typedef struct GaAttentionHdArgs {
    int32_t batch, heads, Lq, Lk, head_dim;
    const ga_bf16 *q, *k, *v;
    int64_t q_stride, k_stride, v_stride;   /* elements */
    ga_bf16 *out;
    int64_t out_stride;
    /* Round 6, the tuned variant: vt != NULL -> v / v_stride are ignored and V is read TRANSPOSED, row (b*heads + h)*head_dim + d of
     * vt holds V[b][:, h, d] with the keys contiguous (vt_ld elements per row, a multiple of 64 >= Lk rounded up to 64, columns >= Lk
     * finite) -- the image GaGemmArgs.vt stores for any width.  q_norm_weight != NULL (fp32 [head_dim], 16-byte aligned; V^T variant
     * only): q arrives WITHOUT its per-head RMSNorm, which is applied here (eps 1e-5, dit/norm.py:29-43). */
    const ga_bf16 *vt;
    int64_t vt_ld;
    const float *q_norm_weight;
    const float *k_norm_weight;   /* the same for k (every workgroup normalises the key rows it stages) */
} GaAttentionHdArgs;

int ga_attention_hd_bf16(const GaAttentionHdArgs *args, void *stream);

/* In place: x[r][h][:] *= rsqrt(mean(x[r][h][:]^2) + 1e-5) * weight[:] for r < rows, h < heads (bf16 storage, fp32 arithmetic; row r starts
 * at x + r*row_stride): the per-head q / k RMSNorm of dit/norm.py:29-43 for head dims the projection GEMM's epilogue does not cover. */
int ga_head_rmsnorm_bf16(ga_bf16 *x, int64_t rows, int64_t row_stride, int32_t heads, int32_t head_dim, const float *weight, void *stream);

/* out_bf16[m][:] = rmsnorm(x[m][:]; eps 1e-5) * weight * (1 + scale[b][:]) + shift[b][:],  b = m / rows_per_batch
 * (scale / shift NULL = plain RMSNorm).  dit/norm.py:29-43 + t2i_modulate.  D % 4 == 0, D <= 2048. */
typedef struct GaRmsNormArgs {
    int32_t M, D, rows_per_batch;
    const float *x;
    const float *weight;
    const float *scale, *shift; /* [M / rows_per_batch, mod_stride]                                  */
    int64_t mod_stride;
    ga_bf16 *out;
    /* optional: rows >= row_bias_first first receive x[row] += row_bias (WRITTEN BACK to x, which is then not const) --
     * how the residual stream of batch items whose cross-attention was skipped gets the projection bias (see ca_batch) */
    const float *row_bias;      /* [D] or NULL */
    int32_t row_bias_first;
} GaRmsNormArgs;

int ga_rmsnorm_modulate(const GaRmsNormArgs *args, void *stream);

/* Small dense layer for the conditioning path (a handful of rows):
 *   y[b][n] = act_out( sum_k act_in(x[b][k]) * W[n][k] + bias[n] ) (+ add[b][n]);  act: 0 none, 1 SiLU.  B <= 16, K % 8 == 0. */
typedef struct GaSmallLinearArgs {
    int32_t B, N, K, act_in, act_out;   /* act_*: 0 none, 1 SiLU; act_in 2: x is [B] timesteps, the input row is the reference's
                                           sinusoidal timestep_embedding(t, 256) formed on the fly (K = 256)                      */
    const float *x;      /* [B, K]            */
    const ga_bf16 *W;    /* [N, K]            */
    const float *bias;   /* [N] or NULL       */
    const float *add;    /* [B, N] or NULL    */
    float *y;            /* [B, N]            */
} GaSmallLinearArgs;

int ga_small_linear(const GaSmallLinearArgs *args, void *stream);

/* ---- whole forward --------------------------------------------------------------------------------------------- */

typedef struct GaDitBlockWeights {
    /* cross-attention on the image tokens (cross_attn_dino, prenorm_ca_dino) */
    const float *prenorm_ca_w;           /* [D]                                         */
    const ga_bf16 *ca_q_w;               /* [D, D]      to_q (no bias)                  */
    const ga_bf16 *ca_q_w_prenorm;       /* [D, D] or NULL: to_q with prenorm_ca_w folded into its columns (to_q[n][k] *
                                          * prenorm_ca_w[k]); when given, blocks after the first run the pre-norm folded
                                          * into the previous fc2 and this projection (see GaGemmArgs) */
    const ga_bf16 *ca_kv_w;              /* [2D, ctx]   to_k rows then to_v rows        */
    const float *ca_q_norm_w, *ca_k_norm_w; /* [64]                                     */
    const ga_bf16 *ca_out_w;             /* [D, D]      to_out.0                        */
    const float *ca_out_b;               /* [D]                                         */
    /* self-attention (attn, norm1) */
    const float *norm1_w;                /* [D]                                         */
    const ga_bf16 *qkv_w;                /* [3D, D]                                     */
    const float *qkv_b;                  /* [3D]                                        */
    const float *q_norm_w, *k_norm_w;    /* [64]                                        */
    const ga_bf16 *proj_w;               /* [D, D]                                      */
    const float *proj_b;                 /* [D]                                         */
    /* FusedMLP (mlp, norm2) */
    const float *norm2_w;                /* [D]                                         */
    const ga_bf16 *fc1_w;                /* [4D, D]                                     */
    const float *fc1_b;                  /* [4D]                                        */
    const ga_bf16 *fc2_w;                /* [D, 4D]                                     */
    const float *fc2_b;                  /* [D]                                         */
    const float *scale_shift_table;      /* [6, D]                                      */
} GaDitBlockWeights;

typedef struct GaDitModel {
    int32_t hidden, depth, heads, in_channels, out_channels, context_dim, stage2;
    const ga_bf16 *t_mlp0_w; const float *t_mlp0_b;     /* [D,256], [D]   t_embedder.mlp.0            */
    const ga_bf16 *t_mlp2_w; const float *t_mlp2_b;     /* [D,D], [D]     t_embedder.mlp.2            */
    const float *pool_ln_w, *pool_ln_b;                 /* [ctx]          pooled_vec_embedder.0       */
    const ga_bf16 *pool_w; const float *pool_b;         /* [D,ctx], [D]   pooled_vec_embedder.1       */
    const ga_bf16 *adaln_w; const float *adaln_b;       /* [6D,D], [6D]   adaLN_modulation.1          */
    const float *xe_fc1_w, *xe_fc1_b;                   /* [D,C], [D]     x_embedder.fc1 (fp32, K tiny) */
    const ga_bf16 *xe_fc2_w; const float *xe_fc2_b;     /* [D,D], [D]     x_embedder.fc2              */
    const float *xyz_w, *xyz_b;                         /* [D,63], [D]    xyz_pos_embed.xyz_projection (stage 2) */
    const float *final_table;                           /* [2,D]          final_layer.scale_shift_table */
    const float *final_w, *final_b;                     /* [Cout,D], [Cout] final_layer.linear (fp32) */
    const GaDitBlockWeights *blocks;                    /* host array [depth]                          */
    int32_t gemm_weights_tiled;                         /* 1: every bf16 [N, K] weight that goes through ga_gemm_bf16 (block weights,
                                                           xe_fc2_w) is stored tiled, see GaGemmArgs.w_tiled; the small-linear
                                                           weights (t_mlp*, pool_w, adaln_w) stay row-major */
} GaDitModel;

/* Optional sampler step fused into the final layer (GaDitForwardArgs.step): the Euler update of the reference's fixed-grid
 * sampling loop (/root/reference/transport/integrators.py:100-119 with torchdiffeq's euler: y += dt * f(t, y)) around
 * forward_with_cfg (/root/reference/dit/dit_i23d.py:159-172):
 *     v      = cfg ? u + cfg_scale * (c - u) : model output          (c, u: rows of batch item b < B'/2 and b + B'/2)
 *     state += dt * v                                                 (both CFG halves receive the same v)
 *     traj[(*counter + 1) * traj_stride ...] = state                  (the [num_steps, *x.shape] output of sample_ode)
 * every product and sum rounded once, in this order -- bit-identical to the eager PyTorch loop.  `state` is the x of the
 * evaluation (GaDitForwardArgs.x must point at it); `out` is not written.  ga_dit_sampler_advance moves the step on. */
typedef struct GaDitSamplerStep {
    float cfg_scale;
    int32_t cfg;              /* 1: combine the two CFG halves; 0: v = model output                     */
    const float *dt;          /* device scalar: step size of this step                                  */
    float *state;             /* [B', L, Cout] fp32, updated in place                                    */
    float *traj;              /* trajectory base (slice 0 = initial state, written by the caller) or NULL */
    int64_t traj_stride;      /* floats per slice = B' * L * Cout                                        */
    const int32_t *counter;   /* device: index of the grid interval this step integrates                */
    float *velocity;          /* NULL: the Euler step above.  Otherwise ONLY the velocity of the evaluation -- the model output, or with
                                 cfg the guided combination u + s (c - u) in BOTH halves (forward_with_cfg's return value) -- is written
                                 here, [B', L, Cout]; dt / state / traj / counter are not used (the stages of ga_ode_dopri5_*)           */
} GaDitSamplerStep;

typedef struct GaDitForwardArgs {
    int32_t batch;            /* B' (CFG batch: 2 x samples), <= 16                                    */
    int32_t tokens;           /* L latent tokens per batch item                                        */
    int32_t ctx_tokens;       /* M image tokens per batch item                                         */
    const float *x;           /* [B', L, C]                                                            */
    const float *timesteps;   /* [B']                                                                  */
    const float *img_vector;  /* [B', ctx]                                                             */
    const float *fps_xyz;     /* [B', L, 3] (stage 2) or NULL                                          */
    const ga_bf16 *ca_k;      /* [depth][B'*M, D]        cached K projections of the image tokens (ga_dit_cache_context) */
    const ga_bf16 *ca_vt;     /* [depth][B'*D, Mp] cached V projections, transposed (row b*D + h*head_dim + d), Mp = M rounded up to 64,
                                 zero pad -- the same two images for every head dim since round 6                                     */
    float *out;               /* [B', L, Cout] fp32 (the reference returns x.float())                  */
    void *workspace;          /* ga_dit_workspace_bytes()                                              */
    size_t workspace_bytes;
    /* Number of LEADING batch items whose image tokens are not all zero (<= 0 or > batch: all).  For an all-zero context
     * (the unconditional half of a CFG batch: sgm's force_uc_zero_embeddings) K = V = 0 exactly (to_k / to_v have no bias,
     * RMSNorm(0) = 0), the softmax is uniform over zeros and the block's cross-attention reduces to `x += to_out.bias`:
     * those items skip the q projection, the 1369-key attention and the output projection -- bit-identical result. */
    int32_t ca_batch;
    const GaDitSamplerStep *step;   /* host pointer, NULL = plain function evaluation into `out`                */
    /* [B', D] fp32 or NULL: pooled_vec_embedder(img_vector) = Linear(LayerNorm(img_vector)) (dit_i23d.py:540-545) as ga_dit_pooled_vector
     * computed it for THIS img_vector.  It does not depend on the time: a sampling loop computes it once per conditioning instead of
     * in each of its ~250 evaluations (two launches of the evaluation's serial head).  NULL: computed inside, into the workspace. */
    const float *pooled_vec;
} GaDitForwardArgs;

size_t ga_dit_workspace_bytes(const GaDitModel *model, int32_t batch, int32_t tokens, int32_t ctx_tokens);

/* K/V of every block's cross-attention depend only on the (step-invariant) image tokens: project them once per
 * sample.  img_crossattn: bf16 [B'*M, ctx]; outputs as described in GaDitForwardArgs (ca_vt must be zero-filled by the
 * caller beforehand: only the first M keys of every row are written). */
int ga_dit_cache_context(const GaDitModel *model, int32_t batch, int32_t ctx_tokens, const ga_bf16 *img_crossattn,
                         ga_bf16 *ca_k, ga_bf16 *ca_vt, void *stream);

int ga_dit_forward(const GaDitModel *model, const GaDitForwardArgs *args, void *stream);

/* pooled_vec_embedder of the reference model (LayerNorm with affine over context_dim, then Linear context_dim -> D, bf16-rounded
 * inputs as under autocast): img_vector [B', ctx] fp32 -> out [B', D] fp32; scratch [B', ctx] fp32.  See GaDitForwardArgs.pooled_vec. */
int ga_dit_pooled_vector(const GaDitModel *model, int32_t batch, const float *img_vector, float *scratch, float *out, void *stream);

/* host: one tiny kernel closing a fused sampler step: ++*counter; timesteps[0..batch) = t_grid[*counter];
 * *dt = dt_grid[*counter] (both grids fp32 device arrays of num_steps - 1 entries; reads past the end are clamped). */
int ga_dit_sampler_advance(int32_t *counter, const float *t_grid, const float *dt_grid, int32_t grid_len, float *timesteps,
                           int32_t batch, float *dt, void *stream);

/* ---- device-resident Dormand-Prince 5(4): the reference's default sampler (Sampler.sample_ode(sampling_method="dopri5"),
 * /root/reference/transport/transport.py:384-431 -> torchdiffeq.odeint, transport/integrators.py:111-118).  One attempted step is
 *     for i in 0..5:  ga_ode_dopri5_stage(o, i);  ga_dit_forward(x = o->ystage, timesteps = o->timesteps, step.velocity = o->k[i + 1])
 *     ga_ode_dopri5_finish(o)
 * -- a fixed launch sequence (capture it into a hipGraph and replay it) that keeps time, step size, the accept / reject decision and
 * the counters in `ctl`; the host only reads ctl[GA_ODE_DONE] after a replay.  Before the first step the caller puts the initial
 * state into y, f(t0, y0) into k[0], and t0 / the first step size / atol / rtol / JNEXT = 1 (out[0] = y0 is the caller's) into ctl. */
#define GA_ODE_T 0          /* time of the state y (fp64)                                  */
#define GA_ODE_DT 1         /* size of the NEXT attempted step                             */
#define GA_ODE_SUMSQ 2      /* sum of squares the last decision was taken on (diagnostic)  */
#define GA_ODE_ATOL 3
#define GA_ODE_RTOL 4
#define GA_ODE_DONE 5       /* 1: the last requested time has been produced (or ERROR set) */
#define GA_ODE_STEPS 6      /* attempted steps                                             */
#define GA_ODE_REJECTED 7
#define GA_ODE_ACCEPT 8     /* the step just attempted was accepted                        */
#define GA_ODE_TA 9         /* its interval                                                */
#define GA_ODE_TB 10
#define GA_ODE_DT_USED 11
#define GA_ODE_JNEXT 12     /* next requested time not produced yet                        */
#define GA_ODE_JBEG 13      /* requested times inside the accepted step: first, count      */
#define GA_ODE_JCOUNT 14
#define GA_ODE_ERROR 15     /* 1: non-finite error ratio (NaN / inf model output), 2: step size underflow */
#define GA_ODE_RATIO 16     /* error ratio of the step just attempted (diagnostic)         */
#define GA_ODE_CTL_WORDS 24          /* the scalar head: what the host reads back after a replay                                      */
#define GA_ODE_MAX_PARTIALS 2048     /* behind the head: one partial sum of the error norm per workgroup of the error launch, added  */
#define GA_ODE_CTL_ALLOC (GA_ODE_CTL_WORDS + GA_ODE_MAX_PARTIALS)   /* by the controller in a fixed order (bit-reproducible steps)      */

typedef struct GaOdeDopri5 {
    int64_t n;              /* floats of the state: B' * L * C                             */
    int32_t batch;          /* B': entries of `timesteps`                                  */
    int32_t grid_len;       /* requested times                                             */
    float *y;               /* [n] state at ctl[T]                                         */
    float *k[7];            /* [n] each: k[0] = f(t, y) (FSAL), k[1..6] the stage derivatives */
    float *ystage;          /* [n] input of the next function evaluation; after stage 5: the 5th-order solution */
    float *timesteps;       /* [B'] time of the next function evaluation (fp32, as the reference passes it)     */
    double *ctl;            /* [GA_ODE_CTL_ALLOC] device scalars (head of GA_ODE_CTL_WORDS) + the error-norm partials */
    const double *t_grid;   /* [grid_len] requested times, increasing                     */
    float *out;             /* [grid_len, n] dense output (slice 0 is the caller's)        */
    int64_t ctl_words;      /* doubles allocated behind `ctl`: at least GA_ODE_CTL_WORDS + min(ceil(n / 256), GA_ODE_MAX_PARTIALS);
                             * GA_ODE_CTL_ALLOC always suffices.  ga_ode_dopri5_finish returns GA_DIT_ERR_BAD_SHAPE below that
                             * (the error launch writes one partial per workgroup behind the head).                          */
} GaOdeDopri5;

int ga_ode_dopri5_stage(const GaOdeDopri5 *ode, int32_t stage, void *stream);
int ga_ode_dopri5_finish(const GaOdeDopri5 *ode, void *stream);

const char *ga_dit_version(void);

#ifdef __cplusplus
}
#endif
#endif /* GA_DIT_H */
