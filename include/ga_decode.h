/*
 * ga_decode.h -- C-ABI of the MI355X-native surfel decode: the operators that, together with
 * ga_gemm_bf16 / ga_attention_bf16 of ga_dit.h, compute what the reference computes in
 *     pcd_structured_latent_space_vae_decoder_cascaded.vit_decode_backbone / vit_decode_postprocess
 *                                                  /root/reference/vit/vit_triplane.py:1415-1427, 1467-1501, 1645-1676
 *     DiT2.forward / DiTBlock2.forward             /root/reference/dit/dit_decoder.py:99-176, 19-35
 *     GS_Adaptive_Read_Write_CA_adaptive_2dgs.forward   vit_triplane.py:995-1064
 *     SRT Transformer / PreNorm                    /root/reference/nsr/srt/layers.py:82-92, 146-190
 * latent tokens + anchor points -> 768 base surfels -> x8 -> x4 -> x3 = 73 728 surfels (13 floats each: xyz, opacity,
 * scale(2), quaternion wxyz, rgb -- the tensor GaussianRenderer2DGS.render consumes).
 *
 * The sequence of calls is host code (gaussiananything_amd/decode.py, the mirror of the reference classes); the
 * entry points below are the pieces that have no counterpart in ga_dit.h.  Conventions as in ga_dit.h: device
 * pointers, bf16 as raw uint16, row-major, everything enqueued on `stream`, no allocation, 0 or GA_DIT_ERR_*.
 */
#ifndef GA_DECODE_H
#define GA_DECODE_H

#include "ga_dit.h"

#ifdef __cplusplus
extern "C" {
#endif

/* post_quant_conv (timm Mlp, tanh-GELU; vit_triplane.py:1322-1326) followed by the SiLU every consumer applies
 * (adaLN_modulation = SiLU -> Linear, dit_models_xformers.py:288-289):
 *   out_bf16[m][:] = silu( W2 gelu_tanh(W1 x[m] + b1) + b2 ),   W1 [Ch, Cin], W2 [D, Ch],  Cin, Ch <= 16 */
typedef struct GaTinyMlpArgs {
    int32_t M, Cin, Ch, D;
    const float *x;               /* [M, Cin] */
    const float *w1, *b1, *w2, *b2;
    ga_bf16 *out;                 /* [M, D]   */
} GaTinyMlpArgs;
int ga_tiny_mlp_silu(const GaTinyMlpArgs *args, void *stream);

/* LayerNorm over the last dimension (biased variance, fp32), optional affine, optional per-row modulation:
 *   out_bf16[m][:] = ( (x[m] - mean) * rstd * weight + bias ) * (1 + scale[m]) + shift[m]
 * DiTBlock2: no affine, eps 1e-6, scale/shift per TOKEN (rows of the adaLN output, stride mod_stride);
 * PreNorm (nsr/srt/layers.py:82-92): affine, eps 1e-5, no modulation.   D % 4 == 0, D <= 2048. */
typedef struct GaLayerNormArgs {
    int32_t M, D;
    float eps;
    const float *x;               /* [M, D]                      */
    const float *weight, *bias;   /* [D] or NULL (both)          */
    const float *scale, *shift;   /* [M, mod_stride] or NULL     */
    int64_t mod_stride;
    ga_bf16 *out;                 /* [M, D]                      */
} GaLayerNormArgs;
int ga_layernorm_modulate(const GaLayerNormArgs *args, void *stream);

/* Token groups of the upsampler: group p of S = 1 + f rows = [ feature of anchor p | the f learned query embeddings ]
 * (vit_triplane.py:1009-1016).  The anchor feature is row p of `src` (src_f == 0: the decoder output) or row
 * (p / src_f) * (1 + src_f) + 1 + p % src_f (src_f > 0: `src` is the previous level's token stream, whose groups of
 * 1 + src_f rows each start with a row that is not an anchor of this level). */
typedef struct GaAssembleArgs {
    int32_t P, f, D, src_f;
    const float *src;             /* fp32 rows of width D        */
    const float *latent_embedding;/* [f, D]                      */
    float *out;                   /* [P * (1 + f), D]            */
} GaAssembleArgs;
int ga_assemble_tokens(const GaAssembleArgs *args, void *stream);

/* Self-attention inside groups of S <= 16 consecutive rows (batch = number of groups, 768 ... 24 576 here), head_dim
 * 64, softmax scale 1/8; q and k arrive RMS-normalised from the projection GEMM (ga_gemm_bf16 qk_w0/qk_w1).
 * The reference chunks this shape at 32 768 groups and forces the cutlass op (vision_transformer.py:254-279). */
typedef struct GaTinyAttentionArgs {
    int32_t groups, S, heads;
    const ga_bf16 *qkv;           /* [groups * S, 3 * heads * 64]: q | k | v   */
    ga_bf16 *out;                 /* [groups * S, heads * 64]                  */
} GaTinyAttentionArgs;
int ga_tiny_attention(const GaTinyAttentionArgs *args, void *stream);

/* Surfel head: 13-channel prediction + activations (vit_triplane.py:287-341, 1385-1412, 1430-1440, 1036-1057).
 *   mode 0 (base level):   pre = Linear13(SiLU(x[r]));                pos = tanh(pre[0:3]) * 0.225 * skip + anchor[r]
 *   mode 1 (upsampler):    pre = Linear13(LayerNorm_affine(x[r'])) + base_pre[p];   pos = tanh(res[0:3]) * 0.225 + base_pos[p]
 *                          (r' = p * (1 + f) + 1 + j skips the group's leading row; res = the part before the base is added)
 *   gaussians[r] = [pos, sigmoid(pre[3]), softplus(pre[4:6]) * (0.0045 / ln 2), normalize(pre[6:10]), 0.5 tanh(pre[10:13]) + 0.5]
 * `pre_out` receives the pre-activation 13-vector (the next level adds its residual to it). */
typedef struct GaSurfelHeadArgs {
    int32_t rows, D, mode, f;     /* rows = anchors (mode 0) or P * f (mode 1) */
    const float *x;               /* mode 0: [rows, D]; mode 1: [P * (1 + f), D] */
    const float *ln_weight, *ln_bias; /* mode 1 */
    const float *w, *b;           /* [13, D], [13]                             */
    const float *anchor;          /* mode 0: [rows, 3] anchor points; mode 1: [P, 13] gaussians of the previous level */
    const float *base_pre;        /* mode 1: [P, 13]                           */
    float skip_weight;            /* mode 0 */
    float *gaussians;             /* [rows, 13]                                */
    float *pre_out;               /* [rows, 13]                                */
} GaSurfelHeadArgs;
int ga_surfel_head(const GaSurfelHeadArgs *args, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GA_DECODE_H */
