/*
 * m3ae_hip.h -- C ABI of libm3ae_hip.so: the MI355X (gfx950) kernels behind the M3AE Med-VQA hot path.
 *
 * The reference (better62/MM-VQA-Healthcare) is 100 % Python: it has no FFI / plugin interface for this path.
 * Its boundary is the nn.Module tree of M3AETransformerSS (m3ae/modules/m3ae_module.py:16) whose blocks call
 * PyTorch ATen ops.  Each entry point below replaces the ATen call chain of one reference code site (cited per
 * function); the Python modules in mm-vqa-healthcare_amd/m3ae_amd/modules/ keep the reference's module / parameter
 * names and call these entry points through ctypes (m3ae_amd/_lib.py).  See INTEGRATION.md for the binding stub.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch types.  All pointers are DEVICE pointers unless noted.
 *   - Every function is stream-ordered on `stream` (a hipStream_t passed as void*), never synchronises the host,
 *     allocates nothing and keeps no mutable global state; the caller owns every buffer incl. workspaces.
 *   - Return value: 0 = ok; M3AE_ERR_* (negative) = rejected before launch; positive = hipError_t of the launch.
 *   - dtype: activations are M3AE_BF16 ("perf mode") or M3AE_F32 ("parity mode"); parameters that are vectors
 *     (biases, LayerNorm scale/shift), embedding tables, masks, statistics and gradients of parameters are fp32.
 */
#ifndef M3AE_HIP_H
#define M3AE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: m3ae_gemm_desc grew `preact_grad` and `launch_flags` (round 2), m3ae_xattn_desc grew `launch_flags` and dir 1 takes
 *    probs == NULL (forward-only calls); m3ae_desc_sizes() lets a binding check its struct layouts at load time. */
/* 3: dropout salt (round 4).  m3ae_gemm_desc / m3ae_attn_desc / m3ae_xattn_desc grew `dropout_salt`, m3ae_layernorm_bwd_drop and
 *    m3ae_dropout take it as an argument: an optional DEVICE pointer to a 32-bit value the kernels fold into the mask key, so a
 *    step captured in a hipGraph (kernel arguments frozen, seeds included) draws new masks at every replay once the caller bumps
 *    that value; m3ae_adamw takes `hyper_dev` (device {lr, step size} of the step) for the same reason. */
#define M3AE_ABI_VERSION 3

enum { M3AE_F32 = 0, M3AE_BF16 = 1 };
enum { M3AE_ACT_NONE = 0, M3AE_ACT_GELU = 1, M3AE_ACT_QUICKGELU = 2, M3AE_ACT_TANH = 3, M3AE_ACT_RELU = 4,
       M3AE_ACT_MULAUX = 5 /* dact only: dact_aux already holds act'(pre) (written by a forward with preact_grad) */ };
enum { M3AE_ERR_ARG = -1, M3AE_ERR_UNSUPPORTED = -2, M3AE_ERR_ALIGN = -3, M3AE_ERR_WORKSPACE = -4 };

int m3ae_abi_version(void);
/* sizeof(m3ae_gemm_desc), sizeof(m3ae_attn_desc), sizeof(m3ae_xattn_desc) as this library was compiled: a binding compares
 * them with its own struct definitions before the first call (a descriptor that is too short is read past its end). */
void m3ae_desc_sizes(int64_t out3[3]);
/* name of the kernel family the last m3ae_gemm call on this thread dispatched to ("mfma_nt", "mfma_tn", "generic") */
const char* m3ae_last_gemm_path(void);

/* ------------------------------------------------------------------------------------------------------------
 * GEMM with fused epilogue.  Replaces nn.Linear / torch.matmul / F.conv2d(k=s=patch) call sites:
 *   clip_model.py:44,47-49,58,94 (packed in-proj, out-proj, c_fc, c_proj, patch-embed conv as GEMM);
 *   bert_model.py:224-226,263,276-277,356,361,419,426,434,439 (Q/K/V, attention output, intermediate, output);
 *   m3ae_module.py:70-73,120-125,235,252 (modality projections, vqa_head); prediction_heads.py:12,17 (Pooler);
 *   and their autograd backward (dgrad / wgrad).
 *
 *   C[b1][b2][m][n] = epi( alpha * sum_k A[b1][b2][m][k] * B[b1][b2][k][n] )
 *   epi(x): x += bias[n]; if (preact) preact[m][n] = preact_grad ? act'(x) : x; x = act(x);
 *           if (residual) x += residual[m][n];
 *           if (dact_aux) x *= act'(dact_aux[m][n]) (derivative of `dact` at the saved pre-activation);
 *           if (accumulate) C += x else C = x.
 * Strides are in ELEMENTS.  preact / residual / dact_aux share C's strides and dtype.
 * a_rowsum (single-batch only): additionally accumulates the row sums of op(A) over the reduction -- for wgrad
 * (A = dY^T) that is the bias gradient, so no separate column-sum pass over dY is needed.
 * Dispatch: bf16 A,B with K-contiguous operands (a_sk == b_sk == 1) -> MFMA "NT" kernel;
 *           bf16 A,B with reduction-strided operands (a_sm == b_sn == 1), fp32 C, accumulate -> MFMA "TN" (wgrad)
 *           kernel with split-K fp32 atomics; anything else (fp32 operands, odd shapes, batched) -> generic kernel.
 */
typedef struct {
    int64_t M, N, K;
    int64_t batch1, batch2;
    const void* A; int64_t a_sm, a_sk, a_sb1, a_sb2;
    const void* B; int64_t b_sk, b_sn, b_sb1, b_sb2;
    void* C;       int64_t c_sm, c_sn, c_sb1, c_sb2;
    int32_t dtype_a, dtype_b, dtype_c;
    float alpha;
    int32_t accumulate;
    const float* bias;
    int32_t act;
    void* preact;
    const void* residual;
    const void* dact_aux;
    int32_t dact;
    int32_t force_generic; /* tests: bypass the MFMA kernels */
    float* a_rowsum;       /* optional fp32 [M]: a_rowsum[m] += sum_k A[m][k] (bias gradient fused into wgrad) */
    float dropout_p;       /* > 0: x = dropout(x) after bias/activation, BEFORE the residual add (nn.Dropout of */
    uint64_t dropout_seed; /*      bert_model.py:362,440); keep-mask = m3ae counter hash of (seed, m * N + n)     */
    int32_t preact_grad;   /* forward: store act'(pre-activation) in `preact` instead of the pre-activation: the erf / exp
                            * terms are already in registers there, and the backward GEMM (dact = M3AE_ACT_MULAUX) then
                            * multiplies by the saved derivative instead of re-evaluating it (its epilogue was VALU-bound) */
    int32_t launch_flags;  /* M3AE_GEMM_NO_PERSISTENT: never take the persistent (one workgroup per CU, static tile lists)
                            * form of the NT kernel -- for callers that run collectives next to the GEMMs (RCCL kernels
                            * hold CUs; a persistent workgroup that finds none starts after another has walked its list) */
    const void* dropout_salt; /* NULL, or device uint32: folded into the dropout mask key at kernel entry (ABI 3, above) */
} m3ae_gemm_desc;
enum { M3AE_GEMM_NO_PERSISTENT = 1 };
/* Diagnostic selectors in launch_flags (0 in the product path = kernel chosen by shape): tests pin the kernel variants
 * per call to compare them bit for bit, tools time them against each other.  The library keeps no tuning state.
 *   NT variant v: 0 = 128x128 tile, 4 = 256x256 2-stage, 7 = 256x256 ping-pong, 8 = its persistent form;
 *   TN (wgrad) variant v: 0 / 2 = 128x128 tile with 64- / 32-row steps, 5 = 256x256 ping-pong;
 *   column-tile group width g (1..12) of the ping-pong kernels' tile order;
 *   bits 20-21: cache policy of the NT epilogue's streams (1 plain, 2 output stores nt, 3 + residual / aux loads nt; 0 = by size).
 *   NT variants 9 / 10 (round 4): the second-generation ping-pong kernel, one workgroup per tile / persistent. */
#define M3AE_GEMM_NT_VARIANT(v) ((((v) + 1) & 0xf) << 8)
#define M3AE_GEMM_TN_VARIANT(v) ((((v) + 1) & 0xf) << 12)
#define M3AE_GEMM_COL_GROUP(g) (((g) & 0xf) << 16)
int m3ae_gemm(const m3ae_gemm_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Scaled-dot-product attention, forward and backward (flash-style in bf16, materialised in fp32).
 * Replaces bert_model.py:301-340 (QK^T / sqrt(dh) + additive mask, softmax, PV) for self- and cross-attention in
 * BertCrossLayer / RobertaLayer, and torch-1.9 F.multi_head_attention_forward's bmm-softmax-bmm in
 * clip_model.py:58; with pos_bias also HF T5Attention (called from m3ae_t5_mm_encoder_input.py:202,244).
 *   q [B, Lq, H*Dh], k / v [B, Lk, H*Dh] with explicit batch and token strides (head h at element offset h*Dh, Dh
 *   contiguous), so packed QKV buffers are addressed in place.  o like q.
 *   key_mask: additive fp32 [B, Lk] (the reference's (1-mask)*-10000.0 extended mask, m3ae_module.py:232) or NULL.
 *   pos_bias: additive fp32 [H, Lq, Lk] or NULL.  lse: fp32 [B, H, lse_stride] (lse_stride >= Lq, multiple of 32).
 *   Dh must be 64 for the bf16 kernels.  workspace: see m3ae_attn_workspace_bytes (fp32 path only).
 */
typedef struct {
    int64_t B, H, Lq, Lk, Dh;
    const void* q; int64_t q_sb, q_sl;
    const void* k; int64_t k_sb, k_sl;
    const void* v; int64_t v_sb, v_sl;
    void* o;       int64_t o_sb, o_sl;
    const float* key_mask;
    const float* pos_bias;
    float scale;
    int32_t causal;
    float* lse; int64_t lse_stride;
    int32_t dtype;
    void* workspace; int64_t workspace_bytes;
    /* backward only */
    const void* d_o;               /* layout of o */
    void* dq; void* dk; void* dv;  /* layouts of q, k, v */
    float* delta;                  /* fp32 [B, H, lse_stride] scratch: rowsum(dO * O) */
    float* d_pos_bias;             /* fp32 [H, Lq, Lk], accumulated; or NULL */
    /* attention-probability dropout (bert_model.py:334), bf16 kernels only: P is dropped (and rescaled by 1/(1-p))
     * after the softmax normaliser is taken; mask = counter hash of (seed, ((b*H+h)*Lq+q)*Lk+k), regenerated in bwd */
    float dropout_p;
    uint64_t dropout_seed;
    const void* dropout_salt; /* as m3ae_gemm_desc.dropout_salt */
    int32_t launch_flags;     /* M3AE_ATTN_LEGACY_KERNELS: the round-3 bf16 kernels (tests compare the two generations); ABI 3 */
} m3ae_attn_desc;
enum { M3AE_ATTN_LEGACY_KERNELS = 1 };
int64_t m3ae_attn_workspace_bytes(const m3ae_attn_desc* d, int backward);
int m3ae_attn_fwd(const m3ae_attn_desc* d, void* stream);
int m3ae_attn_bwd(const m3ae_attn_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Fused cross-attention sub-block of BertCrossLayer (north_star's kernel), bf16: BertAttention as `crossattention`
 * (bert_model.py:480-488) = BertSelfAttention.forward cross branch (:253-350, :275-278) + BertSelfOutput (:353-364):
 *     out = LayerNorm(x + dropout(dense(softmax(QK^T / sqrt(dh) + mask) V)))     Q from x, K / V from y.
 * The projection of the LONG side (the 577 image tokens) is absorbed into the short side (32 text tokens); see
 * csrc/xattn.hip.  dir 0: text queries over image keys (Lq = 32); dir 1: image queries over text keys (Lk = 32).
 * Results equal the unfused composition (m3ae_gemm + m3ae_attn_fwd + m3ae_gemm + m3ae_layernorm_fwd) up to bf16
 * rounding of the intermediates; the dropout masks are THE SAME masks (same seeds, same index convention), so the
 * two can be compared under dropout.  All buffers are caller-owned; the intermediates are what the backward reads.
 *   weights (bf16): wq [D, D], wq_t = wq^T, wkv [2D, D] (key rows, then value rows), wkv_t = wkv^T [D, 2D], wo [D, D],
 *                   wo_t = wo^T; biases and LayerNorm parameters fp32.
 *   key_mask: additive fp32 [B, Lk] or NULL.
 *   proj:  dir 0: q [B*Lq, D];          dir 1: k | v [B*Lk, 2D]
 *   prime: dir 0: Q' [B, Lq*H, D];      dir 1: K' then V', each [B, H*Lk, D]
 *   colbias (dir 1): fp32 [B, H*Lk].    probs / probs_drop: dir 0 [B, Lq*H, 640]; dir 1 [B, Lq, H*Lk]
 *   (probs_drop, rowsum [B, Lq*H] fp32 (dir 0): only with dropout_p > 0).   zctx (dir 0): [B, Lq*H, D]; ctx (dir 0): [B*Lq, D]
 *   s: pre-LayerNorm sum [B*Lq, D]; out [B*Lq, D]; mean / rstd fp32 [B*Lq].
 */
typedef struct {
    int32_t dir;
    int64_t B, Lq, Lk, D, H;
    const void* x;
    const void* y;
    const float* key_mask;
    const void *wq, *wq_t, *wkv, *wkv_t, *wo, *wo_t;
    const float *bq, *bkv, *bo, *ln_g, *ln_b;
    float ln_eps;
    float dropout_p;
    uint64_t seed_attn, seed_hidden;
    void* proj;
    void* prime;
    float* colbias;
    void* probs;
    void* probs_drop;
    float* rowsum;
    void* zctx;
    void* ctx;
    void* s;
    void* out;
    float *mean, *rstd;
    /* ---- backward only (m3ae_xattn_bwd): the forward's fields above hold what the forward wrote ---- */
    const void* d_out;            /* [B*Lq, D] gradient of `out` */
    void* dx;                     /* [B*Lq, D] gradient of x (residual branch included) */
    void* dy;                     /* [B*Lk, D] gradient of y */
    /* parameter gradients, fp32, ACCUMULATED: weights [D, D] / [2D, D] like the parameters, biases [D] / [2D] */
    float *g_wq, *g_wkv, *g_wo, *g_bq, *g_bkv, *g_bo, *g_ln_g, *g_ln_b;
    /* caller-owned scratch */
    void* ws_ds;                  /* [B*Lq, D]                       gradient of s */
    void* ws_dsd;                 /* [B*Lq, D]   (dropout_p > 0)     gradient of the dense output before the hidden dropout */
    void* ws_dscores;             /* like `probs`                    gradient of the scores */
    void* ws_dprime;              /* like `prime`                    gradient of Q' (dir 0) / K', V' (dir 1) */
    void* ws_dproj;               /* like `proj`                     gradient of q (dir 0) / k | v (dir 1) */
    void* ws_dz;                  /* dir 0: [B, Lq*H, D]             gradient of zctx */
    void* ws_dctx;                /* dir 0: [B*Lq, D]                gradient of ctx */
    float* ws_vec;                /* fp32 [3 * B * H * T], T = min(Lq, Lk) text tokens */
    float* ws_ln;                 /* fp32 [2 * m3ae_layernorm_bwd_blocks(B*Lq) * D] */
    int32_t launch_flags;         /* M3AE_XATTN_*: per-call launch policy (the library keeps no state) */
    const void* dropout_salt;     /* as m3ae_gemm_desc.dropout_salt: both dropout sites of the sub-block */
} m3ae_xattn_desc;
enum { M3AE_XATTN_NO_PERSISTENT = 1,  /* the internal m3ae_gemm calls never take the persistent NT kernel (callers that run
                                       * collectives next to the step: see m3ae_gemm_desc.launch_flags) */
       M3AE_XATTN_LEGACY_CHAIN = 2    /* dir 1 forward: the round-2 chain (score GEMM + softmax epilogue, P through HBM, P V'
                                       * GEMM) instead of the one-launch kernel of csrc/xflash.hip -- A/B measurements only */ };
/* dir 1 (image queries): Lk = 32 or 64 text keys, any Lq; probs / probs_drop may be NULL in a forward-only call (the scores and
 * probabilities then never leave the chip).  dir 0 (text queries): Lq = 32 or 64, Lk <= 640. */
int m3ae_xattn_supported(const m3ae_xattn_desc* d);   /* 1 if the fused forward covers these shapes */
int m3ae_xattn_bwd_supported(const m3ae_xattn_desc* d);   /* 1 if m3ae_xattn_bwd does too */
int64_t m3ae_xattn_probs_ld(const m3ae_xattn_desc* d);
int m3ae_xattn_fwd(const m3ae_xattn_desc* d, void* stream);
/* Backward of m3ae_xattn_fwd in the same absorbed form: two more per-sample products per direction (the gradient of the
 * scores with the softmax backward in its epilogue, the gradients of the absorbed operands), the other stream's gradient,
 * and the per-head weight gradients as split-K fp32 atomics.  d(b_k) is exactly zero (b_k drops out of the softmax) and is
 * not touched.  Dropout masks are regenerated from the forward's seeds. */
int m3ae_xattn_bwd(const m3ae_xattn_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * LayerNorm (biased variance, eps inside the sqrt, fp32 statistics), optional fused activation on the output.
 * Replaces nn.LayerNorm at bert_model.py:357,363,435,441 (eps 1e-12 / 1e-5), clip_model.py:27-33 (fp32 upcast),
 * m3ae_module.py:122 (+ nn.GELU :123 via `act`).  mean / rstd: fp32 [M] saved for backward.
 * bwd: dx = LN'(dy) (+ dx_add: the residual branch's gradient of a pre-LN block, fused); dgamma / dbeta are
 * ACCUMULATED (+=) in fp32 (dgamma may be NULL for frozen parameters).  workspace: fp32 [2 * nblk * D] with
 * nblk = m3ae_layernorm_bwd_blocks(M).  rms != 0 selects T5 RMSNorm (no mean subtraction, no beta).
 */
int m3ae_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                       int64_t M, int64_t D, float eps, int dtype, int act, int rms, void* stream);
int64_t m3ae_layernorm_bwd_blocks(int64_t M);
int m3ae_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* beta, const float* mean,
                       const float* rstd, void* dx, const void* dx_add, float* dgamma, float* dbeta, float* workspace,
                       int64_t M, int64_t D, int dtype, int act, int rms, void* stream);
/* same, plus a second output dx_drop = dropout_mask(seed, p) * LN'(dy) / (1 - p): the gradient entering the dense layer
 * whose output was dropped before the residual add (post-LN BERT blocks), produced without an extra pass over dx. */
int m3ae_layernorm_bwd_drop(const void* dy, const void* x, const float* gamma, const float* beta, const float* mean,
                            const float* rstd, void* dx, void* dx_drop, float dropout_p, uint64_t dropout_seed,
                            const void* dropout_salt, float* dgamma, float* dbeta, float* workspace, int64_t M, int64_t D,
                            int dtype, void* stream);
/* out = dropout(x) on a dense [rows][cols] array with the library's counter-hash mask (forward and backward are
 * the same map).  Every dropout site of the library -- this call, the GEMM epilogue (rows = M, cols = N), the
 * LayerNorm backward second output, and attention probabilities (rows = (b*H + h)*Lq + q, cols = Lk) -- uses the mask
 * index row * ld + col with ld = cols rounded up to a multiple of 4, so a mask exported here (keep_mask, uint8
 * [rows][cols], optional) is the mask those kernels apply for the same (p, seed). */
int m3ae_dropout(const void* x, void* out, uint8_t* keep_mask, int64_t rows, int64_t cols, float p, uint64_t seed,
                 const void* salt, int dtype, void* stream);

/* out[n] (+)= sum_m x[m][n]  (bias gradients; x has row stride ldx elements). */
int m3ae_colsum(const void* x, float* out, int64_t M, int64_t N, int64_t ldx, int dtype, int accumulate,
                void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * RoBERTa embeddings (HF RobertaEmbeddings called at m3ae_module.py:230): position ids =
 * cumsum(ids != pad) * (ids != pad) + pad; out = word[ids] + type[0] + pos[position id]  (LayerNorm is a separate
 * call).  Tables are the fp32 master parameters.  bwd scatter-adds d_out into the fp32 table gradients.
 */
int m3ae_roberta_embed_fwd(const int64_t* ids, const float* word, const float* pos, const float* type, void* out,
                           int64_t B, int64_t S, int64_t D, int64_t pad_id, int dtype, void* stream);
int m3ae_roberta_embed_bwd(const int64_t* ids, const void* d_out, float* d_word, float* d_pos, float* d_type,
                           int64_t B, int64_t S, int64_t D, int64_t pad_id, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * ViT token assembly (clip_model.py:94-99): im2col of non-overlapping patches (column order c, ph, pw = the
 * row-major flattening of conv1.weight[width, 3, P, P]) and [cls | patches] + positional_embedding.
 */
int m3ae_patchify(const float* img, void* out, int64_t B, int64_t R, int64_t P, int dtype, void* stream);
/* Input pipeline tail (transforms/transform.py:60-67 ToTensor + Normalize, after the host-side PIL resize / centre crop):
 * uint8 [B, H, W, 3] (RGB, as uploaded from pinned host memory: a quarter of the fp32 PCIe bytes) ->
 * fp32 [B, 3, H, W] with out = (u / 255 - mean[c]) / std[c], the same two IEEE divisions torch performs. */
int m3ae_image_normalize_u8(const uint8_t* in, float* out, int64_t B, int64_t H, int64_t W, const float* mean3,
                            const float* std3, void* stream);
int m3ae_vit_tokens_fwd(const void* patch, const float* cls, const float* pos, void* out, int64_t B, int64_t G,
                        int64_t D, int dtype, void* stream);
int m3ae_vit_tokens_bwd(const void* d_out, void* d_patch, float* d_cls, float* d_pos, int64_t B, int64_t G,
                        int64_t D, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Losses.  bce: objectives.py:201  loss = mean(BCEWithLogits(x, z)) * C.  Writes loss[0] (fp32, overwritten) and
 * d_logits = d(loss)/dx * grad_scale.  xent: F.cross_entropy(ignore_index=-100, mean) (objectives.py:19-23,101).
 */
int m3ae_bce_logits(const void* logits, const float* targets, float* loss, void* d_logits, int64_t B, int64_t C,
                    float grad_scale, int dtype, void* stream);
int m3ae_xent(const void* logits, const int64_t* labels, float* loss, void* d_logits, float* workspace,
              int64_t rows, int64_t C, int64_t ld, float grad_scale, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * AdamW as transformers==4.6.0 implements it (m3ae_utils.py:206): bias-corrected Adam update, then decoupled
 * decay p -= lr * wd * p.  One call per contiguous fp32 segment (a parameter group of the flat buffer).
 * g is multiplied by grad_scale first (1 / world_size for DDP averaging).  If shadow != NULL also writes the bf16
 * copy used by the forward pass.
 */
int m3ae_adamw(float* p, const float* g, float* m, float* v, void* shadow_bf16, int64_t n, float lr, float beta1,
               float beta2, float eps, float wd, int64_t step, float grad_scale, const float* hyper_dev, void* stream);
/* hyper_dev (ABI 3): NULL, or a device pointer to {lr, lr * sqrt(1 - beta2^step) / (1 - beta1^step)} of THIS step: the kernel then
 * takes both from there instead of from `lr` / `step` (a hipGraph-captured step replays with frozen arguments; the caller
 * uploads the two floats per group before every replay). */

/* bf16 [R, C] <- fp32 [R, C] cast; and the transposed bf16 copy out_t [C, R] (dgrad operand). either may be NULL */
int m3ae_cast_transpose(const float* in, void* out, void* out_t, int64_t R, int64_t C, void* stream);
/* every weight unit in one launch: jobs_dev = device array of {const float* in; bf16* out_t; int64 R, C, first_tile}
 * (first_tile = running sum of ceil(R/32)*ceil(C/32)), total_tiles = the final running sum. */
int m3ae_cast_transpose_batched(const void* jobs_dev, int njobs, int64_t total_tiles, void* stream);
/* the same from bf16 sources (the weight shadows the optimizer kernel has just written): jobs_dev = device array of
 * {const bf16* in; bf16* out_t; int64 R, C, first_tile} with first_tile = running sum of ceil(R/64)*ceil(C/64).
 * (the reference has no counterpart: weights are used transposed-on-the-fly by torch.nn.functional.linear's backward,
 * /root/reference/m3ae/modules/language_encoders/bert_model.py:419-441) */
int m3ae_transpose_bf16_batched(const void* jobs_dev, int njobs, int64_t total_tiles, void* stream);
/* elementwise: out = cast(in) (dtype_in -> dtype_out), n elements */
int m3ae_cast(const void* in, void* out, int64_t n, int dtype_in, int dtype_out, void* stream);
/* out = a + b (same dtype) */
/* stream-ordered zero fill of `bytes` bytes (the flat gradient buffer before a step; hipMemsetAsync) */
int m3ae_zero(void* p, int64_t bytes, void* stream);
int m3ae_add(const void* a, const void* b, void* out, int64_t n, int dtype, void* stream);
/* dx = dy * act'(x_pre) ; y = act(x) standalone forms */
int m3ae_act_fwd(const void* x, void* y, int64_t n, int act, int dtype, void* stream);
int m3ae_act_bwd(const void* dy, const void* x_pre, void* dx, int64_t n, int act, int dtype, void* stream);
/* rows gather / scatter-add along dim 0 of a [R, D] matrix: out[i] = in[idx[i]] ; d_in[idx[i]] += d_out[i]
 * (MIM masking, m3ae_module.py:170; prediction_heads.py:68). */
int m3ae_gather_rows(const void* in, const int64_t* idx, void* out, int64_t n_out, int64_t D, int dtype,
                     void* stream);
int m3ae_scatter_add_rows(const void* d_out, const int64_t* idx, void* d_in, int64_t n_out, int64_t D, int dtype,
                          void* stream);

/* Masked-image-modelling bookkeeping of pre-training.
 * m3ae_mask_ranks: M3AETransformerSS.random_masking's index work (m3ae_module.py:153-183) for a given noise [B, L]
 *   (fp32): ids_restore[b][j] = rank of patch j (= argsort(argsort(noise)), ties by index); mask[b][j] = 1 for removed
 *   patches (rank >= len_keep); keep_rows [B, len_keep + 1]: flat row ids into the [B * (L + 1)] token rows, class row
 *   first, then the kept patches in rank order (the gather of :176-180).
 * m3ae_mim_targets: patchify (m3ae_module.py:185-192) of img [B, C, H, W] fp32 into out [B, (H/P)(W/P), P*P*C] fp32,
 *   element order (p, q, c); norm_pix != 0 standardises every patch with the unbiased variance + 1e-6
 *   (objectives.py:52-56).
 * m3ae_mim_loss_fwd / _bwd: objectives.py:58-62 on the decoder output x [B, L + 1, D] (class row skipped,
 *   prediction_heads.py:86), target [B, L, D] fp32, mask [B, L] fp32: loss[0] = sum_n mask mean_d (x - t)^2 / sum mask.
 *   acc: fp32[2] scratch owned by the caller, written by fwd (masked error sum, mask sum) and read by bwd;
 *   gout: fp32[1] device scalar (upstream gradient); dx [B, L + 1, D] (class rows zero). */
int m3ae_mask_ranks(const float* noise, int64_t* ids_restore, int64_t* keep_rows, float* mask, int64_t B, int64_t L,
                    int64_t len_keep, void* stream);
int m3ae_mim_targets(const float* img, float* out, int64_t B, int64_t C, int64_t H, int64_t W, int64_t P, int norm_pix,
                     void* stream);
int m3ae_mim_loss_fwd(const void* x, const float* target, const float* mask, float* acc, float* loss, int64_t B, int64_t L,
                      int64_t D, int dtype, void* stream);
int m3ae_mim_loss_bwd(const void* x, const float* target, const float* mask, const float* acc, const float* gout, void* dx,
                      int64_t B, int64_t L, int64_t D, int dtype, void* stream);

/* self-test of hardware idioms the kernels rely on (MFMA fragment maps, ds_read_b64_tr_b16, accumulator-as-
 * operand k-order).  out: int32[8 + 256] device buffer (tail = scratch), out[0] = number of mismatches. */
int m3ae_selftest(int32_t* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* M3AE_HIP_H */
