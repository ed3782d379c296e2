/* infv_vqf.h -- C ABI of the video Q-former path around the LTM (libinfv_ltm.so), MI355X / gfx950.
 *
 * The reference has no FFI for this path either: the boundary is Python (`video_Qformer.bert(...)` called by
 * `encode_video`, infty-Video-LLaMA/InfVideoLLaMA/models/infinityqa.py:280-344).  These entry points are what the
 * Python mirror (infinite_video_amd/video_qformer.py) binds with ctypes.  Plain pointers and sizes only; all
 * pointers are DEVICE pointers (fp32, row-major) unless stated; every call is asynchronous on `stream`.
 * Errors: int status as in infv_ltm.h, message via infv_ltm_last_error().
 *
 * Scope: the query-token-only, eval-mode, all-ones-mask case the video Q-former is run in
 * (Qformer.py:197-312 BertSelfAttention, :322-326 BertSelfOutput, :442-522 BertLayer with the query FFN,
 * :85-112 BertEmbeddings on query_embeds), head size 64, <= 32 query tokens, tokens_per_frame a multiple of 32.
 */
#ifndef INFV_VQF_H
#define INFV_VQF_H

#include "infv_ltm.h"

#ifdef __cplusplus
extern "C" {
#endif

#define INFV_VQF_MAX_LAYERS 8

typedef struct infv_vqf_s* infv_vqf_handle;

typedef struct {
    int32_t n_layers;          /* BertConfig.num_hidden_layers of the video Q-former (infinityqa.py:38: 2) */
    int32_t n_heads;           /* 12 */
    int32_t hidden;            /* 768 = n_heads * 64 */
    int32_t inter;             /* 3072, intermediate_query width */
    int32_t enc_width;         /* width of the frame tokens (encoder_width, infinityqa.py:39) */
    int32_t tokens_per_frame;  /* 32 */
    int32_t n_query;           /* video query tokens (32) */
    int32_t proj_out;          /* llama_proj output width (4096); 0 = no projection */
    int32_t nb_samples;        /* S = 512: stride of the per-layer Gibbs uniforms in `u` */
    float   alpha;             /* merge weight of the short-term context (Qformer.py:129,304) */
    float   ln_eps;            /* BertConfig.layer_norm_eps = 1e-12 */
} infv_vqf_config;

typedef struct { const float* w; const float* b; } infv_linear;          /* nn.Linear: w [out][in], b [out] */
typedef struct { const float* gamma; const float* beta; } infv_layernorm;

/* One BertLayer of the video Q-former (Qformer.py:420-441; the text FFN is removed, infinityqa.py:206-208). */
typedef struct {
    infv_linear self_q, self_k, self_v, self_o;   infv_layernorm self_ln;   /* layer.attention            */
    infv_linear x_q, x_k, x_v, x_o;               infv_layernorm x_ln;      /* layer.crossattention       */
    infv_linear ffn_in, ffn_out;                  infv_layernorm ffn_ln;    /* intermediate_query / output_query */
} infv_vqf_layer;

typedef struct {
    const float*   query_tokens;                  /* video_query_tokens [n_query][hidden] (infinityqa.py:52-55) */
    infv_layernorm emb_ln;                        /* bert.embeddings.LayerNorm */
    infv_vqf_layer layer[INFV_VQF_MAX_LAYERS];
    infv_linear    llama_proj;                    /* infinityqa.py:342 */
} infv_vqf_weights;

int infv_vqf_create(const infv_vqf_config* cfg, infv_vqf_handle* out);
int infv_vqf_destroy(infv_vqf_handle h);

/* Arithmetic of the two big contractions of the short-term attention ([H*Q x d x T*P] each):
 *   0 (default)  split-bf16: every fp32 operand as hi + lo bf16, three bf16 MFMA products, fp32 accumulation
 *                (error ~1e-5 relative, inside the path's 1e-3 budget, ~2.5x faster);
 *   1            exact fp32 MFMA (bitwise an fp32 fma chain). */
int infv_vqf_set_precision(infv_vqf_handle h, int32_t exact_fp32);

/* Layer 0's hidden states entering the cross-attention come from the learned query tokens and the weights only, so
 * infv_vqf_encode_chunk can reuse them (embedding LayerNorm, self-attention block, cross query, pre-multiplied query)
 * from one chunk to the next.  The host vouches for the weights with an epoch: while the same non-zero epoch is set,
 * the cached prefix is reused; change it whenever any weight of that prefix changes; 0 (default) disables reuse. */
int infv_vqf_set_weights_epoch(infv_vqf_handle h, uint64_t epoch);

/* Short-term cross-attention of one layer over one chunk's frame tokens, merged with the long-term context:
 *   merged = alpha * softmax((xq W-free restatement, see vqf_kernels.hip)) ... = Qformer.py:232-304 for a cross layer.
 * frames [n_tokens][enc_width], xq [n_query][hidden] (= self.query(hidden_states), bias applied),
 * a_long [n_query][hidden] or NULL (then merged = short-term context, the alpha == 1.0 / image-Q-former case). */
int infv_vqf_short_attention(infv_vqf_handle h, const float* frames, int32_t n_tokens, const float* xq,
                             const infv_linear* key, const infv_linear* value, const float* a_long,
                             float* merged, void* stream);

/* One chunk through the whole video Q-former + llama_proj (the device side of encode_video, infinityqa.py:325-343):
 *   ltm[l]      LTM handle of cross-attention layer l (n_layers = 1 each, plan for T set); ignored if alpha == 1
 *   frames      [T * tokens_per_frame][enc_width]
 *   u           [n_layers][nb_samples] float64 Gibbs uniforms (device) or NULL (first chunk / non-sticky / alpha == 1)
 *   new_video   resets the memories first (Qformer.py:221 new_doc=new_video)
 *   hidden_out  [n_query][hidden]    last_hidden_state        (may be NULL)
 *   llama_out   [n_query][proj_out]  llama_proj(last_hidden)  (may be NULL)
 * The chunk's frame tokens are read ONCE: one pass yields the split-bf16 operands of every layer's short-term attention
 * (Qformer.py:278-291) and the frame means every layer's memory pools (long_term_attention_gibbs.py:304). */
int infv_vqf_encode_chunk(infv_vqf_handle h, const infv_ltm_handle* ltm, const float* frames, int32_t T,
                          const infv_vqf_weights* w, const double* u, int32_t new_video,
                          float* hidden_out, float* llama_out, void* stream);

/* A whole video (n_chunks chunks of T frames each) through the video Q-former, layer-major:
 *   - layer 0's hidden states do not depend on the chunk (they come from the learned query tokens only), so its LTM
 *     runs as ONE infv_ltm_consolidate over all chunks (constant query) and its short-term attention shares one
 *     pre-multiplied query block;
 *   - later layers have per-chunk queries: their LTM is the sequential per-call chain, issued on an internal side
 *     stream while the caller's stream runs the layer's short-term attention for all chunks;
 *   - every query-token block (linear / LayerNorm / GELU / self-attention) is batched over chunks;
 *   - the frame tokens of the whole video are read once: their split-bf16 copies (2 x the tokens' bytes) are kept for the
 *     call when they fit INFV_VQF_SPLIT_CACHE_GB (default 64), and layer 0's memory runs from the frame means of that pass
 *     (infv_ltm_consolidate_pooled).
 * Results equal n_chunks calls of infv_vqf_encode_chunk (new_video on the first only).
 *   frames [n_chunks][T * tokens_per_frame][enc_width], u [n_chunks][n_layers][nb_samples] or NULL,
 *   hidden_out [n_chunks][n_query][hidden] / llama_out [n_chunks][n_query][proj_out] / llama_mean [n_query][proj_out]
 *   (the eval loop's mean over chunks, run_inference_inf_video_llama_nextqa.py:194); each may be NULL. */
int infv_vqf_encode_video(infv_vqf_handle h, const infv_ltm_handle* ltm, const float* frames, int32_t n_chunks,
                          int32_t T, const infv_vqf_weights* w, const double* u, int32_t new_video,
                          float* hidden_out, float* llama_out, float* llama_mean, void* stream);

/* out[i] = mean over n of in[n][i]  (the eval loop's mean over chunk embeddings,
 * run_inference_inf_video_llama_nextqa.py:194) */
int infv_vqf_mean(const float* in, int32_t n, int64_t elems, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
