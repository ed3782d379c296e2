/*
 * infv_ltm.h -- C ABI of the MI355X-native long-term-memory (LTM) consolidation path of
 * infinity-Video (deep-spin/Infinite-Video).  Implemented by libinfv_ltm.so
 * (infinite-video_amd/csrc/, hand-written HIP for gfx950).
 *
 * The reference has no FFI for this path: its operator boundary is the Python nn.Module
 *     LongTermAttention.forward(k, q, new_doc, layer_n)
 *         infty-Video-LLaMA/InfVideoLLaMA/models/long_term_attention_gibbs.py:288-346
 * called once per cross-attention layer per chunk from
 *     BertSelfAttention.forward   infty-Video-LLaMA/InfVideoLLaMA/models/Qformer.py:216-223.
 * The entry points below are what a binding for that boundary needs; each cites the
 * reference lines it replaces.  Plain pointers and sizes only; no torch types.
 *
 * Conventions
 *   - every pointer documented "device" is a HIP device pointer owned by the caller;
 *     the handle owns only the consolidated memory (coefficients B, their projections,
 *     the sticky histogram) and its workspaces;
 *   - all work is enqueued asynchronously on `stream` (a hipStream_t passed as void*,
 *     NULL = the null stream); no entry point synchronises except infv_ltm_sync and the
 *     get calls that copy to HOST buffers;
 *   - a failure that only shows on the device (the persistent chain kernel of
 *     infv_ltm_consolidate timed out waiting for workgroups that never became resident, e.g.
 *     on a partitioned or shared GPU) is latched in host-visible memory: infv_ltm_sync and
 *     every later entry point on the handle return INFV_ERR_STATE once, the memory is reset;
 *   - return value: 0 on success, negative infv_status on failure; nothing throws across
 *     the ABI.  infv_ltm_last_error() returns a thread-local description;
 *   - a handle is not re-entrant (the reference module is mutable, single-threaded state);
 *   - all floating-point data is fp32, row-major, batch size 1 (reference
 *     long_term_attention_gibbs.py:208,346); only the frame tokens may be bf16 (infv_ltm_set_token_dtype).
 */
#ifndef INFV_LTM_H
#define INFV_LTM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define INFV_LTM_ABI_VERSION 5
#define INFV_LTM_MAX_LAYERS 8

typedef enum {
    INFV_OK = 0,
    INFV_ERR_INVALID = -1,      /* bad argument / shape */
    INFV_ERR_UNSUPPORTED = -2,  /* shape outside what the kernels implement */
    INFV_ERR_NO_PLAN = -3,      /* no plan registered for this chunk length */
    INFV_ERR_STATE = -4,        /* call not valid in the handle's current state */
    INFV_ERR_HIP = -5,          /* a HIP runtime call failed (see last_error) */
    INFV_ERR_NOMEM = -6
} infv_status;

typedef struct infv_ltm_s* infv_ltm_handle;

/* Constructor knobs = the kwargs the Q-former passes (Qformer.py:135-158) that the active
 * path reads, plus the frame-pooling shape the reference hard-codes
 * (long_term_attention_gibbs.py:291,304: 32 x 768; VideoChat2 twin: 196 x 1024). */
typedef struct {
    int32_t num_basis;         /* N, attn_num_basis                                    */
    int32_t n_heads;           /* H                                                    */
    int32_t head_size;         /* dh (64)                                              */
    int32_t d_in;              /* d, encoder width of the frame tokens                 */
    int32_t tokens_per_frame;  /* P                                                    */
    int32_t n_layers;          /* LTM instances stepped together by this handle (>=1)  */
    int32_t nb_samples;        /* S = 512 (long_term_attention_gibbs.py:55)            */
    int32_t sticky;            /* sticky_memories                                      */
    int32_t max_q;             /* largest query length that will be passed             */
    int32_t max_batch_chunks;  /* chunks pooled/projected per sub-batch by consolidate */
} infv_ltm_config;

/* Host-built tables for one chunk length T ("plan").  They are the sparse form of what
 * get_basis()/compute_G() rebuild on every forward (long_term_attention_gibbs.py:67-165):
 * with box basis functions F F^T is diagonal, so each row of G has one non-zero
 * 1/(count+ridge).  All pointers are HOST pointers; the call copies them to the device.
 *   rows:  for r in [0,rows): box row_box[r] receives val[box] * sum of frames
 *          [row_begin[r], row_end[r])                      ("new" coefficient rows)
 *   old:   CSR over boxes of the S resampled-memory slots that land in the box
 *          (infinite-memory operator only; positions tau*s/S, :137)                      */
typedef struct {
    int32_t T;
    int32_t first_rows;
    const int32_t* first_row_box;
    const int32_t* first_row_begin;
    const int32_t* first_row_end;
    const float*   first_box_val;     /* [N] */
    int32_t inf_rows;
    const int32_t* inf_row_box;
    const int32_t* inf_row_begin;
    const int32_t* inf_row_end;
    const float*   inf_box_val;       /* [N] */
    const int32_t* inf_old_ptr;       /* [N+1] */
    const int32_t* inf_old_slot;      /* [inf_old_ptr[N]] values in [0,S) */
    const float*   readout_w;         /* [N]  trapezoid weight of the 1000-point grid in box n (:264-282) */
    float          readout_w_out;     /* weight of grid points in no box (t = 1.0)          */
    int32_t        n_bins;            /* 128 (:163)                                          */
    const int32_t* edge_box;          /* [n_bins+1] box evaluated at each modified edge, -1 = none (:197-200) */
    const float*   edge_dx;           /* [n_bins]   fp32 spacing of the modified edges              */
    const int32_t* bin_box;           /* [n_bins]   box of the unmodified left edge of bin b (:207-208) */
    const int32_t* uniform_idx;       /* [S] non-sticky resample rows, -1 = zero row (:153-157,212)  */
} infv_ltm_plan;

/* Borrowed projection weights of one cross-attention layer (device pointers): the layer's
 * own key/value nn.Linear (Qformer.py:133-134,156-157), read at call time. */
typedef struct {
    const float* wk;   /* [dm, d] */
    const float* bk;   /* [dm]    */
    const float* wv;   /* [dm, d] */
    const float* bv;   /* [dm]    */
} infv_ltm_proj;

int         infv_ltm_abi_version(void);
const char* infv_ltm_last_error(void);

/* LongTermAttention.__init__ (long_term_attention_gibbs.py:26-65). */
int infv_ltm_create(const infv_ltm_config* cfg, infv_ltm_handle* out);
int infv_ltm_destroy(infv_ltm_handle h);

/* get_basis() (long_term_attention_gibbs.py:67-165), cached per chunk length. */
int infv_ltm_set_plan(infv_ltm_handle h, const infv_ltm_plan* plan);
int infv_ltm_has_plan(infv_ltm_handle h, int32_t T);          /* 1 / 0 */

/* Dense form of the operators of one chunk length, for num_basis values whose fp32 box bounds
 * (basis_functions.py:248-250) overlap or leave gaps at a sample position, histogram edge or resampling point: psi(t)
 * then has two ones, F F^T is not diagonal and G (long_term_attention_gibbs.py:68-84) has two non-zeros in some rows.
 * The host computes G with the reference's own fp32 sequence (F^T (F F^T + ridge I)^-1 via LAPACK, trimmed) and hands
 * it over transposed, together with the (up to two) boxes of every point the step evaluates psi at.  Registered IN
 * ADDITION to infv_ltm_set_plan for the same T (whose readout_w / edge_dx / n_bins stay in force; its one-box tables
 * are ignored).  With a dense plan every entry point runs the per-call step with dense kernels (update = x . G on fp32
 * MFMA); infv_ltm_consolidate loops over chunks.  All pointers are HOST pointers. */
typedef struct {
    int32_t T;
    int32_t first_K;                  /* rows of the first-chunk operator (= T)                       */
    const float*   first_GT;          /* [N][first_K]  G_first transposed                             */
    int32_t inf_K;                    /* rows of the infinite-memory operator (= S + T)               */
    const float*   inf_GT;            /* [N][inf_K]    G_inf transposed; rows 0..S-1 = resampled, then the T frames */
    const int32_t* bin_box2;          /* [n_bins][2]   boxes containing the unmodified left edge of bin b, -1 = none (:207-208) */
    const int32_t* edge_box2;         /* [n_bins+1][2] boxes containing each modified histogram edge (:197-200)      */
    const int32_t* uniform_box2;      /* [S][2]        boxes of the non-sticky resample positions (:153-157,212)     */
} infv_ltm_dense_plan;
int infv_ltm_set_dense_plan(infv_ltm_handle h, const infv_ltm_dense_plan* plan);

/* General-psi plan of chunk length T, on top of its dense plan: a basis family whose psi(t) is a dense row -- the reference's
 * GaussianBasisFunctions (basis_functions.py:135-164, built by add_gaussian_basis_functions,
 * long_term_attention_gibbs.py:167-174).  Besides the dense ridge operators (infv_ltm_set_dense_plan: compute_G :68-84) the
 * step then needs psi itself wherever the reference evaluates it (batch_evaluate / evaluate):
 *   psi_edge    at the modified histogram edges (update_inf :197-200 -> score :224-230): the sticky density
 *   psi_bin     at the unmodified left edge of every bin, ts = bins[b] (:207-208): the resampled rows B_past^T psi(ts)
 *   psi_uniform at the non-sticky resample positions (get_basis :153-157, used at :212)
 *   psi_grid    on linspace(0, 1, n_grid) with the trapezoid weights grid_w (expected_value :251-286): the read-out
 * Steps of that length take the per-call path with dense contractions for all of these (ltm_psi.hip).  All arrays are
 * host pointers, copied. */
typedef struct {
    int32_t T;
    int32_t n_grid;                   /* points of the read-out grid (1000, long_term_attention_gibbs.py:251) */
    const float* psi_edge;            /* [n_bins+1][N] */
    const float* psi_bin;             /* [n_bins][N]   */
    const float* psi_uniform;         /* [S][N]        */
    const float* psi_grid;            /* [n_grid][N]   */
    const float* grid_w;              /* [n_grid]      */
} infv_ltm_psi_plan;
int infv_ltm_set_psi_plan(infv_ltm_handle h, const infv_ltm_psi_plan* plan);

/* new_doc=True (long_term_attention_gibbs.py:300-302): forget the memory. */
int infv_ltm_reset(infv_ltm_handle h);
int infv_ltm_has_memory(infv_ltm_handle h);                    /* 1 / 0 */

/* Element type of the frame tokens `k` handed to pool / forward / consolidate.  The reference's tokens are fp32
 * (infinityqa.py:317-322 concatenates the image Q-former's fp32 outputs).  A producer that stores them as bf16
 * halves the bytes of the only HBM-heavy stream of the path; the pooled frames and everything after stay fp32,
 * so results differ from the fp32-token run only by the rounding of the tokens themselves (2^-9 relative each,
 * averaged over P tokens).  Default fp32; set before the first call that takes `k`. */
typedef enum { INFV_TOKENS_F32 = 0, INFV_TOKENS_BF16 = 1 } infv_token_dtype;
int infv_ltm_set_token_dtype(infv_ltm_handle h, int32_t dtype);

/* Frame mean-pool, long_term_attention_gibbs.py:304:  k [n_frames, P, d] -> kbar [n_frames, d] (fp32). */
int infv_ltm_pool(infv_ltm_handle h, const void* k, int64_t n_frames, float* kbar, void* stream);

/* Frame mean-pool and the memory's new rows in one pass (what infv_ltm_consolidate runs per sub-batch): for each of
 * n_chunks chunks of T frames, R[c][r] = sum over the frames f of box row r of val_r * mean_P(k[c][f]) -- the rows
 * `G_inf^T`'s new-signal half contributes in update_inf (long_term_attention_gibbs.py:216 applied to the means of :304);
 * k [n_chunks, T*P, d], R [n_chunks, rows, d] fp32 with rows = infv_ltm_new_rows(h, T).  Same bits as infv_ltm_pool
 * followed by the per-row sums of infv_ltm_step.  Needs infv_ltm_set_plan(T) (sparse plans only). */
int infv_ltm_pool_rows(infv_ltm_handle h, const void* k, int32_t n_chunks, int32_t T, float* R, void* stream);
int infv_ltm_new_rows(infv_ltm_handle h, int32_t T);           /* rows per chunk of the plan for T, or a negative error */

/* One consolidation step of all n_layers instances on one chunk, from pooled frames:
 * update_inf + proj_key/proj_value + expected_value (long_term_attention_gibbs.py:194-222,
 * 312-318).  kbar [T,d]; q [L,Q,dm]; proj[L]; u [L,S] float64 Gibbs uniforms (device;
 * may be NULL when the handle has no memory yet or is not sticky); ctx [L,Q,dm]. */
int infv_ltm_step(infv_ltm_handle h, const float* kbar, int32_t T, const float* q, int32_t Q,
                  const infv_ltm_proj* proj, const double* u, float* ctx, void* stream);

/* `n_chunks` consecutive consolidation steps with a DIFFERENT query per chunk (what a cross-attention layer
 * after the first sees: its query is self.query(hidden_states), Qformer.py:211, and hidden_states depend on the
 * chunk).  kbar [C,T,d]; q [C,L,Q,dm]; u [C,L,S] (rows of chunks that resample nothing are ignored; may be NULL
 * when not sticky); ctx [C,L,Q,dm].  The new-row projections of all chunks are batched into one GEMM; the
 * memory chain (draw, update, attend per chunk) stays sequential.  Equals C calls of infv_ltm_step. */
int infv_ltm_steps(infv_ltm_handle h, const float* kbar, int32_t n_chunks, int32_t T, const float* q, int32_t Q,
                   const infv_ltm_proj* proj, const double* u, float* ctx, void* stream);

/* The per-chunk loop with per-chunk queries: infv_ltm_consolidate's arguments except q [C,L,Q,dm].
 * reset if new_doc, pool every chunk (batched), infv_ltm_steps. */
int infv_ltm_consolidate_q(infv_ltm_handle h, const void* k, int32_t n_chunks, int32_t T, const float* q,
                           int32_t Q, const infv_ltm_proj* proj, const double* u, int32_t new_doc, float* ctx,
                           void* stream);

/* LongTermAttention.forward (long_term_attention_gibbs.py:288-346) for all layers of the
 * handle on one chunk: reset if new_doc, pool, step.  k [T*P, d]. */
int infv_ltm_forward(infv_ltm_handle h, const void* k, int32_t T, const float* q, int32_t Q,
                     const infv_ltm_proj* proj, const double* u, int32_t new_doc, float* ctx,
                     void* stream);

/* infv_ltm_forward for a caller that keeps the pooled frames: pool k (element type `token_dtype`, an infv_token_dtype) into the
 * caller's DEVICE buffer kbar [T,d] (long_term_attention_gibbs.py:304), then step from it (:306-346) -- infv_ltm_set_token_dtype +
 * infv_ltm_pool + infv_ltm_step behind ONE call (round 6: the drop-in module's steady-state forward was three ctypes calls; its
 * second cross-attention layer of the same chunk steps from the same kbar without pooling again).  No reset: new_doc is the
 * caller's infv_ltm_reset. */
int infv_ltm_forward_into(infv_ltm_handle h, const void* k, int32_t token_dtype, int32_t T, float* kbar, const float* q, int32_t Q,
                          const infv_ltm_proj* proj, const double* u, float* ctx, void* stream);

/* The per-chunk loop of the eval drivers (run_inference_inf_video_llama_nextqa.py:179-196)
 * with the LLM/Q-former stubbed: n_chunks chunks of T frames, k [C, T*P, d], the same
 * q [L,Q,dm] for every chunk, u [C,L,S], ctx [C,L,Q,dm].  new_doc applies to chunk 0.
 * Pooling and new-row projections are batched over chunks; the memory chain is sequential. */
int infv_ltm_consolidate(infv_ltm_handle h, const void* k, int32_t n_chunks, int32_t T,
                         const float* q, int32_t Q, const infv_ltm_proj* proj, const double* u,
                         int32_t new_doc, float* ctx, void* stream);

/* infv_ltm_consolidate from frame means the caller already holds: kbar [C,T,d] fp32 instead of k.  Same results bit
 * for bit as infv_ltm_consolidate on tokens whose infv_ltm_pool output is kbar; the pooling stage and its stream drop
 * out.  The video Q-former's single pass over the frame tokens (split + transpose + mean, Qformer.py:236,278-291)
 * produces these means, so the tokens are not read a second time for the memory. */
int infv_ltm_consolidate_pooled(infv_ltm_handle h, const float* kbar, int32_t n_chunks, int32_t T,
                                const float* q, int32_t Q, const infv_ltm_proj* proj, const double* u,
                                int32_t new_doc, float* ctx, void* stream);

/* Consolidated memory of one layer: B_past [N,d] (long_term_attention_gibbs.py:220) and the
 * unnormalised sticky bin masses p[n_bins-1] derived from the last scores (:200-202).
 * Export copies device -> caller's DEVICE buffers (async on stream). */
int infv_ltm_export_state(infv_ltm_handle h, int32_t layer, float* B, float* bin_mass, void* stream);
/* Import + recompute the projected memory with the given weights. */
int infv_ltm_import_state(infv_ltm_handle h, int32_t layer, const float* B, const float* bin_mass,
                          const infv_ltm_proj* proj, void* stream);
/* Exact hand-off of the memory chain between handles (multi-GPU correctness mode: rank r continues where rank r-1
 * stopped, reference semantics of one process walking the whole video, long_term_attention_gibbs.py:194-222).
 * The blob is a 64-byte header (magic, version, L, N, d, dm, H, Q, n_bins as int32) and everything infv_ltm_consolidate
 * carries from chunk to chunk, in fp32:
 *   B [L][N][d] | projected memory [L][N][2][dm] | bias-free scores under the call's query [L][H][Q][N] | sticky bin
 *   masses [L][n_bins]
 * so a consolidate call that follows an import (new_doc = 0) continues BIT FOR BIT as if the exporting handle had gone on
 * itself -- unlike export_state/import_state, which hand over B and the masses only and re-derive the rest (same values up
 * to fp32 rounding).  Conditions, none of which the library can check for the caller: the SAME q and proj on both sides (the
 * blob carries the scores under the exporter's query), and blocks that do not end in a short sub-batch -- a sub-batch of
 * fewer than 1024 new rows (16 chunks at the headline shape) takes the split-K form of the projection, whose sums are
 * ordered differently (same values to fp32 rounding, not the same bits; tests/test_sharding_gpu.py cuts 129 chunks as 65 + 64).
 * import checks the header on the device: a blob of another shape latches an error that the next entry point returns as
 * INFV_ERR_STATE (the memory is reset).  export needs a memory whose last step ran in infv_ltm_consolidate with query
 * length Q; blob = DEVICE buffer of infv_ltm_chain_state_bytes(h, Q) bytes; both are asynchronous on `stream`. */
int64_t infv_ltm_chain_state_bytes(infv_ltm_handle h, int32_t Q);
int infv_ltm_export_chain_state(infv_ltm_handle h, int32_t Q, void* blob, void* stream);
int infv_ltm_import_chain_state(infv_ltm_handle h, int32_t Q, const void* blob, void* stream);
/* Recompute projected memory of every layer from B (weights changed since the last step). */
int infv_ltm_reproject(infv_ltm_handle h, const infv_ltm_proj* proj, void* stream);

/* Diagnostics of the last step of one layer, copied to HOST buffers (synchronises stream):
 * bins [S] drawn histogram bins, idx [S] resampled rows, probs [n_bins-1] as fed to the draw,
 * scores [H,Q,N] (any pointer may be NULL). */
int infv_ltm_get_draw(infv_ltm_handle h, int32_t layer, int32_t* bins, int32_t* idx, float* probs,
                      float* scores, void* stream);
/* Teacher forcing for tests: the next step of `layer` draws from these HOST probs[n_bins-1]
 * instead of the ones derived from its own scores (one-shot). */
int infv_ltm_set_probs(infv_ltm_handle h, int32_t layer, const float* probs);
/* Forced draw (one-shot, per-call step/forward only): the next step of `layer` resamples the
 * rows of these HOST bins[S] (values in [0, n_bins-1)) instead of its own draw.  The step still
 * derives its own probabilities and draw; infv_ltm_get_draw then returns the step's OWN bins
 * next to the idx actually used, so a long chain can be compared draw by draw with another path
 * without the two diverging at the first uniform that falls within rounding of a cdf edge. */
int infv_ltm_set_bins(infv_ltm_handle h, int32_t layer, const int32_t* bins);
/* Draw trace of infv_ltm_consolidate: while set, chunk c of a call (c < capacity_chunks) writes
 * the bins it drew to bins_all[c][L][S] and its probabilities to probs_all[c][L][n_bins]
 * (DEVICE buffers owned by the caller, either may be NULL; rows of a document's first chunk,
 * which draws nothing, are left untouched).  capacity_chunks = 0 clears the trace. */
int infv_ltm_set_trace(infv_ltm_handle h, int32_t* bins_all, float* probs_all, int64_t capacity_chunks);

/* Synchronises `stream`, then reports a latched device-side failure (see Conventions). */
int infv_ltm_sync(infv_ltm_handle h, void* stream);

/* Measurement (bench.py's roofline leg): while enabled, every kernel launch the handle issues
 * is bracketed by HIP events on the launch stream.  profile_read synchronises, returns the
 * number of launches and their summed device time for one kernel family, and clears it. */
typedef enum {
    INFV_KERNEL_POOL = 0,     /* frame mean-pool (the HBM-bound kernel)   */
    INFV_KERNEL_ROWS = 1,     /* new coefficient rows                     */
    INFV_KERNEL_PROJECT = 2,  /* fp32 MFMA projection GEMM                */
    INFV_KERNEL_DRAW = 3,     /* Gibbs draw                               */
    INFV_KERNEL_UPDATE = 4,   /* memory update                            */
    INFV_KERNEL_ATTEND = 5,   /* scores / softmax / read-out              */
    INFV_KERNEL_SCORES = 6,   /* batched new-row scores (fast path)       */
    INFV_KERNEL_CHAIN = 7,    /* per-chunk chain step: draw, score recurrence, softmax weights (fast path) */
    INFV_KERNEL_UC = 8,       /* state update + read-out of a sub-batch (fast path) */
    INFV_KERNEL_COUNT = 9
} infv_kernel;
int infv_ltm_profile_enable(infv_ltm_handle h, int32_t on);
int infv_ltm_profile_read(infv_ltm_handle h, int32_t kernel, int64_t* launches, double* total_ms);
/* Kernel launches this library has issued in this process so far (every handle, both headers' entry points): a
 * benchmark reads it before and after a call to state "launches per chunk".  Host-side counter, no device work. */
int64_t infv_ltm_launch_count(void);

#ifdef __cplusplus
}
#endif
#endif /* INFV_LTM_H */
