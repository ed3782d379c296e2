// Internal interface between the C ABI (ltm_capi.hip) and the gfx950 kernels (ltm_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "knobs.h"

namespace infv {

constexpr int kMaxLayers = 8;
constexpr int kHeadSize = 64;      // dh the attend kernel is written for
constexpr int kQTile = 16;         // query rows per attend workgroup (one MFMA row tile)
// Layout of the fixed-point sticky bin-mass accumulators (u64 words, 128 per layer): bin j lives at word acc_word(j), which
// interleaves the two halves of the histogram -- words (2 i, 2 i + 1) = bins (i, i + 64) -- so that ONE 16-byte load per
// lane hands lane i exactly the two bins it owns in the draw (the poll of the persistent chain kernel).
__host__ __device__ inline int acc_word(int j) { return j < 64 ? 2 * j : 2 * (j - 64) + 1; }
// A ring slot holds, per layer, kAccShards replicas of those 128 words ([L][kAccShards][128]); the total of a bin is the
// (integer) sum over the replicas, whoever clears a slot clears all of them.  Round 4 measured 2-8 replicas for the
// persistent role S (adds to one word serialise at the memory side): 1.77-1.95 us per exchange against 1.90 with one --
// every cross-XCD exchange costs that much, replicas buy nothing -- so the ring is back to one replica.  Who uses it: the
// per-chunk kernels, and chain_batch3_kernel<., false> (one launch per sub-batch, atomics exchange).  The call-long launch
// (chain_batch3_kernel<., true>, round 5) exchanges through mailboxes inside one XCD's L2 instead and only reads the
// accumulators when the step before it ran in a per-chunk launch.
constexpr int kAccShards = 1;
constexpr int kCallSets = 8;               // most rotating projection workspace sets a call-long role-S launch can address (kPSets <= this)

// Device-side view of one ridge operator (first-chunk or infinite-memory) of a plan.
struct OperatorView {
    int32_t rows;                 // boxes that receive >=1 new frame
    const int32_t* row_box;       // [rows]
    const int32_t* row_begin;     // [rows]
    const int32_t* row_end;       // [rows]
    const float* box_val;         // [N]  1/(count+ridge)
    const int32_t* box_row;       // [N]  row index of box n, -1 if it receives no new frame
    const int32_t* old_ptr;       // [N+1] or nullptr (first-chunk operator)
    const int32_t* old_slot;      // CSR payload
    const int32_t* slot_tab;      // [N][tabw] dense form of the CSR: slot ids of box n, -1 padded (nullptr if no old)
    int32_t tabw;                 // max slots per box rounded up to a multiple of 4
};

struct StickyView {
    int32_t n_bins;               // 128
    const int32_t* edge_box;      // [n_bins+1]
    const float* edge_dx;         // [n_bins]
    const int32_t* bin_box;       // [n_bins]
    int32_t points_ok;            // edges 1..n_bins-1 lie in the box of the bin's left edge, edges 0 and n_bins in none
};

struct ProjPtrs {
    const float* wk[kMaxLayers];
    const float* bk[kMaxLayers];
    const float* wv[kMaxLayers];
    const float* bv[kMaxLayers];
};

// Row segments of a GEMM's B operand: output column o in [start[s], start[s+1]) reads row o - start[s] of base[s].
constexpr int kMaxSegs = 2 * kMaxLayers + 1;
struct WSegs {
    const float* base[kMaxSegs];
    int start[kMaxSegs + 1];
};

// ---- launchers (all asynchronous on `stream`) -------------------------------------------
hipError_t launch_pool(const void* k, int k_bf16, float* kbar, int64_t n_frames, int P, int d, hipStream_t stream, int lds_pad = 0);

// R[c][r][:] = val * sum of the frames of row r of chunk c;  Pnew[sk][c][r][l][kv][dm] = split-K
// partials of R . W[l][kv]^T  (sk = project_splitk(n_chunks*rows, d) slabs of n_chunks*rows*L*2*dm floats).
// pool + rows in one pass (fast path): R straight from the tokens, bit-identical to launch_pool + launch_rows:
// one short-lived workgroup per (chunk, row), a wave per (frame of the row, 256-float slice); u: 1-KiB loads per burst,
// lds_pad: occupancy cap
bool pool_rows2_supported(int d);
// ONE pooling launch per consolidate call (round 5): workgroup (chunk, row) of the WHOLE call, in order; besides the fp32 row it can
// write the row's three bf16 planes for the projection GEMM (k-tile-major per sub-batch: split3_rows_kernel's layout), and it counts
// itself into its sub-batch's completion word -- the GEMM stream waits on that (flag_wait_kernel) instead of a launch boundary.
struct PoolCallDesc {
    int sub;                        // chunks per sub-batch (the last one may be shorter)
    int n_chunks;                   // chunks of the launch
    float* R_all;                   // [n_chunks][rows][d] fp32
    void* plane[3];                 // bf16 planes, sub-batch b at element offset b * sub * rows * d, or nullptr
    unsigned int* done;             // [sub-batches] rows written (write-through) so far
    int store_mode;                 // stores of the rows / planes when done == nullptr (a kernel boundary is the hand-off): 0 write-through (sc1), 1 plain, 2 nontemporal
};
hipError_t launch_pool_rows2_call(const void* k, int k_bf16, int T, int P, int d, const OperatorView& op, const PoolCallDesc& pc,
                                  hipStream_t stream, int u, int lds_pad);
// planes (or nullptr): the rows' three bf16 planes for the projection GEMM, k-tile-major over the launch's rows (split3's layout);
// *planes_done says whether the launched kernel wrote them (the grid-stride and LDS-DMA variants of the experiments build do not)
hipError_t launch_pool_rows2(const void* k, int k_bf16, int n_chunks, int T, int P, int d, const OperatorView& op, float* R,
                             hipStream_t stream, int u, int lds_pad, int max_wgs = 0, void* const* planes = nullptr, bool* planes_done = nullptr);
int project_splitk(int M, int K);
hipError_t launch_rows(const float* kbar, int n_chunks, int T, int d, const OperatorView& op, float* R,
                       hipStream_t stream);
hipError_t launch_project(int n_chunks, int d, int dm, int n_layers, const OperatorView& op, const ProjPtrs& proj,
                          const float* R, float* Pnew, hipStream_t stream, int lds_pad = 0);

// fast path (consolidate): V half only, scores through the pre-multiplied queries qt
hipError_t launch_project_fast(int M, int d, int dm, int n_layers, int n_out, const ProjPtrs& proj, const float* qt,
                               const float* R, float* C, int* splitk, hipStream_t stream, int lds_pad = 0);
hipError_t launch_project_scores(int M, int d, int n_out, const float* qt, const float* R, float* C, int ldc,
                                 hipStream_t stream, int lds_pad = 0);
hipError_t launch_project_values(int M, int d, int dm, int n_layers, const ProjPtrs& proj, const float* R, float* C,
                                 int ldc, hipStream_t stream, int lds_pad = 0);
hipError_t launch_qtilde(const float* q, int Q, int H, int d, int n_layers, const ProjPtrs& proj, float* qt, float* cq,
                         hipStream_t stream);

// KV[l][n][0][:] = B[l][n] . Wk[l]^T, KV[l][n][1][:] = B[l][n] . Wv[l]^T (no bias).
hipError_t launch_reproject(const float* B, int N, int d, int dm, int n_layers, const ProjPtrs& proj, float* KV,
                            hipStream_t stream);

struct StepDraw {
    int n_layers;                   // 0: no draw (first chunk of a document / uniform resampling)
    const float* bin_part; int parts; const float* probs_override; unsigned override_mask; StickyView sticky;
    const double* u; int S; float* probs_out; int32_t* bins_out; int32_t* idx_out; const int32_t* bins_forced; unsigned forced_mask;
    // optional: the resolved gather table of the step, tab_out[l][e] = source box of entry e of slot_tab (-1: none), so that the
    // update kernel reads ONE table row per box instead of chasing old_ptr -> old_slot -> idx
    const int32_t* slot_tab; int tab_entries; int32_t* tab_out;
};
hipError_t launch_draw(const float* bin_part, int parts, const float* probs_override, unsigned override_mask,
                       const StickyView& sticky, const double* u, int S, int n_layers, float* probs,
                       int32_t* bins, int32_t* idx, hipStream_t stream, const int32_t* bins_forced = nullptr,
                       unsigned forced_mask = 0);

// next[l][n] = val_n * sum_{slots s of box n} prev[l][idx[l][s]] + new row of box n, for B and [K'|V'].
hipError_t launch_update(const OperatorView& op, int N, int d, int dm, int n_layers, int S, const int32_t* idx,
                         int idx_layer_stride, const float* R, const float* Pnew, int splitk,
                         long split_stride, const float* B_prev, const float* KV_prev, float* B_next,
                         float* KV_next, hipStream_t stream, const float* kbar = nullptr /* R == nullptr: rows built from kbar */,
                         const int32_t* tab = nullptr /* [L][N * tabw] resolved source boxes (StepDraw.tab_out): replaces idx */);

// per-call step, first launch: new-row projection with the rows built on the fly + the draw of every layer (one launch)
hipError_t launch_step_project(const float* kbar, int d, int dm, int n_layers, const OperatorView& op, const ProjPtrs& proj,
                               float* Pnew, int* splitk, const StepDraw& draw, hipStream_t stream);

// scores, count-weighted softmax, read-out and the next sticky histogram partials.
int attend_parts(int Q, int H, int N);      // sticky partial rows per layer the attend launch writes (H x query tiles)
hipError_t launch_attend(const float* q, int Q, int N, int H, int n_layers, const float* KV, const ProjPtrs& proj,
                         const float* readout_w, float readout_w_out, const StickyView& sticky, float* ctx,
                         float* bin_part, float* scores, hipStream_t stream);

// ---- whole-video fast path, role S of one chunk per launch (ltm_chain.hip) ----------------------
struct ChainRoleS {
    int n_blocks;                   // H * QS * L, or 0
    OperatorView op;
    int draw_mode;                  // 0: none (first chunk of a document), 1: sticky Gibbs draw, 2: uniform resample
    const float* part_prev; int parts;            // float partials of the previous step (per-call path hand-over) ...
    const unsigned long long* acc_prev;           // ... or its fixed-point totals [L][kAccShards][128] (fast path steady state)
    unsigned long long* acc_next; unsigned long long* acc_clear;   // this step's totals; the ring slot to zero for the next
    const float* probs_override; unsigned override_mask; const double* u; const int32_t* uniform_idx;
    float* probs_out; int32_t* bins_out; int32_t* idx_out;     // [L][128], [L][S], [L][S]  (diagnostics)
    float* probs_tr; int32_t* bins_tr;                         // draw trace rows of this chunk ([L][128], [L][S]) or nullptr
    int32_t* tab_out;               // [L][N*tabw] resolved source box of every (box, slot) for role U
    const float* Sp_prev; float* Sp_next; const float* cq; const float* w; float w_out;
    const float* Snew; int snew_ld; int snew_splitk; long snew_split_stride;   // this chunk's new-row scores: row pitch snew_ld, column (l*H+h)*Q+q, split-K slabs
    float* alpha_out; float* asum_out;
};
struct ChainArgs {
    int N, H, Q, QT, QS, L, S, d4, dm4;   // QS: 8-row query tiles of role S
    StickyView st;
    ChainRoleS s;
    long long* dbg;                 // timing experiments: phase stamps (100 MHz) of one workgroup per role, or nullptr
};
size_t chain_lds_bytes(int N, int S, int rows, int tabw);
bool chain_supported(int N, int S, int rows_max, int tabw);
int chain_s_tiles(int Q);            // 8-row query tiles of role S (= sticky partial rows per head)
hipError_t launch_chain(const ChainArgs& a, hipStream_t stream);
// ---- role S of a whole sub-batch in one persistent launch (ltm_chain_batch.hip) ----
// description of a call-long launch's sub-batches
struct ChainCallDesc {
    const unsigned int* tiles_s;    // [b] S' tiles of sub-batch b the call-long projection GEMM has completed, for b < n_tiled; or nullptr
    int n_tiled, tiles_full, tiles_last;   // sub-batches covered by it; S' tiles of a full sub-batch / of sub-batch n_tiled - 1
    const unsigned int* ready;      // (the other sub-batches) sub-batches whose projection GEMM is complete (flag_set_kernel on the GEMM's stream)
    unsigned int* done;             // every workgroup adds 1 per sub-batch once the steps it published are written back (flag_wait_kernel on the UC stream polls it)
    int sub;                        // steps per sub-batch (the last one may be shorter)
    int n_batches;                  // sub-batches of the launch
    const float* snew_set[kCallSets];   // S'new origin of workspace set s; sub-batch b reads set b % n_sets
    int n_sets;
    int sk_last; long ss_last;      // split-K form of the LAST sub-batch's projection (all others: snew_splitk / snew_split_stride)
    long long* stats;               // [0]: 100 MHz ticks workgroup 0 spent waiting for `ready`; [1]: sub-batches it had to wait for; or nullptr
};
hipError_t launch_chain_call_desc(ChainCallDesc* dst, const ChainCallDesc& v, hipStream_t stream);
struct ChainBatchArgs {
    int N, H, Q, QS, L, S;
    StickyView st;
    OperatorView op;                // infinite-memory operator
    int draw_mode;                  // 1: sticky Gibbs draw, 2: uniform resample
    int n_steps;
    long step0; int ring;           // step i has global index step0+i: histogram slot (step0+i)%3, output slot (step0+i)%ring
    int first_from_parts;           // step 0 reads the float partials of the per-call path instead of the ring
    int first_from_acc;             // chain_batch3_kernel: step 0 reads the fixed-point totals of a per-chunk launch (acc[(step0+2)%3]) instead of the mailboxes
    unsigned long long* mbox;       // chain_batch3_kernel: mailboxes of the exchange + placement handshake (chain_mailbox_bytes), zeroed at the start of a call
    int xcd_grid;                   // chain_batch3_kernel: launched as 8 * (H*QS) blocks, layer l served by placement class (8 l) / L (set by launch_chain_batch)
    int* xcc_report;                // experiments / tests: per workgroup (plain-store mode << 8) | XCC id, or nullptr
    const float* part_prev; int parts;
    unsigned long long* acc[3];     // fixed-point sticky histograms [L][kAccShards][128], ring of 3
    unsigned int* arrive;           // [L] arrival counters, zero at launch
    unsigned int* error;            // host-visible word, set to 1 if a wait timed out
    int spin_limit;                 // polls before a wait gives up
    int poll_delay;                 // chain_batch3_kernel: units of 64 clocks between a step's deposit and its first poll
    int exp_flags;                  // INFV_S_FLAGS (timing experiments): 1 no s_setprio, 2 long sleep between polls, 4 no point-score stores, 8 no loader requests, 16 no exchange (deposit / poll)
    int expect_extra;               // fault injection (tests): arrivals expected beyond the launch's workgroups
    const float* probs_override; unsigned override_mask;     // teacher forcing of step 0
    const double* u;                // [n_steps][L][S]
    const float* uf;                // chain_batch3_kernel: the same uniforms as fp32 round-ups (launch_round_up_uniforms), or nullptr
    const int32_t* uniform_idx;
    float* probs_out; int32_t* bins_out; int32_t* idx_out;    // diagnostics of the last step
    float* probs_tr; int32_t* bins_tr; int trace_steps;       // draw trace of steps [0, trace_steps): [.][L][128], [.][L][S] (either may be null)
    int32_t* tab_ring; long tab_slot;
    int32_t* tabb_ring;             // chain_batch3_kernel: drawn bin of every (box, slot), same slot layout as tab_ring
    float* crit_ring; long crit_slot;   // chain_batch3_kernel: point scores after every step, [ring][L][H][Q][128]
    int publish_init;               // chain_batch3_kernel: also write the state BEFORE step 0 to the slot before slot0's
    float* alpha_ring; long alpha_slot; float* asum_ring; long asum_slot;
    const float* Sp_in; float* Sp_out;                        // [L][H][Q][N] bias-free scores before / after the sub-batch
    const float* Snew; int snew_ld; int snew_splitk; long snew_split_stride;   // [n_steps][rows] rows of pitch snew_ld, column (l*H+h)*Q+q, split-K slabs
    const float* cq; const float* w; float w_out;
    long long* dbg;                 // timing experiments: phase stamps of workgroup 0 at step 5, or nullptr
    long long* wg_stamps;           // residency experiment (wg_stamps.h), or nullptr
    // ---- call-long launch of chain_batch3_kernel: ONE launch per infv_ltm_consolidate, resident from the call's first sub-batch to its
    // last (ready == nullptr: one launch per sub-batch, Snew / snew_splitk / snew_split_stride describe that sub-batch) ----
    const ChainCallDesc* call;      // device memory (written by chain_call_desc_kernel ahead of the launch: the kernel's argument registers are full), or nullptr
    int call_sub;                   // = call->sub (the one field every wave needs every step)
};
// role S -> UC stream / GEMM stream -> role S hand-offs of a call-long launch (ltm_chain_batch.hip)
hipError_t launch_flag_set(unsigned int* flag, unsigned int value, hipStream_t stream);
hipError_t launch_flag_wait(const unsigned int* counter, unsigned int target, int spin_limit, unsigned int* error, hipStream_t stream);
bool launch_flag_wait_available();       // the flag kernels are compiled in (experiments build)
bool chain_batch_supported(int N, int S, int rows, int tabw, int n_blocks);
bool chain_batch2_applies(const ChainBatchArgs& a);        // the launch will run chain_batch3_kernel (scores rebuilt by alpha_rows2)
bool chain_batch3_shape_ok(int draw_mode, int points_ok, int rows, int S, int Q);   // the shape runs chain_batch3_kernel (needs ChainBatchArgs.uf)
hipError_t launch_round_up_uniforms(const double* u, float* uf, long n, hipStream_t stream);   // uf[i] = smallest float >= u[i]
bool chain_batch3_mailboxes();                               // role S exchanges through mailboxes in one XCD's L2 (experiments build, INFV_CHAIN_XCD=0: memory-side atomics)
bool chain_call_long();                                      // role S is ONE launch per consolidate call (experiments build, INFV_CHAIN_CALL=0: one per sub-batch)
size_t chain_mailbox_bytes(int L, int G);                    // G = workgroups of a layer (chain_batch_blocks / L)
// part[l][0][j] = total of bin j held by the mailboxes of parity `parity` (the last step of a chain_batch3 launch), same sum order as the kernel
hipError_t launch_mailbox_to_part(const unsigned long long* mbox, int n_layers, int G, int parity, int parts_pitch, float* part, hipStream_t stream);
int chain_batch_blocks(int H, int Q, int L, int draw_mode, int points_ok, int rows, int S);   // workgroups of the launch for this shape
// the chunk-parallel half of chain_batch3_kernel: full score rows from the published point scores + drawn bins, then alpha
struct AlphaRows2Args {
    int N, H, Q, L, rows, tabw, n_steps;
    long slot0; int ring;
    const float* crit_ring; long crit_slot;
    const int32_t* tabb_ring; long tab_slot;
    const float* Snew; int snew_ld; int snew_splitk; long snew_split_stride;
    const float* cq; const float* w; float w_out;
    const float* box_val; const int32_t* box_row;
    float* alpha_ring; long alpha_slot; float* asum_ring; long asum_slot;
    float* Sp_out;                  // full bias-free score rows of the LAST step [L][H][Q][N], or nullptr
    long long* wg_stamps;           // residency experiment (wg_stamps.h), or nullptr
    int prio;                       // experiments build: s_setprio level of the kernel's waves (0 = leave)
    int regs_ok;                    // set by the launcher: the next unit's inputs fit the kernel's register stage
    long long* dbg;                 // experiments build, INFV_ALPHA_STAMPS: phase time sums (100 MHz ticks), or nullptr
};
hipError_t launch_alpha_rows2(const AlphaRows2Args& a, hipStream_t stream);
bool chain_batch_resident(int N, int S, int rows, int tabw, int n_blocks, int draw_mode, int points_ok, int Q);   // all workgroups of the kernel the launch will use fit on the device at once
hipError_t launch_chain_batch(const ChainBatchArgs& a, hipStream_t stream);

// ---- state update + read-out of a sub-batch in one launch (ltm_uc.hip) ----
struct UcArgs {
    int N, H, Q, L, d, dm, tabw;
    OperatorView op;
    int gather;                     // 0: new rows only (first chunk of a document)
    int have_state;                 // 0: start from an empty memory (first chunk)
    int n_chunks;
    long slot0; int ring;           // chunk i of the launch uses ring slot (slot0 + i) % ring
    const int32_t* tab; long tab_slot;   // [ring] slots of tab_slot ints, each [L][N*tabw]: gather tables written by role S
    const float* alpha;             // [ring][L][H][Q][N]  softmax weights written by role S
    const float* asum;              // [ring][L][H][Q]
    const float* R; const float* Pnew; int p_ld; int splitk; long split_stride;   // new rows of the launch's first chunk onwards; Pnew rows of pitch p_ld: [L][dm] (V' half), split-K slabs
    const float* B_prev; const float* KV_prev; float* B_next; float* KV_next;
    const float* bv[kMaxLayers];
    float* ctx;                     // [n_chunks][L][Q][dm] outputs of the launch's chunks
    long long* dbg;                 // timing experiments: phase stamps of one V' workgroup, or nullptr
    int v16;                        // uc_fast_kernel: V' slices of 16 columns (twice the workgroups, half the read-out each)
    long long* wg_stamps;           // residency experiment (wg_stamps.h), or nullptr
    int prio;                       // experiments build: s_setprio level of the kernel's waves (0 = leave)
};
// scores -> softmax weights + row sums of the ring slots the persistent role S filled (in place)
hipError_t launch_alpha_rows(float* alpha_ring, long alpha_slot, float* asum_ring, long asum_slot, long slot0, int ring,
                             int n_steps, int rows_per_step, int N, const float* w, float w_out, hipStream_t stream);
bool uc_supported(int N, int d, int dm, int tabw, int rows_max);
hipError_t launch_uc(const UcArgs& a, hipStream_t stream);

// S'new[c][l][h][q][r] = (q_h[q]/sqrt(dh)) . Kmat(c,r,l)_h ; optionally cq[l][h][q] = q_h[q].bk_h/sqrt(dh)
hipError_t launch_new_scores(const float* q, int Q, int H, int n_layers, int n_chunks, int rows, const float* Kmat,
                             long chunk_stride, long row_stride, long layer_stride, int splitk, long split_stride,
                             const ProjPtrs& proj, float* Snew, float* cq, hipStream_t stream);

// ---- dense-operator form of the step (ltm_dense.hip): num_basis whose fp32 boxes overlap ----
// B_next[l][n] = sum_r GT[n][r] x_l[r];  x_l[r < n_old] = sum of the memory rows of pos_box2[p(l, r)] (p = bins[l][r] for
// the sticky draw, r itself for the uniform resampling), x_l[r >= n_old] = kbar[r - n_old]
hipError_t launch_dense_update(const float* GT, int K, int ldg, int n_old, const int32_t* bins, int bins_stride,
                               const int32_t* pos_box2, const float* B_prev, const float* kbar, float* B_next, int N, int d,
                               int n_layers, hipStream_t stream);
// part[l][h][j] = sum_q trapezoid mass of interval j+1 of row (h, q)'s 129-edge density; an edge may lie in two boxes
hipError_t launch_dense_masses(const float* scores, int Q, int N, int H, int n_layers, const int32_t* edge_box2,
                               const float* edge_dx, float* part, hipStream_t stream);

// ---- general-psi form of the step (ltm_psi.hip): basis families whose psi(t) is a dense row (the reference's Gaussian family) ----
hipError_t launch_psi_gemm(bool transB, const float* A, int lda, long sA, const float* B, int ldb, long sB, float* C, int ldc, long sC,
                           int M, int Nc, int K, int batch, hipStream_t stream);      // C[z] = A[z] . B[z]^T (transB) or A[z] . B[z], fp32 MFMA tiles
hipError_t launch_psi_update(const float* GT, int K, int ldg, int n_old, const int32_t* bins, int bins_stride, const float* Y, int n_pos,
                             const float* kbar, float* B_next, int N, int d, int n_layers, hipStream_t stream);
hipError_t launch_psi_masses(const float* E, int ldE, int Q, int H, int n_layers, const float* edge_dx, float* part, hipStream_t stream);
hipError_t launch_psi_grid(float* Eg, int ldg, int n_grid, long n_rows, const float* w, hipStream_t stream);
hipError_t launch_psi_ctx(const float* alpha, const float* KV, const ProjPtrs& proj, int Q, int N, int H, int dh, int n_layers, float* ctx,
                          hipStream_t stream);

// part[l][0][j] = acc[l][j] / 2^40 (fast path -> per-call path hand-over of the sticky histogram)
hipError_t launch_acc_to_part(const unsigned long long* acc, int n_layers, int parts_pitch, float* part, hipStream_t stream);
// bin_mass[j] = sum over parts of bin_part[layer][p][j]
hipError_t launch_sum_parts(const float* bin_part_layer, int parts, int pitch, float* bin_mass, hipStream_t stream);

}  // namespace infv
