// C ABI of libinfv_ltm.so (include/infv_ltm.h): handle, plans, state and the launch sequences.
#include "../../include/infv_ltm.h"
#include "ltm_internal.h"
#include "capi_common.h"
#include "vqf_internal.h"

#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <vector>

using namespace infv;

namespace infv {
thread_local char g_err[512] = "";
std::atomic<long long> g_kernel_launches{0};

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
}  // namespace infv

namespace {

// Timing experiments only (results become garbage): INFV_SKIP bit 0 = no pooling launches, bit 1 = no projection GEMM,
// bit 2 = no UC launches, bit 3 = no role-S launches.  Shows how much each stream slows the others.
// INFV_VPROJ_ON_UC (default on): in the sub-batch pipeline only the SCORE half of the new-row projection stays on the
// side stream (role S waits for it); the V' half, which only the UC kernel reads, is issued on the UC stream right
// before that kernel -- the side stream stops being the longest of the three.
// (the per-handle choice: infv_ltm_s::vproj_on_uc)

int skip_mask() { static const int m = [] { const char* e = exp_env("INFV_SKIP"); return e ? atoi(e) : 0; }(); return m; }

struct Operator {
    int rows = 0;
    DeviceBuf row_box, row_begin, row_end, box_val, box_row, old_ptr, old_slot, slot_tab;
    bool has_old = false;
    int tabw = 4;                      // max slots per box, rounded up to a multiple of 4
    OperatorView view() const {
        OperatorView v;
        v.rows = rows;
        v.row_box = row_box.as<int32_t>();
        v.row_begin = row_begin.as<int32_t>();
        v.row_end = row_end.as<int32_t>();
        v.box_val = box_val.as<float>();
        v.box_row = box_row.as<int32_t>();
        v.old_ptr = has_old ? old_ptr.as<int32_t>() : nullptr;
        v.old_slot = has_old ? old_slot.as<int32_t>() : nullptr;
        v.slot_tab = has_old ? slot_tab.as<int32_t>() : nullptr;
        v.tabw = tabw;
        return v;
    }
};

// Dense form of a plan's operators (infv_ltm_set_dense_plan): num_basis whose fp32 boxes overlap
struct DensePlan {
    bool on = false;
    int first_K = 0, inf_K = 0;
    DeviceBuf first_GT, inf_GT;        // [N][K]
    DeviceBuf bin_box2, edge_box2, uniform_box2;
    // general-psi plan (infv_ltm_set_psi_plan): psi where the reference evaluates it, and the step's scratch
    bool psi_on = false;
    int n_grid = 0;
    DeviceBuf psi_edge, psi_bin, psi_uniform, psi_grid, grid_w;
};

struct Plan {
    int T = 0;
    DensePlan dense;
    Operator first, inf;
    DeviceBuf w, edge_box, edge_dx, bin_box, uniform_idx;
    float w_out = 0.f;
    int n_bins = 0;
    bool points_ok = false;            // the histogram edges are the bins' left edges (what chain_batch3_kernel assumes)
    StickyView sticky() const {
        StickyView s;
        s.points_ok = points_ok ? 1 : 0;
        s.n_bins = n_bins;
        s.edge_box = edge_box.as<int32_t>();
        s.edge_dx = edge_dx.as<float>();
        s.bin_box = bin_box.as<int32_t>();
        return s;
    }
};

struct Profiler {
    bool on = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[INFV_KERNEL_COUNT];
    ~Profiler() { clear_all(); }
    void clear(int k) {
        for (auto& p : ev[k]) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
        ev[k].clear();
    }
    void clear_all() { for (int k = 0; k < INFV_KERNEL_COUNT; ++k) clear(k); }
};

// Brackets one launch with events when profiling is on.
struct Timed {
    Profiler& prof; int kernel; hipStream_t stream; hipEvent_t a = nullptr, b = nullptr;
    Timed(Profiler& p, int k, hipStream_t s) : prof(p), kernel(k), stream(s) {
        if (prof.on && hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess) (void)hipEventRecord(a, stream);
    }
    ~Timed() {
        if (prof.on && a && b) { (void)hipEventRecord(b, stream); prof.ev[kernel].emplace_back(a, b); }
    }
};

}  // namespace

// New-row buffers R of the sub-batch pipeline: the pooling stream fills set b % kRSets for sub-batch b and may run that many
// sub-batches ahead of the UC kernel, the last reader of a set (three sets coupled the pooling to the UC stream's progress).
constexpr int kRSets = 8;
#ifndef INFV_PSETS
#define INFV_PSETS 5
#endif
constexpr int kPSets = INFV_PSETS;            // rotating sets of the sub-batch workspaces (projections, new-row scores, pooled frames) and their events

struct infv_ltm_s {
    infv_ltm_config cfg;
    int N, H, dh, d, dm, P, L, S, maxQ, maxC;
    std::map<int, Plan*> plans;
    DeviceBuf B[2], KV[2];             // [L][N][d], [L][N][2][dm]   ping-pong
    int cur = 0;
    bool has_memory = false;
    int lastQ = 0;                     // query length of the last attend
    bool last_fast = false;            // last step ran in the fused chain kernel (scores = Sp + cq)
    bool k_stale = false;              // the K' half of KV is out of date (the fast path does not maintain it)
    bool carry_scores = false;         // one-shot (import_chain_state): the next consolidate continues from Sp / V' as they are
    int parts = 0;                     // row count of bin_part per layer, set by the last attend
    DeviceBuf bin_part[2];             // [L][max_parts][n_bins]  sticky partials (ping-pong in the fast path)
    int pc = 0;                        // which bin_part holds the latest partials
    DeviceBuf probs, probs_override, bins, idx, scores;
    unsigned override_mask = 0;        // layers whose next draw uses probs_override (teacher forcing)
    // workspaces of the chunk-parallel stage, two sets: consolidate() fills set b&1 for sub-batch b on a side
    // stream while the chain of sub-batch b-1 runs on the caller's stream
    DeviceBuf kbar_ws, kbar_side[kPSets], R_ws[kRSets + 1], P_ws[kPSets], Snew_ws[kPSets];   // R: sets 0..kRSets-1 rotate over the sub-batches, kRSets = first chunk of a document
    DeviceBuf kbar_all;                // pooled frames of a whole consolidate_q call
    DeviceBuf wv_hi, wv_lo, R_hi, R_lo;  // split-bf16 operands of the V' half of the new-row projection (fast path)
    bool wv_split_valid = false;         // the value weights of this consolidate call have been split
    // The projection GEMM of the whole-video path runs on the bf16 MFMA pipe, fp32-accurate (three exact bf16 pieces per operand,
    // six products, fp32 accumulation: split_gemm.hip, gemm_x6_wide_kernel); INFV_PROJ_X6=0 at create selects gemm_nt_lw_kernel's
    // fp32 MFMAs instead (rounds 1-3's default).  Planes of [Wv ; q~] are made once per call, of a sub-batch's new rows per launch.
    DeviceBuf w3[3], r3[3];
    DeviceBuf r3_ring[kRSets][3];      // bf16 planes of the new rows written by the pooling kernel itself, one set per R set
    bool proj_x6 = true, w3_valid = false;
    hipStream_t side = nullptr;
    hipStream_t pools = nullptr;        // stream of the pooling kernels (HBM-bound; runs ahead of the GEMM stream)
    hipEvent_t ev_pool[kPSets] = {};
    hipEvent_t ev_r[kRSets] = {};      // the UC kernel that read R set i is done
    hipEvent_t ev_in = nullptr, ev_start = nullptr, ev_q = nullptr, ev_p[kPSets] = {};
    // fast path (consolidate): bias-free scores (ping-pong), softmax weights + row sums (ring of 3,
    // read two launches later), resolved gather tables (ring of 2, read one launch later)
    DeviceBuf Sp[2], cqbuf;
    DeviceBuf qt_buf;                  // fast path: pre-multiplied queries qt[(l*H+h)*Q+q][d] of the current call
    DeviceBuf alpha_ring, asum_ring, tab_ring;   // per-chunk outputs of role S for the UC kernel: ring of 2*maxC+2 slots
    DeviceBuf crit_ring, tabb_ring;              // chain_batch3_kernel -> alpha_rows2_kernel: point scores, drawn-bin tables
    int ring = 0;
    hipStream_t ucs = nullptr;          // stream of the UC kernels (state update + read-out of a sub-batch)
    hipStream_t chain_s = nullptr;      // role S's own stream (highest priority: a hardware queue it shares with no worker stream), where the call-long launch lives
    hipEvent_t ev_s[kPSets] = {}, ev_uc[kPSets] = {};
    DeviceBuf sync_words;              // [0..7] arrival counters per layer
    // error word of the persistent chain kernel: pinned host memory mapped into the device, so a time-out is
    // visible to the host without any synchronisation or copy
    unsigned int* err_host = nullptr; unsigned int* err_dev = nullptr;
    int k_bf16 = 0;                     // frame tokens arrive as bf16 (infv_ltm_set_token_dtype)
    bool v_split = false;               // INFV_VPROJ_SPLIT=1 at create: V' half of the sub-batch projection as a split-bf16 contraction
    // Where the V' half of a sub-batch's projection runs.  Default (round 3): inside the side stream's one [V' | S'] GEMM.  The
    // UC stream (V' GEMM -> softmax weights -> update, serial per sub-batch) was the second-longest stream of the pipeline
    // and a GEMM workgroup cannot share a CU with a UC workgroup (272 + 320 of the 512 registers per SIMD), so the V' GEMM
    // only ever delayed the UC kernel behind it; with 16-row chain tiles (48 role-S workgroups instead of 96) the merged
    // GEMM has the CUs to finish well before the chain needs its score half.  Round 2 ran long calls (>= 768 chunks) with the
    // V' half on the UC stream (INFV_VPROJ_ON_UC=1 restores that; the split-bf16 V' projection implies it).
    int v_on_uc_mode = -1;
    bool vproj_on_uc(int /*n_chunks*/) const { return v_on_uc_mode >= 0 ? v_on_uc_mode != 0 : v_split; }
    int spin_limit = 1 << 22; int expect_extra = 0;     // INFV_CHAIN_FAULT=1 (tests): expect one arrival too many -> every wait times out
    int32_t* trace_bins = nullptr; float* trace_probs = nullptr; long trace_cap = 0;   // draw trace of consolidate (caller's device buffers)
    DeviceBuf bins_forced; unsigned forced_mask = 0;    // one-shot forced draw of the per-call path
    DeviceBuf mass_acc[3];             // fixed-point sticky bin masses [L][kAccShards][128] u64, ring of 3 (read / accumulate / being cleared)
    DeviceBuf uf_all;                  // a consolidate call's Gibbs uniforms as fp32 round-ups [n_chunks][L][S] (chain_batch3_kernel's search)
    DeviceBuf step_tab;                // per-call path: resolved gather table of the step [L][N][tabw] (written by the draw plane of step_project)
    DeviceBuf psi_Y, psi_E, psi_Eg, psi_alpha;   // general-psi step: resampled rows, edge scores, grid scores / probabilities, read-out weights
    DeviceBuf R_all, planes_all[3];    // call-long pooling launch: every new row of the call (fp32, and as three bf16 planes per sub-batch for the projection GEMM)
    DeviceBuf pool_done;               // ... and its per-sub-batch completion counts
    DeviceBuf gemm_flags;              // call-long projection GEMM: [0] tile queue head, [64] sub-batches whose UC kernel is done, [128 + b] S' tiles, [128 + cap + b] V' tiles of sub-batch b; the descriptor behind them
    long gemm_flags_cap = 0;
    DeviceBuf call_flags;              // call-long role S: [0] sub-batches projected (GEMM stream -> role S), [64] workgroup x sub-batch completions (role S -> UC stream); words 256 B apart
    DeviceBuf call_stats;              // call-long role S: [0] ticks (100 MHz) workgroup 0 waited for projections, [1] how many sub-batches it waited for
    hipEvent_t ev_chain = nullptr;     // the call-long role-S launch has finished (recorded on its stream)
    std::mutex* issue_mu = nullptr;    // the device's consolidate-issue lock (SharedStreams::issue)
    DeviceBuf mbox;                    // chain_batch3_kernel: mailboxes of role S's exchange + placement handshake (chain_mailbox_bytes)
    int mbox_G = 0;                    // workgroups per layer the mailboxes are laid out for
    int sc = 0;
    int n_bins = 128;
    Profiler prof;
    ~infv_ltm_s() {
        for (auto& kv : plans) delete kv.second;
        // (the streams belong to the process-wide pool, see shared_streams(): synchronise, do not destroy)
        if (side) (void)hipStreamSynchronize(side);
        if (ucs) (void)hipStreamSynchronize(ucs);
        if (pools) (void)hipStreamSynchronize(pools);
        if (chain_s) (void)hipStreamSynchronize(chain_s);
        for (int i = 0; i < kPSets; ++i) if (ev_pool[i]) (void)hipEventDestroy(ev_pool[i]);
        for (int i = 0; i < kRSets; ++i) if (ev_r[i]) (void)hipEventDestroy(ev_r[i]);
        for (int i = 0; i < kPSets; ++i) { if (ev_s[i]) (void)hipEventDestroy(ev_s[i]); if (ev_uc[i]) (void)hipEventDestroy(ev_uc[i]); }
        if (ev_in) (void)hipEventDestroy(ev_in);
        if (ev_start) (void)hipEventDestroy(ev_start);
        if (ev_q) (void)hipEventDestroy(ev_q);
        if (ev_chain) (void)hipEventDestroy(ev_chain);
        for (int i = 0; i < kPSets; ++i) { if (ev_p[i]) (void)hipEventDestroy(ev_p[i]); }
        if (err_host) (void)hipHostFree(err_host);
    }
};

namespace {

int check_handle(infv_ltm_handle h) {
    if (!h) return fail(INFV_ERR_INVALID, "null handle");
    return INFV_OK;
}

// A time-out of the persistent chain kernel leaves garbage in the memory: report it once (at the next entry point
// that would use or extend the memory, or at infv_ltm_sync) and forget the memory.
// header of a chain-state blob (infv_ltm_export_chain_state): 16 int32 in front of the fp32 payload
constexpr int kBlobHeaderInts = 16;
constexpr int kBlobMagic = 0x43464E49;             // "INFC"
constexpr int kBlobVersion = 1;
constexpr unsigned int kErrBlobMismatch = 0x100u;  // error-latch code (1 = chain time-out)
struct BlobHeader { int v[kBlobHeaderInts]; };
__global__ void blob_header_write_kernel(int* out, BlobHeader hd) { if (threadIdx.x < kBlobHeaderInts) out[threadIdx.x] = hd.v[threadIdx.x]; }
__global__ void blob_header_check_kernel(const int* in, BlobHeader hd, unsigned int* err) {
    bool bad = false;
    for (int i = 0; i < kBlobHeaderInts; ++i) bad = bad || in[i] != hd.v[i];
    if (bad) __hip_atomic_store(err, kErrBlobMismatch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

int check_chain_error(infv_ltm_handle h) {
    if (h->err_host && *reinterpret_cast<volatile unsigned int*>(h->err_host)) {
        const unsigned int code = *reinterpret_cast<volatile unsigned int*>(h->err_host);
        *reinterpret_cast<volatile unsigned int*>(h->err_host) = 0;
        h->has_memory = false; h->parts = 0; h->k_stale = false;
        if (code == kErrBlobMismatch)
            return fail(INFV_ERR_STATE, "the blob handed to an earlier infv_ltm_import_chain_state was not exported by a handle of "
                                        "this shape (magic / version / L, N, d, dm, H, Q, n_bins in its header); the memory has been reset");
        return fail(INFV_ERR_STATE, "the persistent chain kernel of an earlier infv_ltm_consolidate timed out waiting for the "
                                    "other workgroups of its layer (not all co-resident: partitioned or shared GPU?); "
                                    "its outputs and the memory are invalid, the memory has been reset");
    }
    return INFV_OK;
}

ProjPtrs make_proj(const infv_ltm_proj* proj, int L) {
    ProjPtrs p;
    memset(&p, 0, sizeof(p));
    for (int l = 0; l < L; ++l) { p.wk[l] = proj[l].wk; p.bk[l] = proj[l].bk; p.wv[l] = proj[l].wv; p.bv[l] = proj[l].bv; }
    return p;
}

int upload_operator(Operator& op, int N, int rows, const int32_t* row_box, const int32_t* row_begin,
                    const int32_t* row_end, const float* box_val, const int32_t* old_ptr,
                    const int32_t* old_slot, int T, int S) {
    if (rows < 0 || rows > N) return fail(INFV_ERR_INVALID, "plan: rows=%d outside [0,%d]", rows, N);
    std::vector<int32_t> box_row(N, -1);
    for (int r = 0; r < rows; ++r) {
        if (row_box[r] < 0 || row_box[r] >= N || box_row[row_box[r]] != -1)
            return fail(INFV_ERR_INVALID, "plan: bad or duplicate row_box[%d]=%d", r, row_box[r]);
        if (row_begin[r] < 0 || row_end[r] > T || row_begin[r] >= row_end[r])
            return fail(INFV_ERR_INVALID, "plan: bad frame range of row %d", r);
        box_row[row_box[r]] = r;
    }
    op.rows = rows;
    HIP_TRY(upload(op.row_box, row_box, rows));
    HIP_TRY(upload(op.row_begin, row_begin, rows));
    HIP_TRY(upload(op.row_end, row_end, rows));
    HIP_TRY(upload(op.box_val, box_val, N));
    HIP_TRY(upload(op.box_row, box_row.data(), N));
    op.has_old = old_ptr != nullptr;
    if (old_ptr) {
        const int nnz = old_ptr[N];
        if (old_ptr[0] != 0 || nnz < 0 || nnz > S) return fail(INFV_ERR_INVALID, "plan: bad old_ptr");
        for (int i = 0; i < nnz; ++i)
            if (old_slot[i] < 0 || old_slot[i] >= S) return fail(INFV_ERR_INVALID, "plan: old_slot out of range");
        int mx = 0;
        for (int n = 0; n < N; ++n) {
            if (old_ptr[n + 1] < old_ptr[n]) return fail(INFV_ERR_INVALID, "plan: old_ptr not monotone");
            if (old_ptr[n + 1] - old_ptr[n] > mx) mx = old_ptr[n + 1] - old_ptr[n];
        }
        op.tabw = mx < 4 ? 4 : (mx + 3) & ~3;
        std::vector<int32_t> tab((size_t)N * op.tabw, -1);
        for (int n = 0; n < N; ++n)
            for (int k = 0; k < old_ptr[n + 1] - old_ptr[n]; ++k) tab[(size_t)n * op.tabw + k] = old_slot[old_ptr[n] + k];
        HIP_TRY(upload(op.slot_tab, tab.data(), tab.size()));
        HIP_TRY(upload(op.old_ptr, old_ptr, N + 1));
        HIP_TRY(upload(op.old_slot, old_slot, nnz));
    }
    return INFV_OK;
}

// draw (if the memory exists) -> update -> attend, for all layers, on one chunk's new rows.
// kbar_rows != nullptr (single-chunk step): R is null and the update builds the new rows of B from the pooled frames;
// draw_done: the draw already ran inside the projection launch (launch_step_project).
int chain_step(infv_ltm_handle h, const Plan& plan, const float* R, const float* Pnew, int splitk,
               long split_stride, const float* q, int Q, const ProjPtrs& pp, const double* u, float* ctx,
               hipStream_t stream, const float* kbar_rows = nullptr, bool draw_done = false, const int32_t* tab = nullptr) {
    const bool inf = h->has_memory;
    const Operator& op = inf ? plan.inf : plan.first;
    const int32_t* idx = nullptr;
    int idx_stride = 0;
    if (inf) {
        if (h->cfg.sticky) {
            if (!draw_done) {
                if (!u) return fail(INFV_ERR_INVALID, "sticky step on an existing memory needs the Gibbs uniforms u");
                if (h->parts <= 0) return fail(INFV_ERR_STATE, "no sticky histogram available (import_state or step first)");
                Timed t_(h->prof, INFV_KERNEL_DRAW, stream);
                HIP_TRY(launch_draw(h->bin_part[h->pc].as<float>(), h->parts, h->probs_override.as<float>(),
                                    h->override_mask, plan.sticky(), u, h->S, h->L,
                                    h->probs.as<float>(), h->bins.as<int32_t>(), h->idx.as<int32_t>(), stream,
                                    h->bins_forced.as<int32_t>(), h->forced_mask));
            }
            h->override_mask = 0;
            h->forced_mask = 0;
            idx = h->idx.as<int32_t>();
            idx_stride = h->S;
        } else {
            idx = plan.uniform_idx.as<int32_t>();
            idx_stride = 0;
        }
    }
    const int nxt = h->cur ^ 1;
    {
    Timed t_(h->prof, INFV_KERNEL_UPDATE, stream);
    HIP_TRY(launch_update(op.view(), h->N, h->d, h->dm, h->L, h->S, idx, idx_stride, R, Pnew, splitk, split_stride,
                          h->B[h->cur].as<float>(), h->KV[h->cur].as<float>(), h->B[nxt].as<float>(),
                          h->KV[nxt].as<float>(), stream, kbar_rows, (inf && h->cfg.sticky && draw_done) ? tab : nullptr));
    }
    h->cur = nxt;
    h->has_memory = true;
    const int parts = attend_parts(Q, h->H, h->N);
    {
    Timed t_(h->prof, INFV_KERNEL_ATTEND, stream);
    HIP_TRY(launch_attend(q, Q, h->N, h->H, h->L, h->KV[h->cur].as<float>(), pp, plan.w.as<float>(), plan.w_out,
                          plan.sticky(), ctx, h->bin_part[h->pc].as<float>(), h->scores.as<float>(), stream));
    }
    h->parts = parts;
    h->lastQ = Q;
    h->last_fast = false;
    return INFV_OK;
}

// One step with dense operators (ltm_dense.hip): draw -> B = x . G -> project the whole memory -> scores, softmax weights,
// read-out -> sticky masses with two-box edges.  Same state layout as chain_step; the K'/V' rows are re-projected from B.
int dense_step(infv_ltm_handle h, const Plan& plan, const float* kbar, int T, const float* q, int Q, const ProjPtrs& pp,
               const double* u, float* ctx, hipStream_t stream) {
    const bool inf = h->has_memory;
    const DensePlan& dp = plan.dense;
    const int32_t* bins = nullptr;
    int bins_stride = 0;
    const int32_t* pos_box2 = dp.uniform_box2.as<int32_t>();
    if (inf && h->cfg.sticky) {
        if (!u) return fail(INFV_ERR_INVALID, "sticky step on an existing memory needs the Gibbs uniforms u");
        if (h->parts <= 0) return fail(INFV_ERR_STATE, "no sticky histogram available (import_state or step first)");
        // (checked BEFORE the draw consumes the one-shot masks: an unsupported request leaves the handle as it was)
        if (h->forced_mask && h->forced_mask != (1u << h->L) - 1u)
            return fail(INFV_ERR_UNSUPPORTED, "dense plans: infv_ltm_set_bins must force every layer of the handle or none");
        Timed t_(h->prof, INFV_KERNEL_DRAW, stream);
        HIP_TRY(launch_draw(h->bin_part[h->pc].as<float>(), h->parts, h->probs_override.as<float>(), h->override_mask,
                            plan.sticky(), u, h->S, h->L, h->probs.as<float>(), h->bins.as<int32_t>(), h->idx.as<int32_t>(),
                            stream, h->bins_forced.as<int32_t>(), h->forced_mask));
        h->override_mask = 0;
        // a forced draw replaces the resampled bins (infv_ltm_set_bins); get_draw still returns the step's own
        bins = h->forced_mask ? h->bins_forced.as<int32_t>() : h->bins.as<int32_t>();
        h->forced_mask = 0;
        bins_stride = h->S;
        pos_box2 = dp.bin_box2.as<int32_t>();
    }
    const int nxt = h->cur ^ 1;
    if (dp.psi_on) {
        // psi(t) is a dense row: the resampled rows are rows of Y = Psi_pos . B_past (128 positions when sticky, S otherwise)
        Timed t_(h->prof, INFV_KERNEL_UPDATE, stream);
        const bool use_bins = bins != nullptr;
        const int n_pos = use_bins ? h->n_bins : h->S;
        if (inf) {
            HIP_TRY(h->psi_Y.reserve((size_t)h->L * n_pos * h->d * sizeof(float)));
            HIP_TRY(launch_psi_gemm(false, use_bins ? dp.psi_bin.as<float>() : dp.psi_uniform.as<float>(), h->N, 0,
                                    h->B[h->cur].as<float>(), h->d, (long)h->N * h->d, h->psi_Y.as<float>(), h->d, (long)n_pos * h->d,
                                    n_pos, h->d, h->N, h->L, stream));
        }
        HIP_TRY(launch_psi_update(inf ? dp.inf_GT.as<float>() : dp.first_GT.as<float>(), inf ? dp.inf_K : dp.first_K,
                                  inf ? dp.inf_K : dp.first_K, inf ? h->S : 0, bins, bins_stride, h->psi_Y.as<float>(), n_pos, kbar,
                                  h->B[nxt].as<float>(), h->N, h->d, h->L, stream));
    } else {
        Timed t_(h->prof, INFV_KERNEL_UPDATE, stream);
        HIP_TRY(launch_dense_update(inf ? dp.inf_GT.as<float>() : dp.first_GT.as<float>(), inf ? dp.inf_K : dp.first_K,
                                    inf ? dp.inf_K : dp.first_K, inf ? h->S : 0, bins, bins_stride, pos_box2,
                                    h->B[h->cur].as<float>(), kbar, h->B[nxt].as<float>(), h->N, h->d, h->L, stream));
    }
    h->cur = nxt;
    h->has_memory = true;
    HIP_TRY(launch_reproject(h->B[h->cur].as<float>(), h->N, h->d, h->dm, h->L, pp, h->KV[h->cur].as<float>(), stream));
    h->k_stale = false;
    {
        Timed t_(h->prof, INFV_KERNEL_ATTEND, stream);
        HIP_TRY(launch_attend(q, Q, h->N, h->H, h->L, h->KV[h->cur].as<float>(), pp, plan.w.as<float>(), plan.w_out,
                              plan.sticky(), ctx, h->bin_part[h->pc].as<float>(), h->scores.as<float>(), stream));
        if (dp.psi_on) {
            // the attend kernel above supplied the scores S = q K^T / sqrt(dh); its read-out and masses assume boxes.  Dense forms:
            const long rows_hq = (long)h->L * h->H * Q;
            const int ldE = h->n_bins + 4, ldG = (dp.n_grid + 3) & ~3;
            HIP_TRY(h->psi_E.reserve((size_t)rows_hq * ldE * sizeof(float)));
            HIP_TRY(h->psi_Eg.reserve((size_t)rows_hq * ldG * sizeof(float)));
            HIP_TRY(h->psi_alpha.reserve((size_t)rows_hq * h->N * sizeof(float)));
            // edge scores z(t_j) = sum_n S[n] psi_n(t_j) -> sticky masses (LTM.py:197-203,224-230)
            HIP_TRY(launch_psi_gemm(true, h->scores.as<float>(), h->N, 0, dp.psi_edge.as<float>(), h->N, 0, h->psi_E.as<float>(), ldE, 0,
                                    (int)rows_hq, h->n_bins + 1, h->N, 1, stream));
            HIP_TRY(launch_psi_masses(h->psi_E.as<float>(), ldE, Q, h->H, h->L, plan.edge_dx.as<float>(), h->bin_part[h->pc].as<float>(), stream));
            // read-out on the 1000-point grid (LTM.py:251-286): z -> w o prob -> alpha = (w o prob) . Psi_grid -> ctx = alpha . (V' + bv)
            HIP_TRY(launch_psi_gemm(true, h->scores.as<float>(), h->N, 0, dp.psi_grid.as<float>(), h->N, 0, h->psi_Eg.as<float>(), ldG, 0,
                                    (int)rows_hq, dp.n_grid, h->N, 1, stream));
            HIP_TRY(launch_psi_grid(h->psi_Eg.as<float>(), ldG, dp.n_grid, rows_hq, dp.grid_w.as<float>(), stream));
            HIP_TRY(launch_psi_gemm(false, h->psi_Eg.as<float>(), ldG, 0, dp.psi_grid.as<float>(), h->N, 0, h->psi_alpha.as<float>(), h->N, 0,
                                    (int)rows_hq, h->N, dp.n_grid, 1, stream));
            HIP_TRY(launch_psi_ctx(h->psi_alpha.as<float>(), h->KV[h->cur].as<float>(), pp, Q, h->N, h->H, h->dm / h->H, h->L, ctx, stream));
        } else {
        // the attend kernel's partials assume one box per edge: recompute them from the scores with the two-box table
        HIP_TRY(launch_dense_masses(h->scores.as<float>(), Q, h->N, h->H, h->L, dp.edge_box2.as<int32_t>(),
                                    plan.edge_dx.as<float>(), h->bin_part[h->pc].as<float>(), stream));
        }
    }
    h->parts = h->H;
    h->lastQ = Q;
    h->last_fast = false;
    return INFV_OK;
}

int find_plan(infv_ltm_handle h, int T, Plan** out) {
    auto it = h->plans.find(T);
    if (it == h->plans.end()) return fail(INFV_ERR_NO_PLAN, "no plan registered for chunk length T=%d", T);
    *out = it->second;
    return INFV_OK;
}

int check_q(infv_ltm_handle h, int Q) {
    if (Q <= 0 || Q > h->maxQ) return fail(INFV_ERR_INVALID, "Q=%d outside (0, max_q=%d]", Q, h->maxQ);
    return INFV_OK;
}

// project the new rows of `n_chunks` pooled chunks into workspace set `set`
int project_chunks(infv_ltm_handle h, const Plan& plan, bool inf, const float* kbar, int n_chunks, int T,
                   const ProjPtrs& pp, int set, int* splitk, long* split_stride, hipStream_t stream, int gemm_pad = 0) {
    const Operator& op = inf ? plan.inf : plan.first;
    const long M = (long)n_chunks * op.rows;
    const long n_cols = (long)h->L * 2 * h->dm;
    const int sk = project_splitk((int)M, h->d);
    DeviceBuf& Rw = h->R_ws[set];
    DeviceBuf& Pw = h->P_ws[set];
    const size_t needR = (size_t)(M ? M : 1) * h->d * sizeof(float), needP = (size_t)(M ? M : 1) * n_cols * sk * sizeof(float);
    if (needR > Rw.bytes || needP > Pw.bytes) HIP_TRY(hipDeviceSynchronize());   // growing frees: nothing may still read it
    HIP_TRY(Rw.reserve(needR));
    HIP_TRY(Pw.reserve(needP));
    {
    Timed t_(h->prof, INFV_KERNEL_ROWS, stream);
    HIP_TRY(launch_rows(kbar, n_chunks, T, h->d, op.view(), Rw.as<float>(), stream));
    }
    {
    Timed t_(h->prof, INFV_KERNEL_PROJECT, stream);
    HIP_TRY(launch_project(n_chunks, h->d, h->dm, h->L, op.view(), pp, Rw.as<float>(), Pw.as<float>(), stream, gemm_pad));
    }
    *splitk = sk;
    *split_stride = M * n_cols;
    return INFV_OK;
}

}  // namespace

extern "C" {

int infv_ltm_abi_version(void) { return INFV_LTM_ABI_VERSION; }
int64_t infv_ltm_launch_count(void) { return (int64_t)::infv::g_kernel_launches.load(std::memory_order_relaxed); }
const char* infv_ltm_last_error(void) { return g_err; }

int infv_ltm_create(const infv_ltm_config* cfg, infv_ltm_handle* out) {
    if (!cfg || !out) return fail(INFV_ERR_INVALID, "null argument");
    *out = nullptr;
    if (cfg->head_size != kHeadSize) return fail(INFV_ERR_UNSUPPORTED, "head_size=%d (kernels are built for %d)", cfg->head_size, kHeadSize);
    if (cfg->num_basis <= 0 || cfg->num_basis % 16) return fail(INFV_ERR_UNSUPPORTED, "num_basis=%d must be a positive multiple of 16", cfg->num_basis);
    if (cfg->n_layers < 1 || cfg->n_layers > INFV_LTM_MAX_LAYERS) return fail(INFV_ERR_INVALID, "n_layers=%d outside [1,%d]", cfg->n_layers, INFV_LTM_MAX_LAYERS);
    if (cfg->d_in <= 0 || cfg->d_in % 32) return fail(INFV_ERR_UNSUPPORTED, "d_in=%d must be a positive multiple of 32", cfg->d_in);
    if (cfg->n_heads <= 0 || (cfg->n_heads * cfg->head_size) % 64) return fail(INFV_ERR_UNSUPPORTED, "n_heads*head_size must be a multiple of 64");
    if (cfg->tokens_per_frame <= 0 || cfg->nb_samples <= 0 || cfg->max_q <= 0) return fail(INFV_ERR_INVALID, "tokens_per_frame, nb_samples, max_q must be positive");
    infv_ltm_s* h = new (std::nothrow) infv_ltm_s();
    if (!h) return fail(INFV_ERR_NOMEM, "out of host memory");
    h->cfg = *cfg;
    h->N = cfg->num_basis; h->H = cfg->n_heads; h->dh = cfg->head_size; h->d = cfg->d_in;
    h->dm = cfg->n_heads * cfg->head_size; h->P = cfg->tokens_per_frame; h->L = cfg->n_layers;
    h->S = cfg->nb_samples; h->maxQ = cfg->max_q; h->maxC = cfg->max_batch_chunks > 0 ? cfg->max_batch_chunks : 32;
    const size_t nb = (size_t)h->L * h->N;
    const int max_parts = h->H * ((h->maxQ + 3) / 4);      // finest partial granularity (4-row tiles of the per-call attend kernel)
    hipError_t e = hipSuccess;
    for (int i = 0; i < 2 && e == hipSuccess; ++i) {
        e = h->B[i].reserve(nb * h->d * sizeof(float));
        if (e == hipSuccess) e = h->KV[i].reserve(nb * 2 * h->dm * sizeof(float));
    }
    const size_t nsq = (size_t)h->L * h->H * h->maxQ * h->N;
    for (int i = 0; i < 2 && e == hipSuccess; ++i) {
        e = h->bin_part[i].reserve((size_t)h->L * max_parts * h->n_bins * sizeof(float));
        if (e == hipSuccess) e = hipMemset(h->bin_part[i].p, 0, h->bin_part[i].bytes);
        if (e == hipSuccess) e = h->Sp[i].reserve(nsq * sizeof(float));
    }
    for (int i = 0; i < 3 && e == hipSuccess; ++i) {
        e = h->mass_acc[i].reserve((size_t)h->L * kAccShards * 128 * sizeof(unsigned long long));
        if (e == hipSuccess) e = hipMemset(h->mass_acc[i].p, 0, h->mass_acc[i].bytes);
    }
    if (e == hipSuccess) e = h->sync_words.reserve(16 * sizeof(unsigned int));
    if (e == hipSuccess) e = hipMemset(h->sync_words.p, 0, 16 * sizeof(unsigned int));
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&h->err_host), 64, hipHostMallocMapped);
    if (e == hipSuccess) { memset(h->err_host, 0, 64); e = hipHostGetDevicePointer(reinterpret_cast<void**>(&h->err_dev), h->err_host, 0); }
    if (e == hipSuccess) e = h->bins_forced.reserve((size_t)h->L * h->S * sizeof(int32_t));
    if (const char* f = exp_env("INFV_CHAIN_FAULT")) if (atoi(f) != 0) { h->expect_extra = 1; h->spin_limit = 1 << 12; }
    if (const char* f = getenv("INFV_VPROJ_SPLIT")) h->v_split = atoi(f) != 0;
    if (const char* f = getenv("INFV_PROJ_X6")) h->proj_x6 = atoi(f) != 0;
    if (const char* f = exp_env("INFV_VPROJ_ON_UC")) h->v_on_uc_mode = atoi(f) != 0 ? 1 : 0;
    h->ring = kPSets * h->maxC + 2;         // a slot is rewritten kPSets sub-batches after the one whose UC kernel read it
    if (e == hipSuccess) e = h->cqbuf.reserve((size_t)h->L * h->H * h->maxQ * sizeof(float));
    if (e == hipSuccess) e = h->qt_buf.reserve((size_t)h->L * h->H * h->maxQ * h->d * sizeof(float));
    if (e == hipSuccess) e = h->probs.reserve((size_t)h->L * h->n_bins * sizeof(float));
    if (e == hipSuccess) e = h->probs_override.reserve((size_t)h->L * h->n_bins * sizeof(float));
    if (e == hipSuccess) e = h->bins.reserve((size_t)h->L * h->S * sizeof(int32_t));
    if (e == hipSuccess) e = h->idx.reserve((size_t)h->L * h->S * sizeof(int32_t));
    if (e == hipSuccess) e = h->scores.reserve((size_t)h->L * h->H * h->maxQ * h->N * sizeof(float));
    // what infv_ltm_get_draw returns before any draw has happened (non-sticky handles never draw) is defined: zeros
    if (e == hipSuccess) e = hipMemset(h->probs.p, 0, h->probs.bytes);
    if (e == hipSuccess) e = hipMemset(h->bins.p, 0, h->bins.bytes);
    if (e == hipSuccess) e = hipMemset(h->idx.p, 0, h->idx.bytes);
    if (e != hipSuccess) {
        delete h;
        return fail(INFV_ERR_HIP, "device allocation failed: %s", hipGetErrorString(e));
    }
    *out = h;
    return INFV_OK;
}

int infv_ltm_destroy(infv_ltm_handle h) {
    delete h;
    return INFV_OK;
}

int infv_ltm_set_plan(infv_ltm_handle h, const infv_ltm_plan* p) {
    if (int rc = check_handle(h)) return rc;
    if (!p || p->T < 2) return fail(INFV_ERR_INVALID, "plan: T must be >= 2 (the reference operator is empty for T=1)");
    if (p->n_bins != h->n_bins) return fail(INFV_ERR_UNSUPPORTED, "plan: n_bins=%d (kernels are built for %d)", p->n_bins, h->n_bins);
    Plan* plan = new (std::nothrow) Plan();
    if (!plan) return fail(INFV_ERR_NOMEM, "out of host memory");
    plan->T = p->T;
    int rc = upload_operator(plan->first, h->N, p->first_rows, p->first_row_box, p->first_row_begin,
                             p->first_row_end, p->first_box_val, nullptr, nullptr, p->T, h->S);
    if (rc == INFV_OK)
        rc = upload_operator(plan->inf, h->N, p->inf_rows, p->inf_row_box, p->inf_row_begin, p->inf_row_end,
                             p->inf_box_val, p->inf_old_ptr, p->inf_old_slot, p->T, h->S);
    if (rc != INFV_OK) { delete plan; return rc; }
    for (int j = 0; j <= p->n_bins; ++j)
        if (p->edge_box[j] < -1 || p->edge_box[j] >= h->N) { delete plan; return fail(INFV_ERR_INVALID, "plan: edge_box out of range"); }
    for (int j = 0; j < p->n_bins; ++j)
        if (p->bin_box[j] < -1 || p->bin_box[j] >= h->N) { delete plan; return fail(INFV_ERR_INVALID, "plan: bin_box out of range"); }
    for (int s = 0; s < h->S; ++s)
        if (p->uniform_idx[s] < -1 || p->uniform_idx[s] >= h->N) { delete plan; return fail(INFV_ERR_INVALID, "plan: uniform_idx out of range"); }
    plan->w_out = p->readout_w_out;
    plan->n_bins = p->n_bins;
    plan->points_ok = p->edge_box[0] == -1 && p->edge_box[p->n_bins] == -1;
    for (int j = 0; j < p->n_bins; ++j) {
        if (p->bin_box[j] < 0) plan->points_ok = false;
        if (j > 0 && p->edge_box[j] != p->bin_box[j]) plan->points_ok = false;
    }
    hipError_t e = upload(plan->w, p->readout_w, h->N);
    if (e == hipSuccess) e = upload(plan->edge_box, p->edge_box, p->n_bins + 1);
    if (e == hipSuccess) e = upload(plan->edge_dx, p->edge_dx, p->n_bins);
    if (e == hipSuccess) e = upload(plan->bin_box, p->bin_box, p->n_bins);
    if (e == hipSuccess) e = upload(plan->uniform_idx, p->uniform_idx, h->S);
    if (e != hipSuccess) { delete plan; return fail(INFV_ERR_HIP, "plan upload failed: %s", hipGetErrorString(e)); }
    auto it = h->plans.find(p->T);
    if (it != h->plans.end()) { delete it->second; h->plans.erase(it); }
    h->plans[p->T] = plan;
    return INFV_OK;
}

int infv_ltm_has_plan(infv_ltm_handle h, int32_t T) {
    if (int rc = check_handle(h)) return rc;
    return h->plans.count(T) ? 1 : 0;
}

int infv_ltm_set_dense_plan(infv_ltm_handle h, const infv_ltm_dense_plan* p) {
    if (int rc = check_handle(h)) return rc;
    if (!p || !p->first_GT || !p->inf_GT || !p->bin_box2 || !p->edge_box2 || !p->uniform_box2)
        return fail(INFV_ERR_INVALID, "dense plan: null argument");
    Plan* plan = nullptr;
    if (int rc = find_plan(h, p->T, &plan)) return rc;          // the regular plan of this T carries readout_w / edge_dx
    if (p->first_K != p->T || p->inf_K != h->S + p->T)
        return fail(INFV_ERR_INVALID, "dense plan: first_K=%d inf_K=%d, expected %d and %d", p->first_K, p->inf_K, p->T, h->S + p->T);
    auto check_pairs = [&](const int32_t* t, int n) {
        for (int i = 0; i < 2 * n; ++i) if (t[i] < -1 || t[i] >= h->N) return false;
        return true;
    };
    if (!check_pairs(p->bin_box2, h->n_bins) || !check_pairs(p->edge_box2, h->n_bins + 1) || !check_pairs(p->uniform_box2, h->S))
        return fail(INFV_ERR_INVALID, "dense plan: box index out of range");
    DensePlan& d = plan->dense;
    HIP_TRY(upload(d.first_GT, p->first_GT, (size_t)h->N * p->first_K));
    HIP_TRY(upload(d.inf_GT, p->inf_GT, (size_t)h->N * p->inf_K));
    HIP_TRY(upload(d.bin_box2, p->bin_box2, (size_t)2 * h->n_bins));
    HIP_TRY(upload(d.edge_box2, p->edge_box2, (size_t)2 * (h->n_bins + 1)));
    HIP_TRY(upload(d.uniform_box2, p->uniform_box2, (size_t)2 * h->S));
    d.first_K = p->first_K; d.inf_K = p->inf_K;
    d.on = true;
    return INFV_OK;
}

int infv_ltm_set_psi_plan(infv_ltm_handle h, const infv_ltm_psi_plan* p) {
    if (int rc = check_handle(h)) return rc;
    if (!p || !p->psi_edge || !p->psi_bin || !p->psi_uniform || !p->psi_grid || !p->grid_w || p->n_grid < 2)
        return fail(INFV_ERR_INVALID, "psi plan: null argument");
    Plan* plan = nullptr;
    if (int rc = find_plan(h, p->T, &plan)) return rc;
    DensePlan& d = plan->dense;
    if (!d.on) return fail(INFV_ERR_STATE, "psi plan: set the dense plan of T=%d first (infv_ltm_set_dense_plan)", p->T);
    HIP_TRY(hipDeviceSynchronize());                            // a step of this length may still be reading the old tables
    HIP_TRY(upload(d.psi_edge, p->psi_edge, (size_t)(h->n_bins + 1) * h->N));
    HIP_TRY(upload(d.psi_bin, p->psi_bin, (size_t)h->n_bins * h->N));
    HIP_TRY(upload(d.psi_uniform, p->psi_uniform, (size_t)h->S * h->N));
    HIP_TRY(upload(d.psi_grid, p->psi_grid, (size_t)p->n_grid * h->N));
    HIP_TRY(upload(d.grid_w, p->grid_w, (size_t)p->n_grid));
    d.n_grid = p->n_grid;
    d.psi_on = true;
    return INFV_OK;
}

int infv_ltm_reset(infv_ltm_handle h) {
    if (int rc = check_handle(h)) return rc;
    h->has_memory = false;
    h->parts = 0;
    h->k_stale = false;
    return INFV_OK;
}

int infv_ltm_has_memory(infv_ltm_handle h) {
    if (int rc = check_handle(h)) return rc;
    return h->has_memory ? 1 : 0;
}

int infv_ltm_set_token_dtype(infv_ltm_handle h, int32_t dtype) {
    if (int rc = check_handle(h)) return rc;
    if (dtype != INFV_TOKENS_F32 && dtype != INFV_TOKENS_BF16) return fail(INFV_ERR_INVALID, "set_token_dtype: unknown dtype %d", dtype);
    h->k_bf16 = dtype == INFV_TOKENS_BF16;
    return INFV_OK;
}

int infv_ltm_pool(infv_ltm_handle h, const void* k, int64_t n_frames, float* kbar, void* stream) {
    if (int rc = check_handle(h)) return rc;
    if (!k || !kbar || n_frames < 0) return fail(INFV_ERR_INVALID, "pool: bad arguments");
    Timed t_(h->prof, INFV_KERNEL_POOL, static_cast<hipStream_t>(stream));
    HIP_TRY(launch_pool(k, h->k_bf16, kbar, n_frames, h->P, h->d, static_cast<hipStream_t>(stream)));
    return INFV_OK;
}

int infv_ltm_new_rows(infv_ltm_handle h, int32_t T) {
    if (int rc = check_handle(h)) return rc;
    Plan* plan = nullptr;
    if (int rc = find_plan(h, T, &plan)) return rc;
    if (plan->dense.on) return fail(INFV_ERR_UNSUPPORTED, "new_rows: the plan for T = %d is dense (no box rows)", T);
    return plan->inf.rows;
}

int infv_ltm_pool_rows(infv_ltm_handle h, const void* k, int32_t n_chunks, int32_t T, float* R, void* stream_) {
    if (int rc = check_handle(h)) return rc;
    if (!k || !R || n_chunks < 0) return fail(INFV_ERR_INVALID, "pool_rows: bad arguments");
    Plan* plan = nullptr;
    if (int rc = find_plan(h, T, &plan)) return rc;
    if (plan->dense.on) return fail(INFV_ERR_UNSUPPORTED, "pool_rows: the plan for T = %d is dense (no box rows)", T);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    Timed t_(h->prof, INFV_KERNEL_POOL, stream);
    if (pool_rows2_supported(h->d)) {
        static const int wgs = [] { const char* e = exp_env("INFV_PR_WGS"); return e ? atoi(e) : 0; }();      // (tools/pool_cus.py)
        static const int pad = [] { const char* e = exp_env("INFV_PR_PAD"); return e ? atoi(e) : 84 * 1024; }();
        HIP_TRY(launch_pool_rows2(k, h->k_bf16, n_chunks, T, h->P, h->d, plan->inf.view(), R, stream, 8, pad, wgs));
        return INFV_OK;
    }
    // widths without a pool_rows2 shape: the two kernels, one chunk group at a time through the pooled-frame workspace
    const size_t need = (size_t)n_chunks * T * h->d * sizeof(float);
    if (need > h->kbar_side[0].bytes) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(h->kbar_side[0].reserve(need)); }
    HIP_TRY(launch_pool(k, h->k_bf16, h->kbar_side[0].as<float>(), (int64_t)n_chunks * T, h->P, h->d, stream));
    HIP_TRY(launch_rows(h->kbar_side[0].as<float>(), n_chunks, T, h->d, plan->inf.view(), R, stream));
    return INFV_OK;
}

int infv_ltm_step(infv_ltm_handle h, const float* kbar, int32_t T, const float* q, int32_t Q,
                  const infv_ltm_proj* proj, const double* u, float* ctx, void* stream_) {
    if (int rc = check_handle(h)) return rc;
    if (!kbar || !q || !proj || !ctx) return fail(INFV_ERR_INVALID, "step: null argument");
    if (int rc = check_chain_error(h)) return rc;
    if (int rc = check_q(h, Q)) return rc;
    Plan* plan = nullptr;
    if (int rc = find_plan(h, T, &plan)) return rc;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const ProjPtrs pp = make_proj(proj, h->L);
    if (plan->dense.on) return dense_step(h, *plan, kbar, T, q, Q, pp, u, ctx, stream);
    if (h->k_stale && h->has_memory)
        if (int rc = infv_ltm_reproject(h, proj, stream_)) return rc;
    // Three launches per step (six in round 2): [new-row projection with the rows built on the fly + the draw] -> update
    // (its new B rows built from the pooled frames too) -> attend.  Same arithmetic in the same order as the batched path.
    const bool inf = h->has_memory;
    const Operator& op = inf ? plan->inf : plan->first;
    const long n_cols = (long)h->L * 2 * h->dm;
    const int sk_max = project_splitk(op.rows > 0 ? op.rows : 1, h->d);
    const size_t needP = (size_t)(op.rows ? op.rows : 1) * n_cols * sk_max * sizeof(float);
    if (needP > h->P_ws[0].bytes) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(h->P_ws[0].reserve(needP)); }
    StepDraw dr;
    memset(&dr, 0, sizeof(dr));
    if (inf && h->cfg.sticky) {
        if (!u) return fail(INFV_ERR_INVALID, "sticky step on an existing memory needs the Gibbs uniforms u");
        if (h->parts <= 0) return fail(INFV_ERR_STATE, "no sticky histogram available (import_state or step first)");
        dr.n_layers = h->L; dr.bin_part = h->bin_part[h->pc].as<float>(); dr.parts = h->parts;
        dr.probs_override = h->probs_override.as<float>(); dr.override_mask = h->override_mask; dr.sticky = plan->sticky();
        dr.u = u; dr.S = h->S; dr.probs_out = h->probs.as<float>(); dr.bins_out = h->bins.as<int32_t>(); dr.idx_out = h->idx.as<int32_t>();
        dr.bins_forced = h->bins_forced.as<int32_t>(); dr.forced_mask = h->forced_mask;
        // the draw also resolves the gather table of the step (per layer: source box of every (box, slot) entry)
        const size_t need_t = (size_t)h->L * h->N * op.tabw * sizeof(int32_t);
        if (op.tabw > 0 && op.slot_tab.p != nullptr) {
            if (need_t > h->step_tab.bytes) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(h->step_tab.reserve(need_t)); }
            dr.slot_tab = op.slot_tab.as<int32_t>(); dr.tab_entries = h->N * op.tabw; dr.tab_out = h->step_tab.as<int32_t>();
        }
    }
    int sk = 1;
    {
        Timed t_(h->prof, INFV_KERNEL_PROJECT, stream);
        HIP_TRY(launch_step_project(kbar, h->d, h->dm, h->L, op.view(), pp, h->P_ws[0].as<float>(), &sk, dr, stream));
    }
    return chain_step(h, *plan, nullptr, h->P_ws[0].as<float>(), sk, (long)op.rows * n_cols, q, Q, pp, u, ctx, stream, kbar,
                      dr.n_layers > 0, dr.tab_out);
}

int infv_ltm_steps(infv_ltm_handle h, const float* kbar, int32_t n_chunks, int32_t T, const float* q, int32_t Q,
                   const infv_ltm_proj* proj, const double* u, float* ctx, void* stream_) {
    if (int rc = check_handle(h)) return rc;
    if (!kbar || !q || !proj || !ctx || n_chunks < 0) return fail(INFV_ERR_INVALID, "steps: bad arguments");
    if (int rc = check_chain_error(h)) return rc;
    if (int rc = check_q(h, Q)) return rc;
    Plan* plan = nullptr;
    if (int rc = find_plan(h, T, &plan)) return rc;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const ProjPtrs pp = make_proj(proj, h->L);
    const size_t chunk_kbar = (size_t)T * h->d, chunk_q = (size_t)h->L * Q * h->dm, chunk_u = (size_t)h->L * h->S;
    int c = 0;
    if (plan->dense.on) {                                     // dense operators: chunk by chunk
        for (; c < n_chunks; ++c)
            if (int rc = infv_ltm_step(h, kbar + c * chunk_kbar, T, q + c * chunk_q, Q, proj, u ? u + c * chunk_u : nullptr,
                                       ctx + c * chunk_q, stream_)) return rc;
        return INFV_OK;
    }
    if (!h->has_memory && n_chunks > 0) {                     // first chunk of a document: its own operator
        if (int rc = infv_ltm_step(h, kbar, T, q, Q, proj, u, ctx, stream_)) return rc;
        c = 1;
    }
    if (c < n_chunks && h->k_stale)
        if (int rc = infv_ltm_reproject(h, proj, stream_)) return rc;
    const size_t rows = plan->inf.rows;
    const long n_cols = (long)h->L * 2 * h->dm;
    // sub-batches of at least 16 chunks: 1024 new rows take the large-tile GEMM without split-K
    const int sub = h->maxC > 16 ? h->maxC : 16;
    while (c < n_chunks) {
        const int nb = n_chunks - c < sub ? n_chunks - c : sub;
        int sk = 1; long ss = 0;
        // workspace set 0 is reused by every sub-batch: its readers (the update kernels of the previous one) are earlier on
        // this same stream
        if (int rc = project_chunks(h, *plan, true, kbar + c * chunk_kbar, nb, T, pp, 0, &sk, &ss, stream)) return rc;
        for (int i = 0; i < nb; ++i)
            if (int rc = chain_step(h, *plan, h->R_ws[0].as<float>() + (size_t)i * rows * h->d,
                                    h->P_ws[0].as<float>() + (size_t)i * rows * n_cols, sk, ss, q + (c + i) * chunk_q, Q, pp,
                                    u ? u + (c + i) * chunk_u : nullptr, ctx + (c + i) * chunk_q, stream)) return rc;
        c += nb;
    }
    return INFV_OK;
}

int infv_ltm_consolidate_q(infv_ltm_handle h, const void* k, int32_t n_chunks, int32_t T, const float* q,
                           int32_t Q, const infv_ltm_proj* proj, const double* u, int32_t new_doc, float* ctx,
                           void* stream) {
    if (int rc = check_handle(h)) return rc;
    if (!k || n_chunks < 0) return fail(INFV_ERR_INVALID, "consolidate_q: bad arguments");
    if (new_doc) infv_ltm_reset(h);
    if (n_chunks == 0) return INFV_OK;
    const size_t need = (size_t)n_chunks * T * h->d * sizeof(float);
    if (need > h->kbar_all.bytes) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(h->kbar_all.reserve(need)); }
    if (int rc = infv_ltm_pool(h, k, (int64_t)n_chunks * T, h->kbar_all.as<float>(), stream)) return rc;
    return infv_ltm_steps(h, h->kbar_all.as<float>(), n_chunks, T, q, Q, proj, u, ctx, stream);
}

int infv_ltm_forward(infv_ltm_handle h, const void* k, int32_t T, const float* q, int32_t Q,
                     const infv_ltm_proj* proj, const double* u, int32_t new_doc, float* ctx, void* stream) {
    if (int rc = check_handle(h)) return rc;
    if (!k) return fail(INFV_ERR_INVALID, "forward: null k");
    if (new_doc) infv_ltm_reset(h);
    HIP_TRY(h->kbar_ws.reserve((size_t)T * h->d * sizeof(float)));
    if (int rc = infv_ltm_pool(h, k, T, h->kbar_ws.as<float>(), stream)) return rc;
    return infv_ltm_step(h, h->kbar_ws.as<float>(), T, q, Q, proj, u, ctx, stream);
}

int infv_ltm_forward_into(infv_ltm_handle h, const void* k, int32_t token_dtype, int32_t T, float* kbar, const float* q, int32_t Q,
                          const infv_ltm_proj* proj, const double* u, float* ctx, void* stream) {
    if (int rc = infv_ltm_set_token_dtype(h, token_dtype)) return rc;
    if (T <= 0) return fail(INFV_ERR_INVALID, "forward_into: T must be positive");
    if (int rc = infv_ltm_pool(h, k, T, kbar, stream)) return rc;
    return infv_ltm_step(h, kbar, T, q, Q, proj, u, ctx, stream);
}

}  // extern "C"

// ---- whole-video fast path -----------------------------------------------------------------------
// Per chunk ONE small launch of chain_kernel's role S (draw -> score recurrence -> alpha, sticky
// histogram, gather table); per sub-batch ONE launch of uc_kernel (memory update + read-out of all its
// chunks) on its own stream, overlapping role S of the next sub-batch.
namespace {

struct StepS {                       // critical work of the chunk entering the chain
    const Operator* op; bool inf; const float* Snew; const double* u; int sk; long ss;   // Snew: scores part of the chunk's first output row
};

constexpr int kPollDelay = 0;       // units of 64 clocks between role S's deposit and its first poll (INFV_S_POLL_DELAY in the experiments build)

struct FastPipe {
    infv_ltm_handle h; const Plan& plan; int Q; const ProjPtrs& pp; hipStream_t stream;
    long counter = 0;                // chunks that entered the chain in this call

    size_t alpha_slot() const { return (size_t)h->L * h->H * Q * h->N; }
    size_t asum_slot() const { return (size_t)h->L * h->H * Q; }
    size_t tab_slot() const { return (size_t)h->L * h->N * 16; }
    size_t crit_slot() const { return (size_t)h->L * h->H * Q * 128; }
    bool last_v2 = false;            // the last persistent launch ran chain_batch3_kernel (scores rebuilt by alpha_rows2)
    const float* last_snew = nullptr; int last_sk = 1; long last_ss = 0; float* last_sp_out = nullptr; int batch_launches = 0;

    int launch_s(const StepS& st) {
        ChainArgs a;
        memset(&a, 0, sizeof(a));
        const int QT = (Q + kQTile - 1) / kQTile, QS = chain_s_tiles(Q);
        a.N = h->N; a.H = h->H; a.Q = Q; a.QT = QT; a.QS = QS; a.L = h->L; a.S = h->S; a.d4 = h->d / 4; a.dm4 = h->dm / 4;
        a.st = plan.sticky();
        ChainRoleS& s = a.s;
        s.n_blocks = h->H * QS * h->L;
        s.op = st.op->view();
        s.draw_mode = st.inf ? (h->cfg.sticky ? 1 : 2) : 0;
        if (s.draw_mode == 1) {
            if (!st.u) return fail(INFV_ERR_INVALID, "sticky consolidation needs the Gibbs uniforms u");
            if (counter == 0 && h->parts <= 0) return fail(INFV_ERR_STATE, "no sticky histogram available");
        }
        const long slot = counter % h->ring;
        // sticky histogram: the first step of a call may inherit float partials from the per-call path;
        // afterwards the totals live in the fixed-point ring (read slot k-1, accumulate k, clear k+1)
        s.part_prev = h->bin_part[h->pc].as<float>(); s.parts = h->parts;
        s.acc_prev = (counter > 0) ? h->mass_acc[(counter + 2) % 3].as<unsigned long long>() : nullptr;
        s.acc_next = h->mass_acc[counter % 3].as<unsigned long long>();
        s.acc_clear = h->mass_acc[(counter + 1) % 3].as<unsigned long long>();
        s.probs_override = h->probs_override.as<float>(); s.override_mask = h->override_mask;
        s.u = st.u; s.uniform_idx = plan.uniform_idx.as<int32_t>();
        s.probs_out = h->probs.as<float>(); s.bins_out = h->bins.as<int32_t>(); s.idx_out = h->idx.as<int32_t>();
        if (h->trace_cap > counter) {
            s.probs_tr = h->trace_probs ? h->trace_probs + (size_t)counter * h->L * h->n_bins : nullptr;
            s.bins_tr = h->trace_bins ? h->trace_bins + (size_t)counter * h->L * h->S : nullptr;
        }
        s.tab_out = h->tab_ring.as<int32_t>() + slot * tab_slot();
        s.Sp_prev = h->Sp[h->sc].as<float>(); s.Sp_next = h->Sp[h->sc ^ 1].as<float>();
        s.Snew = st.Snew; s.snew_ld = h->L * h->dm + h->L * h->H * Q; s.snew_splitk = st.sk; s.snew_split_stride = st.ss;
        s.cq = h->cqbuf.as<float>();
        s.w = plan.w.as<float>(); s.w_out = plan.w_out;
        s.alpha_out = h->alpha_ring.as<float>() + slot * alpha_slot();
        s.asum_out = h->asum_ring.as<float>() + slot * asum_slot();
        {
            static long long* dbg = [] { long long* p = nullptr; if (exp_env("INFV_CHAIN_STAMPS")) { (void)hipMalloc(&p, 32 * sizeof(long long)); (void)hipMemset(p, 0, 32 * sizeof(long long)); } return p; }();
            a.dbg = dbg;
            static int stamp_calls = 0;
            if (dbg && (++stamp_calls % 300) == 0) {
                long long hb[32];
                (void)hipStreamSynchronize(stream);
                (void)hipMemcpy(hb, dbg, sizeof(hb), hipMemcpyDeviceToHost);
                fprintf(stderr, "[stamps x10ns] S:");
                for (int i = 1; i <= 5; ++i) fprintf(stderr, " %lld", hb[i] - hb[i - 1]);
                fprintf(stderr, " [draw: sync %lld probs %lld scan %lld search %lld tab %lld] | clk %.0f MHz | S total %lld\n",
                        hb[12] - hb[1], hb[13] - hb[12], hb[14] - hb[13], hb[15] - hb[14], hb[2] - hb[15],
                        100.0 * (double)(hb[29] - hb[24]) / (double)(hb[5] - hb[0]), hb[5] - hb[0]);
            }
            Timed t_(h->prof, INFV_KERNEL_CHAIN, stream);
            HIP_TRY(launch_chain(a, stream));
        }
        if (s.draw_mode == 1) h->override_mask = 0;
        h->sc ^= 1;
        h->lastQ = Q;
        h->last_fast = true;
        ++counter;
        return INFV_OK;
    }

    // role S of the WHOLE call in one launch: `n` steps in `n_batches` sub-batches of `sub` (the last may be shorter) whose S'new
    // rows arrive in workspace set (sub-batch % n_sets)
    struct CallLong { int sub, n_batches; const float* const* sets; int n_sets; int sk_last; long ss_last;
                      const unsigned int* tiles_s; int n_tiled, tiles_full, tiles_last; };
    int launch_s_call(int n, int sub, int n_batches, const float* const* sets, int n_sets, int sk_main, long ss_main, int sk_last, long ss_last,
                      const double* u, const float* uf, const unsigned int* tiles_s = nullptr, int n_tiled = 0, int tiles_full = 0, int tiles_last = 0) {
        const CallLong cl{sub, n_batches, sets, n_sets, sk_last, ss_last, tiles_s, n_tiled, tiles_full, tiles_last};
        return launch_s_batch(n, sets[0], sk_main, ss_main, u, uf, &cl);
    }

    // role S of `n` consecutive infinite-memory chunks in one persistent launch
    int launch_s_batch(int n, const float* Snew, int sk, long ss, const double* u, const float* uf, const CallLong* cl = nullptr) {
        static int prev_n = 0;                     // (experiments: steps of the previous launch, for the stamps' average)
        ChainBatchArgs b;
        memset(&b, 0, sizeof(b));
        if (cl != nullptr) {
            // the sub-batch description travels through device memory (a small kernel in front of the launch writes it)
            ChainCallDesc cd;
            memset(&cd, 0, sizeof(cd));
            cd.ready = h->call_flags.as<unsigned int>(); cd.done = h->call_flags.as<unsigned int>() + 64;
            cd.tiles_s = cl->tiles_s; cd.n_tiled = cl->n_tiled; cd.tiles_full = cl->tiles_full; cd.tiles_last = cl->tiles_last;
            cd.sub = cl->sub; cd.n_batches = cl->n_batches; cd.n_sets = cl->n_sets;
            for (int i = 0; i < cl->n_sets && i < kCallSets; ++i) cd.snew_set[i] = cl->sets[i];
            cd.sk_last = cl->sk_last; cd.ss_last = cl->ss_last;
            cd.stats = h->call_stats.as<long long>();
            ChainCallDesc* dst = reinterpret_cast<ChainCallDesc*>(h->call_flags.as<char>() + 512);
            HIP_TRY(launch_chain_call_desc(dst, cd, stream));
            b.call = dst; b.call_sub = cl->sub;
        }
        const int QS = chain_s_tiles(Q);
        b.N = h->N; b.H = h->H; b.Q = Q; b.QS = QS; b.L = h->L; b.S = h->S;
        b.st = plan.sticky();
        b.op = plan.inf.view();
        b.draw_mode = h->cfg.sticky ? 1 : 2;
        if (b.draw_mode == 1) {
            if (!u) return fail(INFV_ERR_INVALID, "sticky consolidation needs the Gibbs uniforms u");
            if (counter == 0 && h->parts <= 0) return fail(INFV_ERR_STATE, "no sticky histogram available");
        }
        b.n_steps = n; b.step0 = counter; b.ring = h->ring;
        b.first_from_parts = (counter == 0) ? 1 : 0;
        b.first_from_acc = (counter > 0 && batch_launches == 0) ? 1 : 0;   // the step before ran in a per-chunk launch (first chunk of a document)
        b.mbox = h->mbox.as<unsigned long long>();
        { static int* rep = [] { int* p = nullptr; if (exp_env("INFV_XCC_REPORT")) { (void)hipMalloc(&p, 1024 * sizeof(int)); (void)hipMemset(p, 0xff, 1024 * sizeof(int)); } return p; }();
          b.xcc_report = rep;
          static int rep_calls = 0;
          if (rep && (++rep_calls % 16) == 0) {
              int hb[1024];
              (void)hipStreamSynchronize(stream);
              (void)hipMemcpy(hb, rep, sizeof(hb), hipMemcpyDeviceToHost);
              fprintf(stderr, "[batch-S placement] (plain<<8 | xcc) per workgroup:");
              for (int i = 0; i < 1024 && hb[i] != -1; ++i) fprintf(stderr, " %x", hb[i]);
              fprintf(stderr, "\n");
          } }
        b.part_prev = h->bin_part[h->pc].as<float>(); b.parts = h->parts;
        for (int i = 0; i < 3; ++i) b.acc[i] = h->mass_acc[i].as<unsigned long long>();
        b.arrive = h->sync_words.as<unsigned int>(); b.error = h->err_dev;
        b.spin_limit = h->spin_limit; b.expect_extra = h->expect_extra;
        { static const int pd = [] { const char* e = exp_env("INFV_S_POLL_DELAY"); return e ? atoi(e) : kPollDelay; }(); b.poll_delay = pd; }
        { static const int fl = [] { const char* e = exp_env("INFV_S_FLAGS"); return e ? atoi(e) : 0; }(); b.exp_flags = fl; }
        if (h->trace_cap > counter) {
            b.trace_steps = (int)((h->trace_cap - counter < n) ? h->trace_cap - counter : n);
            b.probs_tr = h->trace_probs ? h->trace_probs + (size_t)counter * h->L * h->n_bins : nullptr;
            b.bins_tr = h->trace_bins ? h->trace_bins + (size_t)counter * h->L * h->S : nullptr;
        }
        b.probs_override = h->probs_override.as<float>(); b.override_mask = h->override_mask;
        b.u = u; b.uf = uf; b.uniform_idx = plan.uniform_idx.as<int32_t>();
        b.probs_out = h->probs.as<float>(); b.bins_out = h->bins.as<int32_t>(); b.idx_out = h->idx.as<int32_t>();
        b.tab_ring = h->tab_ring.as<int32_t>(); b.tab_slot = (long)tab_slot();
        b.tabb_ring = h->tabb_ring.as<int32_t>();
        b.crit_ring = h->crit_ring.as<float>(); b.crit_slot = (long)crit_slot();
        b.publish_init = (batch_launches == 0) ? 1 : 0;
        b.alpha_ring = h->alpha_ring.as<float>(); b.alpha_slot = (long)alpha_slot();
        b.asum_ring = h->asum_ring.as<float>(); b.asum_slot = (long)asum_slot();
        b.Sp_in = h->Sp[h->sc].as<float>(); b.Sp_out = h->Sp[h->sc ^ 1].as<float>();
        b.Snew = Snew; b.snew_ld = h->L * h->dm + h->L * h->H * Q; b.snew_splitk = sk; b.snew_split_stride = ss;
        b.cq = h->cqbuf.as<float>(); b.w = plan.w.as<float>(); b.w_out = plan.w_out;
        {
            static long long* dbg = [] { long long* p = nullptr; if (exp_env("INFV_CHAIN_STAMPS")) { (void)hipMalloc(&p, 32 * sizeof(long long)); (void)hipMemset(p, 0, 32 * sizeof(long long)); } return p; }();
            b.dbg = dbg;
            static int calls = 0;
            if (dbg && (++calls % 8) == 0) {
                long long hb[32];
                (void)hipStreamSynchronize(stream);
                (void)hipMemcpy(hb, dbg, sizeof(hb), hipMemcpyDeviceToHost);
                fprintf(stderr, "[batch-S stamps x10ns] wait+read %lld draw %lld tab %lld recurrence %lld row-phase+add %lld - %lld alpha-out %lld | step %lld\n",
                        hb[1] - hb[0], hb[2] - hb[1], hb[3] - hb[2], hb[4] - hb[3], hb[5] - hb[4], hb[6] - hb[5], hb[7] - hb[6], hb[7] - hb[0]);
                // shader cycles of the same step (s_memtime): cycles / (10 ns ticks) = effective clock in units of 100 MHz
                fprintf(stderr, "[batch-S clock] step %lld shader cycles over %lld x10ns -> %.2f GHz\n", hb[15] - hb[8], hb[7] - hb[0],
                        (hb[7] - hb[0]) > 0 ? 0.1 * (double)(hb[15] - hb[8]) / (double)(hb[7] - hb[0]) : 0.0);
                if (prev_n > 2) fprintf(stderr, "[batch-S avg] %.3f us per step over steps 1..%d of the previous launch\n",
                                        0.01 * (double)(hb[17] - hb[16]) / (prev_n - 2), prev_n - 1);
                fprintf(stderr, "[batch-S launch] previous end -> entry %.2f us, set-up %.2f us, step 0 %.2f us, last step + exit %.2f us, entry -> end %.2f us\n",
                        0.01 * (double)(hb[18] - hb[21]), 0.01 * (double)(hb[19] - hb[18]), 0.01 * (double)(hb[16] - hb[19]),
                        0.01 * (double)(hb[20] - hb[17]), 0.01 * (double)(hb[20] - hb[18]));
            }
        }
        {
            Timed t_(h->prof, INFV_KERNEL_CHAIN, stream);
            if (!(skip_mask() & 8)) HIP_TRY(launch_chain_batch(b, stream));
        }
        last_v2 = chain_batch2_applies(b);
        prev_n = n;
        last_snew = Snew; last_sk = sk; last_ss = ss; last_sp_out = b.Sp_out;
        ++batch_launches;
        if (b.draw_mode == 1) h->override_mask = 0;
        h->sc ^= 1;
        h->lastQ = Q;
        h->last_fast = true;
        counter += n;
        return INFV_OK;
    }

    // the persistent role S leaves SCORES in the alpha ring: turn the `n` slots from slot0 into softmax weights + row sums
    int launch_alpha(int n, long slot0, hipStream_t s, bool last_batch) {
        if (skip_mask() & (4 | 16)) return INFV_OK;                 // (timing experiments: 4 = no UC and no alpha launches, 16 = no alpha, 32 = no UC)
        if (last_v2) {
            // chain_batch3_kernel published point scores + drawn bins: rebuild the full score rows, then the weights
            AlphaRows2Args r;
            memset(&r, 0, sizeof(r));
            r.N = h->N; r.H = h->H; r.Q = Q; r.L = h->L; r.rows = plan.inf.rows; r.tabw = plan.inf.tabw; r.n_steps = n;
            r.slot0 = slot0 % h->ring; r.ring = h->ring;
            r.crit_ring = h->crit_ring.as<float>(); r.crit_slot = (long)crit_slot();
            r.tabb_ring = h->tabb_ring.as<int32_t>(); r.tab_slot = (long)tab_slot();
            r.Snew = last_snew; r.snew_ld = h->L * h->dm + h->L * h->H * Q; r.snew_splitk = last_sk; r.snew_split_stride = last_ss;
            r.cq = h->cqbuf.as<float>(); r.w = plan.w.as<float>(); r.w_out = plan.w_out;
            r.box_val = plan.inf.box_val.as<float>(); r.box_row = plan.inf.box_row.as<int32_t>();
            r.alpha_ring = h->alpha_ring.as<float>(); r.alpha_slot = (long)alpha_slot();
            r.asum_ring = h->asum_ring.as<float>(); r.asum_slot = (long)asum_slot();
            r.Sp_out = last_batch ? last_sp_out : nullptr;
            HIP_TRY(launch_alpha_rows2(r, s));
            return INFV_OK;
        }
        HIP_TRY(launch_alpha_rows(h->alpha_ring.as<float>(), (long)alpha_slot(), h->asum_ring.as<float>(), (long)asum_slot(),
                                  slot0 % h->ring, h->ring, n, h->L * h->H * Q, h->N, plan.w.as<float>(), plan.w_out, s));
        return INFV_OK;
    }

    // memory update + read-out of `n` chunks whose role S used ring slots slot0.. ; flips the B / KV ping-pong
    int launch_uc(const Operator& op, bool inf, int n, long slot0, const float* R, const float* Pn, int sk, long ss,
                  float* ctx, hipStream_t ucs) {
        UcArgs u;
        memset(&u, 0, sizeof(u));
        u.N = h->N; u.H = h->H; u.Q = Q; u.L = h->L; u.d = h->d; u.dm = h->dm; u.tabw = op.tabw;
        u.op = op.view();
        u.gather = inf ? 1 : 0;
        u.have_state = h->has_memory ? 1 : 0;
        u.n_chunks = n; u.slot0 = slot0 % h->ring; u.ring = h->ring;
        u.tab = h->tab_ring.as<int32_t>(); u.tab_slot = (long)tab_slot();
        u.alpha = h->alpha_ring.as<float>(); u.asum = h->asum_ring.as<float>();
        u.R = R; u.Pnew = Pn; u.p_ld = h->L * h->dm + h->L * h->H * Q; u.splitk = sk; u.split_stride = ss;
        u.B_prev = h->B[h->cur].as<float>(); u.KV_prev = h->KV[h->cur].as<float>();
        u.B_next = h->B[h->cur ^ 1].as<float>(); u.KV_next = h->KV[h->cur ^ 1].as<float>();
        for (int l = 0; l < h->L; ++l) u.bv[l] = pp.bv[l];
        u.ctx = ctx;
        {
            static long long* dbg = [] { long long* p = nullptr; if (exp_env("INFV_UC_STAMPS")) { (void)hipMalloc(&p, 16 * sizeof(long long)); (void)hipMemset(p, 0, 16 * sizeof(long long)); } return p; }();
            u.dbg = dbg;
            static int calls = 0;
            if (dbg && (++calls % 8) == 0) {
                long long hb[16];
                (void)hipStreamSynchronize(ucs);
                (void)hipMemcpy(hb, dbg, sizeof(hb), hipMemcpyDeviceToHost);
                fprintf(stderr, "[uc stamps x10ns] park %lld prefetch-issue %lld barrier %lld gather %lld mfma %lld red-barrier %lld epilogue %lld | chunk total %lld\n",
                        hb[1] - hb[0], hb[2] - hb[1], hb[3] - hb[2], hb[4] - hb[3], hb[5] - hb[4], hb[6] - hb[5], hb[7] - hb[6], hb[7] - hb[0]);
            }
            Timed t_(h->prof, INFV_KERNEL_UC, ucs);
            if (!(skip_mask() & (4 | 32))) HIP_TRY(::infv::launch_uc(u, ucs));
        }
        h->cur ^= 1;
        h->has_memory = true;
        return INFV_OK;
    }
};

// fast-path chunk-parallel stage after the pool: new rows R, then ONE GEMM whose output rows are
// [ V' projection of the row (L*dm) | its scores under the call's pre-multiplied queries (L*H*Q) ]
int project_chunks_fast(infv_ltm_handle h, const Plan& plan, bool inf, const float* kbar, int n_chunks, int T, int Q,
                        const ProjPtrs& pp, int set, int* splitk, long* split_stride, hipStream_t stream, int gemm_pad,
                        bool defer_values = false, int rset = -1, bool rows_done = false, const float* r_ext = nullptr,
                        void* const* planes_ext = nullptr, bool small_tiles = false) {
    if (rset < 0) rset = set;                                 // R buffer of the sub-batch (the pooling kernel may have filled it)
    // r_ext / planes_ext: the call-long pooling launch has written the sub-batch's rows (and their bf16 planes) elsewhere
    const Operator& op = inf ? plan.inf : plan.first;
    const long M = (long)n_chunks * op.rows;
    const int n_out = h->L * h->H * Q;
    const long ld = (long)h->L * h->dm + n_out;
    const int sk_max = 8;
    const size_t needR = (size_t)(M ? M : 1) * h->d * sizeof(float), needP = (size_t)(M ? M : 1) * ld * sk_max * sizeof(float);
    if ((r_ext == nullptr && needR > h->R_ws[rset].bytes) || (M < 1024 ? needP : needP / sk_max) > h->P_ws[set].bytes) HIP_TRY(hipDeviceSynchronize());
    if (r_ext == nullptr) HIP_TRY(h->R_ws[rset].reserve(needR));
    HIP_TRY(h->P_ws[set].reserve(M < 1024 ? needP : needP / sk_max));
    const float* Rrows = r_ext != nullptr ? r_ext : h->R_ws[rset].as<float>();
    if (!rows_done && r_ext == nullptr) {
        Timed t_(h->prof, INFV_KERNEL_ROWS, stream);
        HIP_TRY(launch_rows(kbar, n_chunks, T, h->d, op.view(), h->R_ws[rset].as<float>(), stream));
    }
    const int v_cols = h->L * h->dm;
    if (skip_mask() & 2) {
        *splitk = project_splitk((int)M, h->d);
    } else if (defer_values && M >= 1024) {
        Timed t_(h->prof, INFV_KERNEL_PROJECT, stream);
        HIP_TRY(launch_project_scores((int)M, h->d, n_out, h->qt_buf.as<float>(), Rrows,
                                      h->P_ws[set].as<float>() + v_cols, (int)ld, stream, gemm_pad));
        *splitk = 1;
    } else if (h->proj_x6 && h->w3_valid && inf && M >= 1024 && h->d % 32 == 0 && !defer_values) {
        // [V'new | S'new] = R . [Wv ; q~]^T from three bf16 planes per operand (the weights' planes were made once for the call)
        Timed t_(h->prof, INFV_KERNEL_PROJECT, stream);
        const size_t szR = (size_t)M * h->d * sizeof(__bf16);
        if (planes_ext == nullptr) {
            if (szR > h->r3[0].bytes) {
                HIP_TRY(hipDeviceSynchronize());
                for (int i = 0; i < 3; ++i) HIP_TRY(h->r3[i].reserve((size_t)h->maxC * op.rows * h->d * sizeof(__bf16) > szR ? (size_t)h->maxC * op.rows * h->d * sizeof(__bf16) : szR));
            }
            HIP_TRY(launch_split3_rows(Rrows, h->d, M, h->d, h->r3[0].p, h->r3[1].p, h->r3[2].p, 0, M, stream));
        }
        SplitGemm6 g{};
        for (int i = 0; i < 3; ++i) { g.A[i] = planes_ext != nullptr ? static_cast<const __bf16*>(planes_ext[i]) : h->r3[i].as<__bf16>(); g.B[i] = h->w3[i].as<__bf16>(); }
        g.lda = h->d; g.ldb = h->d; g.C = h->P_ws[set].as<float>(); g.ldc = ld; g.M = (int)M; g.N = (int)ld; g.K = h->d;
        g.narrow = small_tiles ? 1 : 0;                          // 128 x 128 tiles (many short workgroups: the same bits) instead of 384 x 256
        HIP_TRY(launch_gemm_x6(g, stream));
        *splitk = 1;
    } else {
        Timed t_(h->prof, INFV_KERNEL_PROJECT, stream);
        HIP_TRY(launch_project_fast((int)M, h->d, h->dm, h->L, n_out, pp, h->qt_buf.as<float>(), Rrows,
                                    h->P_ws[set].as<float>(), splitk, stream, gemm_pad));
    }
    *split_stride = M * ld;
    return INFV_OK;
}

// batched new-row scores of `n_chunks` projected chunks of workspace set `set`
int batch_scores(infv_ltm_handle h, const Operator& op, int n_chunks, const float* q, int Q,
                 const ProjPtrs& pp, int set, int sk, long ss, bool want_cq, hipStream_t stream) {
    const long n_cols = (long)h->L * 2 * h->dm;
    const size_t need = (size_t)n_chunks * h->L * h->H * Q * (op.rows ? op.rows : 1) * sizeof(float);
    if (need > h->Snew_ws[set].bytes) HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(h->Snew_ws[set].reserve(need));
    Timed t_(h->prof, INFV_KERNEL_SCORES, stream);
    HIP_TRY(launch_new_scores(q, Q, h->H, h->L, n_chunks, op.rows, h->P_ws[set].as<float>(), (long)op.rows * n_cols, n_cols,
                              2L * h->dm, sk, ss, pp, h->Snew_ws[set].as<float>(), want_cq ? h->cqbuf.as<float>() : nullptr,
                              stream));
    return INFV_OK;
}

// The worker streams are shared by every handle of a device.  The HIP runtime multiplexes a process's streams onto a few
// hardware queues (4 by default, GPU_MAX_HW_QUEUES): a second handle with four more streams of its own shared queues with
// the first one's and its pipeline ran 12 % slower (107 k against 120 k chunks/s for the second engine of bench.py).
// Handles are not re-entrant and their calls are issued from one host thread at a time, so FIFO order within a shared
// stream is the order the host issued the work in; cross-stream dependencies are events, as before.
struct SharedStreams { hipStream_t side = nullptr, pools = nullptr, ucs = nullptr, chain = nullptr; std::mutex issue; };   // issue: one consolidate call enqueues at a time
// experiment INFV_CU_MASK=K: the first K CUs (in the runtime's CU-mask bit order) belong to role S alone -- its launches go to a
// stream masked to them, the three worker streams are masked to the rest.  0 = no masks (default).
bool host_serial() { static const bool v = exp_env("INFV_SERIAL") != nullptr; return v; }   // timing experiments: the host synchronises the streams (no overlap)
int cu_mask_k() { static const int k = [] { const char* e = exp_env("INFV_CU_MASK"); return e ? atoi(e) : 0; }(); return k; }
int shared_streams(int dev, SharedStreams** out) {
    static std::mutex mu;
    static SharedStreams pool[64];
    if (dev < 0 || dev >= 64) return fail(INFV_ERR_INVALID, "device index out of range");
    std::lock_guard<std::mutex> lock(mu);
    SharedStreams& p = pool[dev];
    if (!p.side) {
        int lo = 0, hi = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));   // lo = least urgent
        // INFV_PRIO_UCS / _POOL / _SIDE (experiments): -1 most urgent, 0 normal, 1 least urgent
        auto prio = [&](const char* name, int dflt) { const char* e = exp_env(name); int v = e ? atoi(e) : dflt; return v < hi ? hi : (v > lo ? lo : v); };
        const int prio_ucs = prio("INFV_PRIO_UCS", 0), prio_pool = prio("INFV_PRIO_POOL", lo), prio_side = prio("INFV_PRIO_SIDE", lo);
        const int K = cu_mask_k();
        if (K > 0) {
            int cus = 0;
            HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
            const int words = (cus + 31) / 32;
            std::vector<uint32_t> ms(words, 0u), mo(words, 0u);
            for (int i = 0; i < cus; ++i) ((i < K) ? ms : mo)[i / 32] |= 1u << (i % 32);
            HIP_TRY(hipExtStreamCreateWithCUMask(&p.chain, words, ms.data()));
            HIP_TRY(hipExtStreamCreateWithCUMask(&p.ucs, words, mo.data()));
            HIP_TRY(hipExtStreamCreateWithCUMask(&p.pools, words, mo.data()));
            HIP_TRY(hipExtStreamCreateWithCUMask(&p.side, words, mo.data()));
            *out = &p;
            return INFV_OK;
        }
        // Role S's stream: the call-long launch spin-waits on work of the three worker streams, so it must not sit in front of them
        // in a hardware queue.  The runtime keeps separate queues per priority level: highest priority for role S alone, normal
        // for the UC stream, lowest for pooling and GEMM.
        static const bool own_chain_stream = [] { const char* e = exp_env("INFV_CHAIN_STREAM"); return e && atoi(e) != 0; }();
        if (own_chain_stream) HIP_TRY(hipStreamCreateWithPriority(&p.chain, hipStreamNonBlocking, hi));
        HIP_TRY(hipStreamCreateWithPriority(&p.ucs, hipStreamNonBlocking, prio_ucs));
        HIP_TRY(hipStreamCreateWithPriority(&p.pools, hipStreamNonBlocking, prio_pool));
        HIP_TRY(hipStreamCreateWithPriority(&p.side, hipStreamNonBlocking, prio_side));   // last: marks the set complete
    }
    *out = &p;
    return INFV_OK;
}

}  // namespace

namespace infv {
// the normal-priority worker stream of the shared set, for the video Q-former's layer-major schedule (its side stream
// runs the per-call LTM chain of the later layers while the caller's stream runs their short-term attention)
int shared_worker_stream(hipStream_t* out) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    SharedStreams* sh = nullptr;
    if (int rc = shared_streams(dev, &sh)) return rc;
    *out = sh->ucs;
    return INFV_OK;
}
}  // namespace infv

namespace {
int ensure_side_stream(infv_ltm_handle h) {
    if (h->side) return INFV_OK;
    SharedStreams* sh = nullptr;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));                              // (the caller's current device: where the handle was created)
    if (int rc = shared_streams(dev, &sh)) return rc;
    h->ucs = sh->ucs; h->pools = sh->pools; h->chain_s = sh->chain;
    for (int i = 0; i < kPSets; ++i) {
        HIP_TRY(hipEventCreateWithFlags(&h->ev_s[i], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&h->ev_uc[i], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&h->ev_p[i], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&h->ev_pool[i], hipEventDisableTiming));
    }
    for (int i = 0; i < kRSets; ++i) HIP_TRY(hipEventCreateWithFlags(&h->ev_r[i], hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&h->ev_in, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&h->ev_start, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&h->ev_q, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&h->ev_chain, hipEventDisableTiming));
    if (chain_call_long()) {                                   // (experiments build: hand-off words of the call-long launches)
        HIP_TRY(h->call_flags.reserve(1024));                  // [0, 512): the two counters, [512, 1024): ChainCallDesc of the current call
        HIP_TRY(h->call_stats.reserve(64));
        HIP_TRY(hipMemset(h->call_stats.p, 0, 64));
    }
    h->issue_mu = &sh->issue;
    h->side = sh->side;                                       // last: h->side != nullptr means "streams and events exist"
    return INFV_OK;
}

}  // namespace

extern "C" {

// k_: frame tokens (pooled here, on a stream of their own) -- or kbar_pre: frame means [n_chunks][T][d] the caller
// already holds (infv_ltm_consolidate_pooled: the video Q-former's one pass over the tokens produces them)
static int consolidate_impl(infv_ltm_handle h, const void* k_, const float* kbar_pre, int32_t n_chunks, int32_t T, const float* q,
                            int32_t Q, const infv_ltm_proj* proj, const double* u, int32_t new_doc, float* ctx,
                            void* stream_) {
    if (int rc = check_handle(h)) return rc;
    const char* k = static_cast<const char*>(k_);               // byte addressing: the token element size depends on the handle
    if ((!k && !kbar_pre) || !q || !proj || !ctx || n_chunks < 0) return fail(INFV_ERR_INVALID, "consolidate: bad arguments");
    if (int rc = check_chain_error(h)) return rc;
    if (int rc = check_q(h, Q)) return rc;
    Plan* plan = nullptr;
    if (int rc = find_plan(h, T, &plan)) return rc;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const ProjPtrs pp = make_proj(proj, h->L);
    const size_t chunk_k = (size_t)T * h->P * h->d * (h->k_bf16 ? 2 : 4);   // bytes of one chunk's tokens
    const size_t chunk_ctx = (size_t)h->L * Q * h->dm;
    const size_t chunk_u = (size_t)h->L * h->S;
    if (new_doc) infv_ltm_reset(h);
    if (n_chunks == 0) return INFV_OK;
    const int rows_max = plan->first.rows > plan->inf.rows ? plan->first.rows : plan->inf.rows;
    if (plan->dense.on || !chain_supported(h->N, h->S, rows_max, plan->inf.tabw) || (h->L * h->H * Q) % 128 != 0 || (h->L * h->dm) % 128 != 0 ||
        !uc_supported(h->N, h->d, h->dm, plan->inf.tabw, rows_max)) {
        // shapes the fused chain kernel cannot hold in LDS: per-chunk stage kernels
        for (int c = 0; c < n_chunks; ++c) {
            if (kbar_pre) {
                if (int rc = infv_ltm_step(h, kbar_pre + (size_t)c * T * h->d, T, q, Q, proj, u ? u + c * chunk_u : nullptr,
                                           ctx + c * chunk_ctx, stream_)) return rc;
            } else if (int rc = infv_ltm_forward(h, k + c * chunk_k, T, q, Q, proj, u ? u + c * chunk_u : nullptr, 0,
                                                 ctx + c * chunk_ctx, stream_)) return rc;
        }
        return INFV_OK;
    }
    if (int rc = ensure_side_stream(h)) return rc;
    // One call enqueues at a time per device: the worker streams are shared by every handle, and a call-long role-S launch
    // spin-waits on work that must not end up behind another call's hand-off kernels in a shared stream.
    // (The per-sub-batch launches of the shipped pipeline only order themselves through events: no lock, as in rounds 1-4.)
    std::unique_lock<std::mutex> issue_lock(*h->issue_mu, std::defer_lock);
    if (chain_call_long()) issue_lock.lock();
    // Padding LDS caps the pooling kernel's occupancy at ONE 512-thread workgroup (84 KB: two do not fit, one leaves room for
    // a 74 KB workgroup of the loader-wave GEMM) per CU, so a role-S workgroup
    // always finds LDS and wave slots and the pool's bytes in flight stay bounded.  The GEMMs carry no padding any more
    // (INFV_GEMM_PAD): their workgroups (36 KB, one wave per SIMD) co-reside with a pooling workgroup -- MFMA work beside
    // memory work -- instead of taking the CU away from it.
    static const int kPoolPad = [] { const char* e = exp_env("INFV_POOL_PAD"); return e ? atoi(e) : 84 * 1024; }();
    static const int kGemmPad = [] { const char* e = exp_env("INFV_GEMM_PAD"); return e ? atoi(e) : 0; }();
    FastPipe pipe{h, *plan, Q, pp, stream};
    h->wv_split_valid = false;                                // the caller's value weights may have changed since the last call
    // the pooling of the first sub-batches depends on the caller's tokens only: it starts here, beside the first chunk
    // ONE pooling launch per call (round 6, shipped for calls of 768 chunks and more): the pooling stream is what bounds a long call,
    // and between its per-sub-batch launches sat an event record, a wait for the R set's last reader and the ramp of 2688 fresh
    // workgroups (13 us per boundary by the residency stamps).  The launch walks every (chunk, row) of the call in order, writes rows
    // and bf16 planes write-through and counts them into one word per sub-batch; the GEMM stream holds on that word with a one-wave
    // flag_wait_kernel (bounded, latches the handle's error word) instead of an event.  Role S, the GEMM, alpha and UC keep their
    // per-sub-batch launches -- round 5 measured this pooling launch only beside a RESIDENT role S, which cost more seats than the
    // boundaries: alone it is 13.3 against 13.7 ms per video on one box, four rounds (profiles/r06_matrix.txt, block 12), same bits.
    // Rows + planes of the whole call live in HBM (0.49 MB per chunk: 1 GB at 2048 chunks); above kPoolCallBudget the call keeps its
    // per-sub-batch pooling launches.  INFV_POOL_CALL (experiments build): 0 never, 1 with the call-long role S, 2 always.
    static const int pool_call_env_mode = [] { const char* e = exp_env("INFV_POOL_CALL"); return e ? atoi(e) : -1; }();
    constexpr size_t kPoolCallBudget = (size_t)16 << 30;
    const size_t pool_call_bytes = (size_t)n_chunks * (size_t)plan->inf.rows * h->d * (sizeof(float) + 3 * sizeof(__bf16));
    const int pool_call_mode = pool_call_env_mode >= 0 ? pool_call_env_mode : ((n_chunks >= 768 && pool_call_bytes <= kPoolCallBudget) ? 2 : 0);
    if (chain_call_long() || pool_call_mode == 2) {   // hand-off counters of the call-long launches restart with the call (before ev_start: the pooling stream starts behind it)
        const size_t need_pd = ((size_t)n_chunks + 1) * sizeof(unsigned int);
        if (need_pd > h->pool_done.bytes) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(h->pool_done.reserve(need_pd < 32768 ? 32768 : need_pd)); }
        HIP_TRY(hipMemsetAsync(h->pool_done.p, 0, need_pd, stream));
        const long cap = ((long)n_chunks + 64) & ~63l;
        if (chain_call_long() && cap > h->gemm_flags_cap) {
            HIP_TRY(hipDeviceSynchronize());
            const long ncap = cap < 8192 ? 8192 : cap;
            HIP_TRY(h->gemm_flags.reserve((size_t)(128 + 2 * ncap) * sizeof(unsigned int) + 1024));
            h->gemm_flags_cap = ncap;
        }
        if (chain_call_long()) {
            HIP_TRY(hipMemsetAsync(h->gemm_flags.p, 0, (size_t)(128 + 2 * h->gemm_flags_cap) * sizeof(unsigned int), stream));
            HIP_TRY(hipMemsetAsync(h->call_flags.p, 0, 512, stream));
        }
    }
    HIP_TRY(hipEventRecord(h->ev_start, stream));
    HIP_TRY(hipMemsetAsync(h->mass_acc[0].p, 0, h->mass_acc[0].bytes, stream));   // slot of the call's first step
    {   // rings of role S's per-chunk outputs (sized for this call's Q)
        const size_t need_a = (size_t)h->ring * pipe.alpha_slot() * sizeof(float);
        if (need_a > h->alpha_ring.bytes) {
            HIP_TRY(hipDeviceSynchronize());
            HIP_TRY(h->alpha_ring.reserve(need_a));
            HIP_TRY(h->asum_ring.reserve((size_t)h->ring * pipe.asum_slot() * sizeof(float)));
            HIP_TRY(h->tab_ring.reserve((size_t)h->ring * pipe.tab_slot() * sizeof(int32_t)));
            HIP_TRY(h->tabb_ring.reserve((size_t)h->ring * pipe.tab_slot() * sizeof(int32_t)));
            HIP_TRY(h->crit_ring.reserve((size_t)h->ring * pipe.crit_slot() * sizeof(float)));
        }
    }
    hipStream_t side = h->side, ucs = h->ucs;
    // pre-multiplied queries qt = (q/sqrt(dh)) . Wk_h and the bias term cq = q_h . bk_h / sqrt(dh), once per call
    HIP_TRY(launch_qtilde(q, Q, h->H, h->d, h->L, pp, h->qt_buf.as<float>(), h->cqbuf.as<float>(), stream));
    h->w3_valid = false;
    if (h->proj_x6 && h->d % 32 == 0 && ((long)h->L * h->dm + (long)h->L * h->H * Q) % 8 == 0) {
        // bf16 planes of the projection GEMM's weight rows [Wv_0 ; ... ; Wv_{L-1} ; q~] (the caller's weights may change between calls)
        const long n_rows = (long)h->L * h->dm + (long)h->L * h->H * Q;
        const size_t szW = (size_t)n_rows * h->d * sizeof(__bf16);
        if (szW > h->w3[0].bytes) {
            HIP_TRY(hipDeviceSynchronize());
            for (int i = 0; i < 3; ++i) HIP_TRY(h->w3[i].reserve(szW));
        }
        for (int l = 0; l < h->L; ++l)
            HIP_TRY(launch_split3_rows(pp.wv[l], h->d, h->dm, h->d, h->w3[0].p, h->w3[1].p, h->w3[2].p, (long)l * h->dm, n_rows, stream));
        HIP_TRY(launch_split3_rows(h->qt_buf.as<float>(), h->d, (long)h->L * h->H * Q, h->d, h->w3[0].p, h->w3[1].p, h->w3[2].p,
                                   (long)h->L * h->dm, n_rows, stream));
        h->w3_valid = true;
    }
    HIP_TRY(hipEventRecord(h->ev_q, stream));                 // the side stream's projections need no more than this
    // the persistent role S searches in fp32 against the round-ups of the f64 uniforms (equivalent to the f64 compare):
    // converted once per call, off the chain
    const float* uf = nullptr;
    if (u && h->cfg.sticky && chain_batch3_shape_ok(1, plan->sticky().points_ok, plan->inf.rows, h->S, Q)) {
        const size_t need_u = (size_t)n_chunks * chunk_u * sizeof(float);
        if (need_u > h->uf_all.bytes) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(h->uf_all.reserve(need_u)); }
        HIP_TRY(launch_round_up_uniforms(u, h->uf_all.as<float>(), (long)n_chunks * (long)chunk_u, stream));
        uf = h->uf_all.as<float>();
        if (chain_batch3_mailboxes()) {
            // (experiments) mailboxes of role S's exchange: zeroed per call (step tags restart with the call's chunk counter)
            const int G = chain_batch_blocks(h->H, Q, h->L, 1, plan->sticky().points_ok, plan->inf.rows, h->S) / h->L;
            const size_t need_m = chain_mailbox_bytes(h->L, G);
            if (need_m > h->mbox.bytes) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(h->mbox.reserve(need_m)); }
            h->mbox_G = G;
            HIP_TRY(hipMemsetAsync(h->mbox.p, 0, need_m, stream));
        }
    }
    int c = 0;
    bool uc_pending[kPSets] = {};               // ev_uc[set] has been recorded in this call
    if (!h->has_memory) {                                     // first chunk of a document: first-chunk operator, set 1
        const float* kb0 = kbar_pre;
        if (!kb0) {
            if ((size_t)T * h->d * sizeof(float) > h->kbar_ws.bytes) HIP_TRY(hipDeviceSynchronize());
            HIP_TRY(h->kbar_ws.reserve((size_t)T * h->d * sizeof(float)));
            if (int rc = infv_ltm_pool(h, k, T, h->kbar_ws.as<float>(), stream_)) return rc;
            kb0 = h->kbar_ws.as<float>();
        }
        int sk = 1; long ss = 0;
        if (int rc = project_chunks_fast(h, *plan, false, kb0, 1, T, Q, pp, 2, &sk, &ss, stream, 0, false, kRSets)) return rc;
        const long v_cols = (long)h->L * h->dm;             // a GEMM output row is [ V' (L*dm) | scores (L*H*Q) ]
        const StepS st{&plan->first, false, h->P_ws[2].as<float>() + v_cols, nullptr, sk, ss};
        if (int rc = pipe.launch_s(st)) return rc;
        HIP_TRY(hipEventRecord(h->ev_s[2], stream));
        HIP_TRY(hipStreamWaitEvent(ucs, h->ev_s[2], 0));
        if (int rc = pipe.launch_uc(plan->first, false, 1, 0, h->R_ws[kRSets].as<float>(), h->P_ws[2].as<float>(), sk, ss, ctx, ucs)) return rc;
        HIP_TRY(hipEventRecord(h->ev_uc[2], ucs));
        uc_pending[2] = true;
        c = 1;
    } else {
        // continue an existing memory
        if (h->carry_scores && h->last_fast && h->lastQ == Q) {
            // ... handed over by infv_ltm_import_chain_state: the scores under this query and the projected memory are the
            // exporting handle's own; nothing is re-derived, the chain goes on bit for bit
        } else {
            // ... bias-free scores of the current K' rows under this query
            if (h->k_stale)
                if (int rc = infv_ltm_reproject(h, proj, stream_)) return rc;
            Timed t_(h->prof, INFV_KERNEL_SCORES, stream);
            HIP_TRY(launch_new_scores(q, Q, h->H, h->L, 1, h->N, h->KV[h->cur].as<float>(), 0, 2L * h->dm,
                                      (long)h->N * 2 * h->dm, 1, 0, pp, h->Sp[h->sc].as<float>(), h->cqbuf.as<float>(),
                                      stream));
        }
    }
    h->carry_scores = false;
    // ---- sub-batches.  Streams: `side` = chunk-parallel stage of batch b+1, caller's stream = role S of
    //      batch b (one launch per chunk), `ucs` = memory update + read-out of batch b-1 ----
    // INFV_PERSISTENT=0 falls back to one role-S launch per chunk
    static const bool want_persistent = [] { const char* e = exp_env("INFV_PERSISTENT"); return !e || atoi(e) != 0; }();
    const bool persistent = want_persistent &&
        chain_batch_supported(h->N, h->S, plan->inf.rows, plan->inf.tabw, h->H * chain_s_tiles(Q) * h->L) &&
        chain_batch_resident(h->N, h->S, plan->inf.rows, plan->inf.tabw,
                             chain_batch_blocks(h->H, Q, h->L, h->cfg.sticky ? 1 : 2, plan->sticky().points_ok, plan->inf.rows, h->S),
                             h->cfg.sticky ? 1 : 2, plan->sticky().points_ok, Q);
    const int first_c = c;
    // sub-batch size: long calls amortise the per-launch gap of role S over more chunks (42 x 64 new rows = 21 row tiles:
    // 126 score tiles, 252 V' tiles); short ones (e.g. a 256-chunk shard of a multi-GPU run) keep 32 so that the
    // pipeline fills and drains quickly.  INFV_SUB_BATCH overrides.
    static const int sub_env = [] { const char* e = exp_env("INFV_SUB_BATCH"); return e ? atoi(e) : 0; }();
    // (experiments) INFV_SUB_RAMP=<n>: short calls run their first sub-batch with 32 chunks (fast fill) and the following ones
    // with n (fewer role-S launches); needs max_batch_chunks >= n
    static const int ramp_env = [] { const char* e = exp_env("INFV_SUB_RAMP"); return e ? atoi(e) : 0; }();
    int sub = h->maxC;
    if (sub_env > 0) sub = sub_env < h->maxC ? sub_env : h->maxC;
    else if (n_chunks < 768 && sub > 32) sub = 32;             // (28 while the V' GEMM ran on the UC stream; 2.84 -> 2.68 ms per 256 chunks)
    // first chunk of every sub-batch (+ the end of the call)
    std::vector<int> bstart;
    {
        int later = sub;
        if (sub_env <= 0 && n_chunks < 768 && ramp_env > sub) later = ramp_env < h->maxC ? ramp_env : h->maxC;
        // (experiments) tapered schedule of a short call: INFV_TAPER="first,mid,last" -- a small first sub-batch (role S starts early), a
        // small last one (the serial tail behind the last pooling launch: GEMM -> role S -> alpha -> UC of the last sub-batch), the rest
        // in equal pieces of at most `mid` chunks
        static const char* taper_env = exp_env("INFV_TAPER");
        int tf = 0, tm = 0, tl = 0;
        const int n_rest = n_chunks - first_c;
        if (sub_env <= 0 && n_chunks < 768 && taper_env && sscanf(taper_env, "%d,%d,%d", &tf, &tm, &tl) == 3 && tf > 0 && tm > 0 && tl > 0 &&
            tf <= h->maxC && tm <= h->maxC && tl <= h->maxC && n_rest > tf + tl) {
            const int middle = n_rest - tf - tl, pieces = (middle + tm - 1) / tm;
            bstart.push_back(first_c);
            int c0 = first_c + tf;
            for (int i = 0; i < pieces; ++i) { bstart.push_back(c0); c0 += middle / pieces + (i < middle % pieces ? 1 : 0); }
            bstart.push_back(c0);                                // the last sub-batch
            bstart.push_back(n_chunks);
            sub = tf > tm ? tf : tm; if (tl > sub) sub = tl;
        } else if (const char* sched = (sub_env <= 0 && n_chunks < 768) ? exp_env("INFV_SCHED") : nullptr) {
            // (experiments) INFV_SCHED="a,b,c,...": explicit sub-batch sizes; used when they add up to the call's chunks
            std::vector<int> sz; int tot = 0, mx = 0;
            for (const char* p_ = sched; *p_;) { const int v = atoi(p_); sz.push_back(v); tot += v; if (v > mx) mx = v; while (*p_ && *p_ != ',') ++p_; if (*p_ == ',') ++p_; }
            bool ok_ = tot == n_rest && mx <= h->maxC;
            for (int v : sz) ok_ = ok_ && v > 0;
            if (ok_) { int c0 = first_c; for (int v : sz) { bstart.push_back(c0); c0 += v; } bstart.push_back(n_chunks); sub = mx; }
            else { for (int c0 = first_c; c0 < n_chunks; c0 += sub) bstart.push_back(c0); bstart.push_back(n_chunks); }
        } else {
        for (int c0 = first_c, i = 0; c0 < n_chunks; ++i) { bstart.push_back(c0); c0 += (i == 0) ? sub : later; }
        bstart.push_back(n_chunks);
        if (later > sub) sub = later;                           // (workspaces below are sized for the largest sub-batch)
        }
    }
    const int n_batches = (int)bstart.size() - 1;
    const size_t rows = plan->inf.rows;
    std::vector<int> sks(n_batches > 0 ? n_batches : 1, 1);
    std::vector<long> sss(n_batches > 0 ? n_batches : 1, 0);
    auto batch_range = [&](int b, int* c0, int* nb) {
        *c0 = bstart[b];
        *nb = bstart[b + 1] - bstart[b];
    };
    // The pooling has its own stream so that the HBM-bound pooling of batch b+2 overlaps the MFMA-bound projection of
    // batch b+1 (pooled frames are triple-buffered either way); INFV_SPLIT_POOL=0 puts it back on the side stream.
    // (Round 1 measured this worse, 87 k vs 95 k chunks/s, because role S was then sensitive to every concurrent
    // kernel; with chain_batch2_kernel and no padding LDS on the GEMMs it is better: 112 k vs 102 k.)
    static const bool split_pool_env = [] { const char* e = exp_env("INFV_SPLIT_POOL"); return !e || atoi(e) != 0; }();
    const bool split_pool = split_pool_env && !kbar_pre;      // (frame means handed in: there is no pooling stage)
    hipStream_t pools = split_pool ? h->pools : side;
    bool p_pending[kPSets] = {};                // ev_p[set] has been recorded in this call
    bool r_pending[kRSets] = {};                              // ev_r[rset] has been recorded in this call
    // Pool + rows in one kernel: the pooling stream writes the sub-batch's new rows R straight from the tokens -- in this path
    // the frame means are consumed by the rows kernel only; same bits either way.  Default: pool_rows2_kernel (one short-lived
    // workgroup per (chunk, row), a wave per frame and slice): no kbar round trip through HBM, no rows kernel on the side
    // stream, and the pooling stream runs up to kRSets sub-batches ahead instead of three.  Measured in situ (round 3,
    // tools/sweep_r03s.sh, alternating on one box): 132.5 k against 130.5 k chunks/s, pooling stream 14.65 against 15.05 ms.
    // The experiments build keeps INFV_POOL_ROWS=0 (pool_frames_kernel + build_rows_kernel, the round-2 form, still used when
    // frame means are handed in or the width has no pool_rows2 shape).  (Round 3's long-lived grid-stride form, pool_rows_kernel,
    // streamed faster alone -- 218-270 us per launch against 304 -- but cost the chain launches their CUs, 116-121 k: deleted.)
    static const int pr_env = [] { const char* e = exp_env("INFV_POOL_ROWS"); return e ? atoi(e) : 2; }();
    const bool use_pr2 = pr_env == 2 && !kbar_pre && pool_rows2_supported(h->d);
    const bool use_pr = use_pr2;
    static const int pr_u = [] { const char* e = exp_env("INFV_PR_U"); return e ? atoi(e) : 8; }();
    static const int pr_pad = [] { const char* e = exp_env("INFV_PR_PAD"); return e ? atoi(e) : 84 * 1024; }();
    static const int pr_wgs = [] { const char* e = exp_env("INFV_PR_WGS"); return e ? atoi(e) : 0; }();
    if (!kbar_pre && !use_pr) {
        const size_t need = (size_t)h->maxC * T * h->d * sizeof(float);
        if (need > h->kbar_side[0].bytes) {
            HIP_TRY(hipDeviceSynchronize());
            for (int i = 0; i < kPSets; ++i) HIP_TRY(h->kbar_side[i].reserve(need));
        }
    }
    if (n_batches > 0) {                                      // the rotating R sets are written off the side stream: size them here
        const size_t needR = (size_t)sub * rows * h->d * sizeof(float);
        bool grow = false;
        for (int i = 0; i < kRSets; ++i) grow = grow || needR > h->R_ws[i].bytes;
        if (grow) {
            HIP_TRY(hipDeviceSynchronize());
            for (int i = 0; i < kRSets; ++i) HIP_TRY(h->R_ws[i].reserve(needR));
        }
    }
    // ---- ONE role-S launch for the whole call (round 5).  It goes out FIRST -- before any pooling, GEMM or UC launch of the call,
    // while its CUs are free: 24 workgroups per layer on one XCD each, where the step's exchange stays in that XCD's L2 -- and
    // stays resident to the call's last step.  Sub-batches reach it through `ready` (raised behind each projection GEMM), it
    // hands them on through `done` (flag_wait_kernel holds the UC stream).  Its ring slots and workspace sets need no wait of
    // their own: the GEMM of sub-batch b is itself ordered behind the UC kernel of sub-batch b - kPSets.
    const long call_slot0 = pipe.counter;                     // ring slot of the call-long launch's first step
    const int call_wgs = chain_batch_blocks(h->H, Q, h->L, 1, plan->sticky().points_ok, plan->inf.rows, h->S);   // role-S workgroups that count a sub-batch in
    hipStream_t call_stream = h->chain_s != nullptr ? h->chain_s : stream;      // where the call-long launch lives
    const bool use_call = persistent && uf != nullptr && chain_call_long() && n_batches > 0 && cu_mask_k() == 0 &&
                          !(sub_env <= 0 && n_chunks < 768 && ramp_env > 0) && kPSets <= kCallSets && !host_serial() && !(skip_mask() & 8);
    // ... and ONE pooling launch for the whole call: no launch boundaries on the HBM stream (with role S resident nothing but the
    // projection GEMM still needs an empty CU), rows and their bf16 planes for the whole call in HBM (402 + 604 MB at 2048 chunks),
    // the GEMM stream follows it through per-sub-batch completion counts
    // (mode 2: the one pooling launch WITHOUT the call-long role S, see above)
    // (the pooling kernel maps a chunk to its sub-batch as chunk / sub: every sub-batch but the last must have `sub` chunks -- the tapered
    //  schedules of the experiments build, short calls only, do not qualify)
    bool uniform_batches = true;
    for (int b = 0; b + 1 < n_batches; ++b) uniform_batches = uniform_batches && bstart[b + 1] - bstart[b] == sub;
    const bool pool_call_alone = pool_call_mode == 2 && persistent && n_batches > 0 && !host_serial() && uniform_batches;
    const bool use_pool_call = ((use_call && pool_call_mode != 0 && uniform_batches) || pool_call_alone) && use_pr2 && pr_wgs == 0 && !(skip_mask() & 1);
    const bool planes_call = use_pool_call && h->proj_x6 && h->d % 32 == 0 && !h->vproj_on_uc(n_chunks);
    // ... and ONE projection-GEMM launch: a few resident workgroups per XCD on a tile queue (gemm_x6_call_kernel).  Sub-batches it
    // covers: every one of >= 1024 rows (all but, possibly, a short last one: that one keeps its own launch behind the resident kernel)
    static const bool gemm_call_env = [] { const char* e = exp_env("INFV_GEMM_CALL"); return e && atoi(e) != 0; }();
    static const int gemm_wgs = [] { const char* e = exp_env("INFV_GEMM_WGS"); const int v = e ? atoi(e) : 32; return v > 0 ? v : 32; }();
    int n_tiled = 0;
    if (planes_call && gemm_call_env && h->w3_valid && !(skip_mask() & 2) && (sub * rows) % 32 == 0 && (long)sub * rows >= 1024 &&
        ((long)h->L * h->dm + (long)h->L * h->H * Q) % 256 == 0 && ((long)h->L * h->dm) % 256 == 0) {
        int c0l, nbl; batch_range(n_batches - 1, &c0l, &nbl);
        n_tiled = ((long)nbl * rows >= 1024 && ((long)nbl * rows) % 32 == 0) ? n_batches : n_batches - 1;
    }
    const bool use_gemm_call = n_tiled > 0;
    const int v_cols_all = h->L * h->dm;
    auto predict_split = [&](int nb, int* sk, long* ss) {       // the split-K form project_chunks_fast will choose for a sub-batch of nb chunks
        const long M = (long)nb * plan->inf.rows;
        const long ld = (long)h->L * h->dm + (long)h->L * h->H * Q;
        const bool defer = h->vproj_on_uc(n_chunks);
        if (skip_mask() & 2) *sk = project_splitk((int)M, h->d);
        else if (defer && M >= 1024) *sk = 1;
        else if (h->proj_x6 && h->w3_valid && M >= 1024 && h->d % 32 == 0 && !defer) *sk = 1;
        else *sk = project_splitk((int)M, h->d);
        *ss = M * ld;
    };
    if (use_call || use_pool_call) {
        // every workspace the loop would grow (a growth synchronises the device: fatal beside a kernel that waits for the loop's work)
        const long ld = (long)h->L * h->dm + (long)h->L * h->H * Q;
        size_t needP = 0;
        for (int b = 0; b < n_batches; ++b) {
            int c0, nb; batch_range(b, &c0, &nb);
            const size_t M = (size_t)nb * rows;
            const size_t np = M * ld * sizeof(float) * (M < 1024 ? 8 : 1);
            if (np > needP) needP = np;
        }
        const size_t need3 = (size_t)(h->maxC > sub ? h->maxC : sub) * rows * h->d * sizeof(__bf16);
        const size_t szW = (size_t)v_cols_all * h->d * 2, szR = (size_t)(h->maxC > sub ? h->maxC : sub) * rows * h->d * 2;
        bool grow = false;
        const size_t needRall = use_pool_call ? (size_t)(n_chunks - first_c) * rows * h->d * sizeof(float) : 0;
        const size_t needPl = planes_call ? needRall / 2 : 0;
        grow = grow || needRall > h->R_all.bytes || needPl > h->planes_all[0].bytes;
        for (int i = 0; i < kPSets; ++i) grow = grow || needP > h->P_ws[i].bytes;
        if (h->proj_x6) for (int i = 0; i < 3; ++i) grow = grow || need3 > h->r3[i].bytes;
        if (h->v_split) grow = grow || szW > h->wv_hi.bytes || szR > h->R_hi.bytes;
        if (grow) {
            HIP_TRY(hipDeviceSynchronize());
            HIP_TRY(h->R_all.reserve(needRall));
            for (int i = 0; i < 3; ++i) HIP_TRY(h->planes_all[i].reserve(needPl));
            for (int i = 0; i < kPSets; ++i) HIP_TRY(h->P_ws[i].reserve(needP));
            if (h->proj_x6) for (int i = 0; i < 3; ++i) HIP_TRY(h->r3[i].reserve(need3));
            if (h->v_split) {
                HIP_TRY(h->wv_hi.reserve(szW)); HIP_TRY(h->wv_lo.reserve(szW));
                HIP_TRY(h->R_hi.reserve(szR)); HIP_TRY(h->R_lo.reserve(szR));
                h->wv_split_valid = false;
            }
        }
    }
    // per-sub-batch pooling launches: the kernel writes the bf16 planes of its rows too (no split3_rows_kernel, no re-read of R).
    // Long calls only (round 6, same-box A/B, `profiles/r06_matrix.txt` block 10): a call of 768 chunks and more is bound by the role-S
    // stream and the planes cost the pooling nothing (150.4 against 149.0 k chunks/s, four rounds); a short call -- the 256-chunk
    // shard of the 8-GPU split -- is bound by its pooling launches, which the extra stores lengthen: 2.14-2.17 ms with
    // split3_rows_kernel on the side stream against 2.20-2.24 with the planes written by the pooling kernel.  Same bits either way.
    // (INFV_POOL_PLANES, experiments build: 0 = never, 2 = always.)
    static const int planes_env = [] { const char* e = exp_env("INFV_POOL_PLANES"); return e ? atoi(e) : 1; }();
    const bool planes_in_pool = (planes_env == 2 || (planes_env == 1 && n_chunks >= 768)) && use_pr2 && !use_pool_call && h->proj_x6 && h->w3_valid && h->d % 32 == 0 && !h->vproj_on_uc(n_chunks) && n_batches > 0;
    std::vector<char> planes_by_pool(n_batches > 0 ? n_batches : 1, 0);
    if (planes_in_pool) {
        const size_t need3r = (size_t)sub * rows * h->d * sizeof(__bf16);
        bool grow3 = false;
        for (int i = 0; i < kRSets; ++i) grow3 = grow3 || need3r > h->r3_ring[i][0].bytes;
        if (grow3) {
            HIP_TRY(hipDeviceSynchronize());
            for (int i = 0; i < kRSets; ++i) for (int j = 0; j < 3; ++j) HIP_TRY(h->r3_ring[i][j].reserve(need3r));
        }
    }
    auto stage_pool = [&](int b) -> int {                      // frame means (or directly the new rows) of batch b, on `pools`
        if (kbar_pre || use_pool_call) return INFV_OK;
        int c0, nb; batch_range(b, &c0, &nb);
        const int set = b % kPSets, rset = b % kRSets;
        if (use_pr) {
            // R set rset was last read by the UC kernel (and the projections) of batch b - kRSets
            if (r_pending[rset]) HIP_TRY(hipStreamWaitEvent(pools, h->ev_r[rset], 0));
            planes_by_pool[b] = false;
            if (!(skip_mask() & 1)) {
                // the pooling kernel also writes the rows' bf16 planes when the six-product GEMM will read them (>= 1024 rows)
                void* pl[3] = {h->r3_ring[rset][0].p, h->r3_ring[rset][1].p, h->r3_ring[rset][2].p};
                const bool want_planes = planes_in_pool && (long)nb * (long)rows >= 1024;
                bool done = false;
                Timed t_(h->prof, INFV_KERNEL_POOL, pools);
                HIP_TRY(launch_pool_rows2(k + c0 * chunk_k, h->k_bf16, nb, T, h->P, h->d, plan->inf.view(), h->R_ws[rset].as<float>(),
                                          pools, pr_u, pr_pad, pr_wgs, want_planes ? pl : nullptr, &done));
                planes_by_pool[b] = done;
            }
            if (split_pool) HIP_TRY(hipEventRecord(h->ev_pool[set], pools));
            return INFV_OK;
        }
        // the rows kernel that read this set's pooled frames (batch b-3) is done once its projection is
        if (split_pool && p_pending[set]) HIP_TRY(hipStreamWaitEvent(pools, h->ev_p[set], 0));
        if (!(skip_mask() & 1)) {
            Timed t_(h->prof, INFV_KERNEL_POOL, pools);
            HIP_TRY(launch_pool(k + c0 * chunk_k, h->k_bf16, h->kbar_side[set].as<float>(), (int64_t)nb * T, h->P, h->d, pools, kPoolPad));
        }
        if (split_pool) HIP_TRY(hipEventRecord(h->ev_pool[set], pools));
        return INFV_OK;
    };
    auto stage_project = [&](int b) -> int {                   // rows -> [V'new | S'new] GEMM of batch b, on `side`
        int c0, nb; batch_range(b, &c0, &nb);
        const int set = b % kPSets, rset = b % kRSets;
        if (b < n_tiled) {                                     // the resident GEMM kernel has this sub-batch on its tile queue
            sks[b] = 1; sss[b] = (long)nb * (long)rows * ((long)h->L * h->dm + (long)h->L * h->H * Q);
            return INFV_OK;
        }
        if (uc_pending[set]) HIP_TRY(hipStreamWaitEvent(side, h->ev_uc[set], 0));   // the UC kernel that read this set is done
        if (r_pending[rset] && !use_pr) HIP_TRY(hipStreamWaitEvent(side, h->ev_r[rset], 0));   // (the rows kernel writes R here)
        if (split_pool && !use_pool_call) HIP_TRY(hipStreamWaitEvent(side, h->ev_pool[set], 0));
        const float* kb = kbar_pre ? kbar_pre + (size_t)c0 * T * h->d : h->kbar_side[set].as<float>();
        const float* r_ext = nullptr;
        void* pl_ext[3] = {nullptr, nullptr, nullptr};
        if (!use_pool_call && planes_by_pool[b]) for (int i = 0; i < 3; ++i) pl_ext[i] = h->r3_ring[rset][i].p;
        if (use_pool_call) {
            // the rows of sub-batch b are complete once every (chunk, row) workgroup of the pooling launch has counted itself in
            HIP_TRY(launch_flag_wait(h->pool_done.as<unsigned int>() + b, (unsigned int)((size_t)nb * rows), h->spin_limit, h->err_dev, side));
            const size_t off = (size_t)(c0 - first_c) * rows * h->d;
            r_ext = h->R_all.as<float>() + off;
            if (planes_call) for (int i = 0; i < 3; ++i) pl_ext[i] = h->planes_all[i].as<__bf16>() + off;
        }
        // (experiments) INFV_SMALL_TILES=<mask>: bit 0 / 1 / 2 = the call's first / second / last sub-batch runs the projection as
        // 128 x 128 tiles (hundreds of short workgroups: faster while the chip is still -- or again -- empty)
        // Shipped (bit 3, round 6): the LAST sub-batch of a long call -- behind the pooling launch the call's tail is serial (GEMM -> role S ->
        // alpha -> UC of the last sub-batch) and the chip is emptying: 288 short workgroups finish sooner than 54 of 120 us.  13.45
        // against 13.63 ms per video over five alternating rounds on one box; the two tilings give the same bits (test-enforced).
        static const int small_mask = [] { const char* e = exp_env("INFV_SMALL_TILES"); return e ? atoi(e) : 8; }();
        static const int small_below = [] { const char* e = exp_env("INFV_SMALL_BELOW"); return e ? atoi(e) : 0; }();   // sub-batches of fewer chunks than this
        const bool small = (n_chunks < 768 && (((small_mask & 1) && b == 0) || ((small_mask & 2) && b == 1) || ((small_mask & 4) && b == n_batches - 1) || nb < small_below)) ||
                           ((small_mask & 8) && n_chunks >= 768 && b == n_batches - 1);      // bit 3: the LAST sub-batch of a long call (the chip is emptying: the serial tail)
        if (int rc = project_chunks_fast(h, *plan, true, kb, nb, T, Q, pp, set, &sks[b], &sss[b], side, kGemmPad,
                                         h->vproj_on_uc(n_chunks), rset, use_pr, r_ext, (planes_call || planes_by_pool[b]) ? pl_ext : nullptr, small)) return rc;
        HIP_TRY(hipEventRecord(h->ev_p[set], side));
        p_pending[set] = true;
        if (use_call) {
            int sk = 1; long ss = 0;
            predict_split(nb, &sk, &ss);
            if (sk != sks[b] || ss != sss[b]) return fail(INFV_ERR_STATE, "consolidate: sub-batch %d was projected in a split-K form the resident chain kernel was not told about", b);
            HIP_TRY(launch_flag_set(h->call_flags.as<unsigned int>(), (unsigned int)(b + 1), side));
        }
        return INFV_OK;
    };
    if (n_batches > 0) {
        HIP_TRY(hipEventRecord(h->ev_in, stream));            // inputs, cq and the first chunk's set are ordered before
        if (use_call) {
            int sk_main = 1, sk_last = 1; long ss_main = 0, ss_last = 0;
            int c0l, nbl; batch_range(n_batches - 1, &c0l, &nbl);
            predict_split(sub, &sk_main, &ss_main);
            predict_split(nbl, &sk_last, &ss_last);
            if (n_batches == 1) { sk_main = sk_last; ss_main = ss_last; }
            if (call_stream != stream) HIP_TRY(hipStreamWaitEvent(call_stream, h->ev_in, 0));
            pipe.stream = call_stream;
            const float* sets[kCallSets] = {};
            for (int i = 0; i < kPSets; ++i) sets[i] = h->P_ws[i].as<float>() + (size_t)v_cols_all;
            const int s_ct = (h->L * h->H * Q) / 256;                                   // S' column tiles of the projection
            const int rt_full = (int)(((long)sub * (long)rows + 383) / 384), rt_last = (int)(((long)nbl * (long)rows + 383) / 384);
            if (int rc = pipe.launch_s_call(n_chunks - first_c, sub, n_batches, sets, kPSets, sk_main, ss_main, sk_last, ss_last,
                                            u + (size_t)first_c * chunk_u, uf + (size_t)first_c * chunk_u,
                                            use_gemm_call ? h->gemm_flags.as<unsigned int>() + 128 : nullptr, n_tiled, s_ct * rt_full,
                                            s_ct * (n_tiled == n_batches ? rt_last : rt_full))) return rc;
            if (call_stream != stream) HIP_TRY(hipEventRecord(h->ev_chain, call_stream));
            pipe.stream = stream;
        }
        HIP_TRY(hipStreamWaitEvent(side, h->ev_q, 0));
        if (use_gemm_call) {
            GemmCallDesc gd;
            memset(&gd, 0, sizeof(gd));
            unsigned int* gf = h->gemm_flags.as<unsigned int>();
            for (int i = 0; i < 3; ++i) { gd.A_all[i] = h->planes_all[i].as<__bf16>(); gd.B[i] = h->w3[i].as<__bf16>(); }
            for (int i = 0; i < kPSets; ++i) gd.C_set[i] = h->P_ws[i].as<float>();
            gd.n_sets = kPSets;
            gd.ldc = (long)h->L * h->dm + (long)h->L * h->H * Q; gd.N = (int)gd.ldc; gd.K = h->d;
            gd.sub_rows = (int)(sub * rows);
            gd.total_rows = (n_tiled == n_batches) ? (long)(n_chunks - first_c) * (long)rows : (long)n_tiled * sub * (long)rows;
            gd.n_batches = n_tiled;
            gd.s_col_tile0 = (h->L * h->dm) / 256;
            gd.pool_done = h->pool_done.as<unsigned int>();
            gd.tile_ctr = gf; gd.uc_done = gf + 64; gd.done_s = gf + 128; gd.done_v = gf + 128 + h->gemm_flags_cap;
            gd.error = h->err_dev; gd.spin_limit = h->spin_limit;
            GemmCallDesc* gd_dev = reinterpret_cast<GemmCallDesc*>(gf + 128 + 2 * h->gemm_flags_cap);
            Timed t_(h->prof, INFV_KERNEL_PROJECT, side);
            HIP_TRY(launch_gemm_x6_call(gd, gd_dev, gemm_wgs, side));
        }
        if (split_pool) HIP_TRY(hipStreamWaitEvent(pools, h->ev_start, 0));
        if (use_pool_call) {
            PoolCallDesc pc;
            memset(&pc, 0, sizeof(pc));
            pc.sub = sub; pc.n_chunks = n_chunks - first_c;
            pc.R_all = h->R_all.as<float>();
            if (planes_call) for (int i = 0; i < 3; ++i) pc.plane[i] = h->planes_all[i].p;
            pc.done = h->pool_done.as<unsigned int>();
            Timed t_(h->prof, INFV_KERNEL_POOL, pools);
            HIP_TRY(launch_pool_rows2_call(k + first_c * chunk_k, h->k_bf16, T, h->P, h->d, plan->inf.view(), pc, pools, pr_u, pr_pad));
        }
        if (int rc = stage_pool(0)) return rc;
        if (n_batches > 1)
            if (int rc = stage_pool(1)) return rc;
        if (int rc = stage_project(0)) return rc;
    }
    auto stage_parallel = [&](int b) -> int {                  // issued while batch b-1's chain is about to start
        if (b + 1 < n_batches)
            if (int rc = stage_pool(b + 1)) return rc;
        return stage_project(b);
    };
    static const bool host_trace = exp_env("INFV_HOST_TRACE") != nullptr;   // host time of every loop iteration (is the host ahead of the device?)
    std::vector<double> host_us;
    const auto host_t0 = std::chrono::steady_clock::now();
    // (CU-mask experiment: role S's launches go to the stream that owns the reserved CUs; it starts behind everything the caller's
    //  stream has done so far and the caller's stream picks up behind it at the join)
    hipStream_t ls = stream;
    if (h->chain_s != nullptr && cu_mask_k() > 0 && persistent && n_batches > 0) {
        ls = h->chain_s;
        HIP_TRY(hipStreamWaitEvent(ls, h->ev_in, 0));
        pipe.stream = ls;
    }
    for (int b = 0; b < n_batches; ++b) {
        if (host_trace) host_us.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - host_t0).count());
        int c0, nb; batch_range(b, &c0, &nb);
        const int set = b % kPSets, rset = b % kRSets;
        // (experiments) INFV_DROP_WAITS=<mask>: what the packets between two role-S launches cost.  1: no wait for the UC kernel of
        // sub-batch b - 5 on the caller's stream (redundant: the GEMM of sub-batch b waited for it, and this stream waits for that GEMM);
        // 2: no event record behind role S / no wait of the UC stream for it (WRONG results: timing only); 4: no wait for the GEMM (WRONG)
        static const int drop_waits = [] { const char* e = exp_env("INFV_DROP_WAITS"); return e ? atoi(e) : 1; }();   // (round 6: bit 0 is the default -- 13.62 against 13.72 ms, tighter)
        if (!use_call) {
            if (!(drop_waits & 4)) HIP_TRY(hipStreamWaitEvent(ls, h->ev_p[set], 0));
            // the ring slots this batch writes were last read by the UC kernel three batches ago (same set)
            if (uc_pending[set] && !(drop_waits & 1)) HIP_TRY(hipStreamWaitEvent(ls, h->ev_uc[set], 0));
        }
        const long slot0 = use_call ? call_slot0 + (long)(c0 - first_c) : pipe.counter;
        const float* r_rows = use_pool_call ? h->R_all.as<float>() + (size_t)(c0 - first_c) * rows * h->d : h->R_ws[rset].as<float>();   // the sub-batch's new rows
        const bool serial = host_serial();                              // timing experiments: no overlap between the streams
        if (serial) { HIP_TRY(hipStreamSynchronize(pools)); HIP_TRY(hipStreamSynchronize(side)); HIP_TRY(hipStreamSynchronize(ucs)); }
        if (use_call) {
            // role S is resident: only the chunk-parallel stage of the next sub-batch goes out here
            if (b + 1 < n_batches)
                if (int rc = stage_parallel(b + 1)) return rc;
        } else if (persistent) {
            // the chunk-parallel stage of the next batch goes out first so it overlaps this batch's chain
            if (b + 1 < n_batches)
                if (int rc = stage_parallel(b + 1)) return rc;
            if (serial) { HIP_TRY(hipStreamSynchronize(pools)); HIP_TRY(hipStreamSynchronize(side)); }
            if (int rc = pipe.launch_s_batch(nb, h->P_ws[set].as<float>() + (size_t)h->L * h->dm, sks[b], sss[b],
                                             u ? u + (size_t)c0 * chunk_u : nullptr, uf ? uf + (size_t)c0 * chunk_u : nullptr)) return rc;
            if (serial) HIP_TRY(hipStreamSynchronize(ls));
        } else {
            for (int i = 0; i < nb; ++i) {
                const size_t ld = (size_t)h->L * h->dm + (size_t)h->L * h->H * Q;
                const StepS st{&plan->inf, true, h->P_ws[set].as<float>() + (size_t)i * rows * ld + (size_t)h->L * h->dm,
                               u ? u + (size_t)(c0 + i) * chunk_u : nullptr, sks[b], sss[b]};
                if (int rc = pipe.launch_s(st)) return rc;
                if (i == 0 && b + 1 < n_batches)
                    if (int rc = stage_parallel(b + 1)) return rc;
            }
        }
        // What the UC kernel needs besides role S's tables -- the softmax weights, and the V' half of the projection when it
        // is not part of the side stream's GEMM -- runs on the UC stream itself (a stream of its own for it was measured much
        // worse in rounds 1 and 2: every extra concurrent kernel slows role S and the GEMMs more than the shorter stream gains).
        hipStream_t vs = ucs;
        if (h->vproj_on_uc(n_chunks) && (long)nb * plan->inf.rows >= 1024 && !(skip_mask() & 2)) {
            // V' half of this sub-batch's projection: needs the new rows (ev_p), feeds only the UC kernel below
            HIP_TRY(hipStreamWaitEvent(vs, h->ev_p[set], 0));
            Timed t_(h->prof, INFV_KERNEL_PROJECT, vs);
            const long Mv = (long)nb * plan->inf.rows;
            const int v_cols = h->L * h->dm, p_ld = h->L * h->dm + h->L * h->H * Q;
            // exact fp32 MFMA by default.  INFV_VPROJ_SPLIT=1: V' only feeds the read-out (1e-3 budget), so it may run as a
            // split-bf16 contraction (three bf16 MFMA products, ~1e-5 relative); bench.py then labels its dtype accordingly
            if (h->v_split && h->d % 64 == 0 && v_cols % 128 == 0) {
                const size_t szW = (size_t)v_cols * h->d * 2, szR = (size_t)h->maxC * plan->inf.rows * h->d * 2;
                if (szW > h->wv_hi.bytes || szR > h->R_hi.bytes) {
                    HIP_TRY(hipDeviceSynchronize());
                    HIP_TRY(h->wv_hi.reserve(szW)); HIP_TRY(h->wv_lo.reserve(szW));
                    HIP_TRY(h->R_hi.reserve(szR)); HIP_TRY(h->R_lo.reserve(szR));
                    h->wv_split_valid = false;
                }
                if (!h->wv_split_valid) {
                    for (int l = 0; l < h->L; ++l)
                        HIP_TRY(launch_split_rows(pp.wv[l], h->d, h->dm, h->d, h->wv_hi.as<__bf16>() + (size_t)l * h->dm * h->d,
                                                  h->wv_lo.as<__bf16>() + (size_t)l * h->dm * h->d, h->d, vs));
                    h->wv_split_valid = true;
                }
                HIP_TRY(launch_split_rows(r_rows, h->d, Mv, h->d, h->R_hi.p, h->R_lo.p, h->d, vs));
                SplitGemm g{};
                g.A_hi = h->R_hi.as<__bf16>(); g.A_lo = h->R_lo.as<__bf16>(); g.lda = h->d; g.strideA = 0;
                g.B_hi = h->wv_hi.as<__bf16>(); g.B_lo = h->wv_lo.as<__bf16>(); g.ldb = h->d; g.strideB = 0;
                g.C = h->P_ws[set].as<float>(); g.ldc = p_ld; g.strideC = 0; g.split_stride = 0;
                g.M = (int)Mv; g.N = v_cols; g.K = h->d; g.k_per_split = h->d; g.splitk = 1; g.nbatch = 1;
                HIP_TRY(launch_split_gemm(g, vs, kGemmPad));
            } else {
                HIP_TRY(launch_project_values((int)Mv, h->d, h->dm, h->L, pp, r_rows,
                                              h->P_ws[set].as<float>(), p_ld, vs, kGemmPad));
            }
        }
        if (use_call) {
            // the UC stream holds until every role-S workgroup has written sub-batch b's steps back (the counter was zeroed on the
            // caller's stream: the first wait of a call is ordered behind that)
            if (b == 0) HIP_TRY(hipStreamWaitEvent(ucs, h->ev_in, 0));
            HIP_TRY(launch_flag_wait(h->call_flags.as<unsigned int>() + 64, (unsigned int)(b + 1) * (unsigned int)call_wgs, h->spin_limit, h->err_dev, ucs));
            if (b < n_tiled) {
                const int rt_b = (int)(((long)nb * (long)rows + 383) / 384);
                HIP_TRY(launch_flag_wait(h->gemm_flags.as<unsigned int>() + 128 + h->gemm_flags_cap + b, (unsigned int)(rt_b * (v_cols_all / 256)),
                                         h->spin_limit, h->err_dev, ucs));
            }
            pipe.last_snew = h->P_ws[set].as<float>() + (size_t)v_cols_all; pipe.last_sk = sks[b]; pipe.last_ss = sss[b];
        } else if (!(drop_waits & 2)) {
            HIP_TRY(hipEventRecord(h->ev_s[set], ls));
            HIP_TRY(hipStreamWaitEvent(ucs, h->ev_s[set], 0));
        }
        if (persistent)
            if (int rc = pipe.launch_alpha(nb, slot0, vs, b == n_batches - 1)) return rc;
        if (int rc = pipe.launch_uc(plan->inf, true, nb, slot0, r_rows, h->P_ws[set].as<float>(),
                                    sks[b], sss[b], ctx + (size_t)c0 * chunk_ctx, ucs)) return rc;
        if (use_gemm_call) HIP_TRY(launch_flag_set(h->gemm_flags.as<unsigned int>() + 64, (unsigned int)(b + 1), ucs));   // the output set of sub-batch b is free again
        HIP_TRY(hipEventRecord(h->ev_uc[set], ucs));
        uc_pending[set] = true;
        HIP_TRY(hipEventRecord(h->ev_r[rset], ucs));
        r_pending[rset] = true;
    }
    if (host_trace && !host_us.empty()) {
        fprintf(stderr, "[host trace] %d iterations, issue time us:", n_batches);
        for (size_t i = 0; i < host_us.size(); i += 4) fprintf(stderr, " %.0f", host_us[i]);
        fprintf(stderr, " | end %.0f\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - host_t0).count());
    }
    // join: the memory and every ctx are complete once the last UC kernel is; then hand the sticky histogram
    // back as one float partial row and bring the K' half of the projected memory up to date
    if (ls != stream) {
        HIP_TRY(hipEventRecord(h->ev_in, ls));
        HIP_TRY(hipStreamWaitEvent(stream, h->ev_in, 0));
        pipe.stream = stream;
    }
    if (use_call && call_stream != stream) HIP_TRY(hipStreamWaitEvent(stream, h->ev_chain, 0));
    for (int i = 0; i < kPSets; ++i)
        if (uc_pending[i]) HIP_TRY(hipStreamWaitEvent(stream, h->ev_uc[i], 0));
    if (pipe.counter > 0) {
        if (pipe.last_v2 && pipe.batch_launches > 0 && chain_batch3_mailboxes())   // (experiments) the last step's totals are in the mailboxes
            HIP_TRY(launch_mailbox_to_part(h->mbox.as<unsigned long long>(), h->L, h->mbox_G, (int)((pipe.counter - 1) & 1), 1,
                                           h->bin_part[h->pc].as<float>(), stream));
        else
            HIP_TRY(launch_acc_to_part(h->mass_acc[(pipe.counter + 2) % 3].as<unsigned long long>(), h->L, 1,
                                       h->bin_part[h->pc].as<float>(), stream));
        h->parts = 1;
    }
    h->k_stale = true;                                        // K' is re-projected from B on demand (per-call path, continuation)
    return INFV_OK;
}

int infv_ltm_consolidate(infv_ltm_handle h, const void* k, int32_t n_chunks, int32_t T, const float* q,
                         int32_t Q, const infv_ltm_proj* proj, const double* u, int32_t new_doc, float* ctx,
                         void* stream) {
    if (!k) return fail(INFV_ERR_INVALID, "consolidate: bad arguments");
    return consolidate_impl(h, k, nullptr, n_chunks, T, q, Q, proj, u, new_doc, ctx, stream);
}

int infv_ltm_consolidate_pooled(infv_ltm_handle h, const float* kbar, int32_t n_chunks, int32_t T, const float* q,
                                int32_t Q, const infv_ltm_proj* proj, const double* u, int32_t new_doc, float* ctx,
                                void* stream) {
    if (!kbar) return fail(INFV_ERR_INVALID, "consolidate_pooled: bad arguments");
    return consolidate_impl(h, nullptr, kbar, n_chunks, T, q, Q, proj, u, new_doc, ctx, stream);
}

}  // extern "C"

extern "C" {

int infv_ltm_export_state(infv_ltm_handle h, int32_t layer, float* B, float* bin_mass, void* stream_) {
    if (int rc = check_handle(h)) return rc;
    if (layer < 0 || layer >= h->L) return fail(INFV_ERR_INVALID, "layer out of range");
    if (int rc = check_chain_error(h)) return rc;
    if (!h->has_memory) return fail(INFV_ERR_STATE, "no memory to export (B_past is None)");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (B) HIP_TRY(hipMemcpyAsync(B, h->B[h->cur].as<float>() + (size_t)layer * h->N * h->d,
                                  (size_t)h->N * h->d * sizeof(float), hipMemcpyDeviceToDevice, stream));
    if (bin_mass) {
        HIP_TRY(launch_sum_parts(h->bin_part[h->pc].as<float>() + (size_t)layer * h->parts * h->n_bins, h->parts, h->n_bins,
                                 bin_mass, stream));
    }
    return INFV_OK;
}

int64_t infv_ltm_chain_state_bytes(infv_ltm_handle h, int32_t Q) {
    if (!h || Q <= 0 || Q > h->maxQ) return -1;
    return (int64_t)sizeof(float) * (kBlobHeaderInts + (int64_t)h->L * h->N * h->d + (int64_t)h->L * h->N * 2 * h->dm +
                                     (int64_t)h->L * h->H * Q * h->N + (int64_t)h->L * h->n_bins);
}

namespace {
BlobHeader blob_header(infv_ltm_handle h, int Q) {
    BlobHeader hd{};
    const int v[] = {kBlobMagic, kBlobVersion, h->L, h->N, h->d, h->dm, h->H, Q, h->n_bins};
    for (size_t i = 0; i < sizeof(v) / sizeof(v[0]); ++i) hd.v[i] = v[i];
    return hd;
}
}  // namespace

int infv_ltm_export_chain_state(infv_ltm_handle h, int32_t Q, void* blob, void* stream_) {
    if (int rc = check_handle(h)) return rc;
    if (!blob) return fail(INFV_ERR_INVALID, "export_chain_state: null blob");
    if (int rc = check_q(h, Q)) return rc;
    if (int rc = check_chain_error(h)) return rc;
    if (!h->has_memory || !h->last_fast || h->lastQ != Q || h->parts != 1)
        return fail(INFV_ERR_STATE, "export_chain_state: the memory's last step must come from infv_ltm_consolidate with Q=%d", Q);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    INFV_LAUNCH(blob_header_write_kernel, dim3(1), dim3(64), 0, stream, static_cast<int*>(blob), blob_header(h, Q));
    HIP_TRY(hipGetLastError());
    float* out = static_cast<float*>(blob) + kBlobHeaderInts;
    const size_t nB = (size_t)h->L * h->N * h->d, nKV = (size_t)h->L * h->N * 2 * h->dm, nS = (size_t)h->L * h->H * Q * h->N,
                 nM = (size_t)h->L * h->n_bins;
    HIP_TRY(hipMemcpyAsync(out, h->B[h->cur].p, nB * sizeof(float), hipMemcpyDeviceToDevice, stream));
    HIP_TRY(hipMemcpyAsync(out + nB, h->KV[h->cur].p, nKV * sizeof(float), hipMemcpyDeviceToDevice, stream));
    HIP_TRY(hipMemcpyAsync(out + nB + nKV, h->Sp[h->sc].p, nS * sizeof(float), hipMemcpyDeviceToDevice, stream));
    HIP_TRY(hipMemcpyAsync(out + nB + nKV + nS, h->bin_part[h->pc].p, nM * sizeof(float), hipMemcpyDeviceToDevice, stream));
    return INFV_OK;
}

int infv_ltm_import_chain_state(infv_ltm_handle h, int32_t Q, const void* blob, void* stream_) {
    if (int rc = check_handle(h)) return rc;
    if (!blob) return fail(INFV_ERR_INVALID, "import_chain_state: null blob");
    if (int rc = check_q(h, Q)) return rc;
    if (int rc = check_chain_error(h)) return rc;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    // the header is checked on the device (the call stays asynchronous): a mismatch latches an error that the next entry point reports
    INFV_LAUNCH(blob_header_check_kernel, dim3(1), dim3(1), 0, stream, static_cast<const int*>(blob), blob_header(h, Q), h->err_dev);
    HIP_TRY(hipGetLastError());
    const float* in = static_cast<const float*>(blob) + kBlobHeaderInts;
    const size_t nB = (size_t)h->L * h->N * h->d, nKV = (size_t)h->L * h->N * 2 * h->dm, nS = (size_t)h->L * h->H * Q * h->N,
                 nM = (size_t)h->L * h->n_bins;
    HIP_TRY(hipMemcpyAsync(h->B[h->cur].p, in, nB * sizeof(float), hipMemcpyDeviceToDevice, stream));
    HIP_TRY(hipMemcpyAsync(h->KV[h->cur].p, in + nB, nKV * sizeof(float), hipMemcpyDeviceToDevice, stream));
    HIP_TRY(hipMemcpyAsync(h->Sp[h->sc].p, in + nB + nKV, nS * sizeof(float), hipMemcpyDeviceToDevice, stream));
    HIP_TRY(hipMemcpyAsync(h->bin_part[h->pc].p, in + nB + nKV + nS, nM * sizeof(float), hipMemcpyDeviceToDevice, stream));
    h->has_memory = true; h->parts = 1; h->lastQ = Q; h->last_fast = true;
    h->k_stale = true;                 // the K' half came from a fast-path handle: re-projected on demand
    h->carry_scores = true;
    return INFV_OK;
}

int infv_ltm_reproject(infv_ltm_handle h, const infv_ltm_proj* proj, void* stream_) {
    if (int rc = check_handle(h)) return rc;
    if (!proj) return fail(INFV_ERR_INVALID, "reproject: null proj");
    if (!h->has_memory) return INFV_OK;
    const ProjPtrs pp = make_proj(proj, h->L);
    HIP_TRY(launch_reproject(h->B[h->cur].as<float>(), h->N, h->d, h->dm, h->L, pp, h->KV[h->cur].as<float>(),
                             static_cast<hipStream_t>(stream_)));
    h->k_stale = false;
    return INFV_OK;
}

int infv_ltm_import_state(infv_ltm_handle h, int32_t layer, const float* B, const float* bin_mass,
                          const infv_ltm_proj* proj, void* stream_) {
    if (int rc = check_handle(h)) return rc;
    if (layer < 0 || layer >= h->L) return fail(INFV_ERR_INVALID, "layer out of range");
    if (!B || !proj) return fail(INFV_ERR_INVALID, "import: null argument");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    HIP_TRY(hipMemcpyAsync(h->B[h->cur].as<float>() + (size_t)layer * h->N * h->d, B,
                           (size_t)h->N * h->d * sizeof(float), hipMemcpyDeviceToDevice, stream));
    if (h->parts <= 0) h->parts = 1;
    float* part = h->bin_part[h->pc].as<float>() + (size_t)layer * h->parts * h->n_bins;
    HIP_TRY(hipMemsetAsync(part, 0, (size_t)h->parts * h->n_bins * sizeof(float), stream));
    if (bin_mass)
        HIP_TRY(hipMemcpyAsync(part, bin_mass, (size_t)(h->n_bins - 1) * sizeof(float), hipMemcpyDeviceToDevice, stream));
    // re-project this layer's memory with its weights
    ProjPtrs one;
    memset(&one, 0, sizeof(one));
    one.wk[0] = proj->wk; one.bk[0] = proj->bk; one.wv[0] = proj->wv; one.bv[0] = proj->bv;
    HIP_TRY(launch_reproject(h->B[h->cur].as<float>() + (size_t)layer * h->N * h->d, h->N, h->d, h->dm, 1, one,
                             h->KV[h->cur].as<float>() + (size_t)layer * h->N * 2 * h->dm, stream));
    h->has_memory = true;
    return INFV_OK;
}

int infv_ltm_get_draw(infv_ltm_handle h, int32_t layer, int32_t* bins, int32_t* idx, float* probs,
                      float* scores, void* stream_) {
    if (int rc = check_handle(h)) return rc;
    if (layer < 0 || layer >= h->L) return fail(INFV_ERR_INVALID, "layer out of range");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    HIP_TRY(hipStreamSynchronize(stream));
    if (int rc = check_chain_error(h)) return rc;
    if (bins) HIP_TRY(hipMemcpy(bins, h->bins.as<int32_t>() + (size_t)layer * h->S, h->S * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (idx) HIP_TRY(hipMemcpy(idx, h->idx.as<int32_t>() + (size_t)layer * h->S, h->S * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (probs) HIP_TRY(hipMemcpy(probs, h->probs.as<float>() + (size_t)layer * h->n_bins, (h->n_bins - 1) * sizeof(float), hipMemcpyDeviceToHost));
    if (scores) {
        if (h->lastQ <= 0) return fail(INFV_ERR_STATE, "no scores yet");
        const size_t n = (size_t)h->H * h->lastQ * h->N;
        if (h->last_fast) {
            std::vector<float> cq((size_t)h->H * h->lastQ);
            HIP_TRY(hipMemcpy(scores, h->Sp[h->sc].as<float>() + (size_t)layer * n, n * sizeof(float), hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(cq.data(), h->cqbuf.as<float>() + (size_t)layer * cq.size(), cq.size() * sizeof(float), hipMemcpyDeviceToHost));
            for (size_t r = 0; r < cq.size(); ++r)
                for (int j = 0; j < h->N; ++j) scores[r * h->N + j] += cq[r];
        } else {
            HIP_TRY(hipMemcpy(scores, h->scores.as<float>() + (size_t)layer * n, n * sizeof(float), hipMemcpyDeviceToHost));
        }
    }
    return INFV_OK;
}

int infv_ltm_set_probs(infv_ltm_handle h, int32_t layer, const float* probs) {
    if (int rc = check_handle(h)) return rc;
    if (layer < 0 || layer >= h->L || !probs) return fail(INFV_ERR_INVALID, "set_probs: bad arguments");
    HIP_TRY(hipMemcpy(h->probs_override.as<float>() + (size_t)layer * h->n_bins, probs,
                      (h->n_bins - 1) * sizeof(float), hipMemcpyHostToDevice));
    h->override_mask |= 1u << layer;
    return INFV_OK;
}

int infv_ltm_set_bins(infv_ltm_handle h, int32_t layer, const int32_t* bins) {
    if (int rc = check_handle(h)) return rc;
    if (layer < 0 || layer >= h->L || !bins) return fail(INFV_ERR_INVALID, "set_bins: bad arguments");
    for (int s = 0; s < h->S; ++s)
        if (bins[s] < 0 || bins[s] >= h->n_bins - 1) return fail(INFV_ERR_INVALID, "set_bins: bins[%d]=%d outside [0,%d)", s, bins[s], h->n_bins - 1);
    HIP_TRY(hipMemcpy(h->bins_forced.as<int32_t>() + (size_t)layer * h->S, bins, h->S * sizeof(int32_t), hipMemcpyHostToDevice));
    h->forced_mask |= 1u << layer;
    return INFV_OK;
}

int infv_ltm_set_trace(infv_ltm_handle h, int32_t* bins_all, float* probs_all, int64_t capacity_chunks) {
    if (int rc = check_handle(h)) return rc;
    if (capacity_chunks < 0) return fail(INFV_ERR_INVALID, "set_trace: negative capacity");
    h->trace_bins = capacity_chunks ? bins_all : nullptr;
    h->trace_probs = capacity_chunks ? probs_all : nullptr;
    h->trace_cap = (h->trace_bins || h->trace_probs) ? (long)capacity_chunks : 0;
    return INFV_OK;
}

int infv_ltm_sync(infv_ltm_handle h, void* stream) {
    if (int rc = check_handle(h)) return rc;
    HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return check_chain_error(h);
}

int infv_ltm_profile_enable(infv_ltm_handle h, int32_t on) {
    if (int rc = check_handle(h)) return rc;
    h->prof.on = on != 0;
    if (!on) h->prof.clear_all();
    return INFV_OK;
}

int infv_ltm_profile_read(infv_ltm_handle h, int32_t kernel, int64_t* launches, double* total_ms) {
    if (int rc = check_handle(h)) return rc;
    if (kernel < 0 || kernel >= INFV_KERNEL_COUNT || !launches || !total_ms) return fail(INFV_ERR_INVALID, "profile_read: bad arguments");
    double ms = 0.0;
    for (auto& p : h->prof.ev[kernel]) {
        HIP_TRY(hipEventSynchronize(p.second));
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, p.first, p.second));
        ms += t;
    }
    *launches = (int64_t)h->prof.ev[kernel].size();
    *total_ms = ms;
    h->prof.clear(kernel);
    return INFV_OK;
}

}  // extern "C"
