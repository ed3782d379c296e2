// Whole-video fast path of the LTM memory chain (used by infv_ltm_consolidate, where the query
// of every layer is the same for all chunks).
//
// The chain  B_c = f(B_{c-1}, S_{c-1}, kbar_c, u_c)  is sequential across chunks, so its cost is
// latency, not bandwidth.  Two restatements take the matrix products off the critical path:
//
//  (1) score recurrence.  K'_c[n] = val_n * sum_{s in slots(n)} K'_{c-1}[idx_s] + P_c[row(n)]
//      (projection is linear), hence with a fixed query
//          S'_c[q][n] = val_n * sum_s S'_{c-1}[q][idx_s] + S'new_c[q][row(n)],
//      S'new_c = (q/sqrt(dh)) . P_c^T  is batched over chunks ahead of time (new_scores_kernel).
//      The chain step then needs no GEMM for the scores (reference :224-230), only a gather.
//  (2) deferred read-out.  ctx_{c} = alpha_c . (V'_c + bv) is not an input of step c+1, so launch
//      c+1 computes it ("role C") beside step c+1's critical work.
//
//  (3) deferred state update.  B_c / V'_c are not inputs of step c+1's scores either, so they are
//      gathered one launch later from the indices step c published ("role U"), and the read-out
//      follows one launch after that.
//
// One launch per chunk, three kinds of workgroups that never talk to each other inside a launch,
// each working on a different chunk of a 3-stage software pipeline (launch k):
//   role S  (head, 8-row q-tile, layer): draw_k -> score recurrence -> alpha_k, sticky partials_k  [critical]
//   role U  (4 boxes, layer)           : B_{k-1}, V'_{k-1} rows from idx_{k-1} (gather + new rows)
//   role C  (head, 16-row q-tile, layer): ctx_{k-2} from alpha_{k-2}, V'_{k-2}
// Every S workgroup repeats the (tiny) Gibbs draw so that no in-launch hand-off is needed.
#include "ltm_device.h"

namespace infv {

// ======================================================================================
// S'new[c][l][h][q][r] = sum_e q[l][q][h*64+e]/sqrt(dh) * Kmat(c, r, l)[h*64+e]
//   Kmat rows: base + c*chunk_stride + r*row_stride + l*layer_stride (+ k*split_stride, summed)
//   cq[l][h][q] = q_h[q] . bk_h / sqrt(dh)   (written by the chunk-0 workgroups if cq != null)
// ======================================================================================
__global__ __launch_bounds__(256) void new_scores_kernel(const float* __restrict__ q, int Q, int H, int rows,
                                                         const float* __restrict__ Kmat, long chunk_stride,
                                                         long row_stride, long layer_stride, int splitk,
                                                         long split_stride, ProjPtrs proj,
                                                         float* __restrict__ Snew, float* __restrict__ cq) {
    const int h = blockIdx.x, l = blockIdx.y, ch = blockIdx.z;
    const int L = gridDim.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int dm = H * kHeadSize;
    const int QT = (Q + kQTile - 1) / kQTile, RT = (rows + 15) / 16;
    const float scale = 1.0f / sqrtf((float)kHeadSize);
    const float* ql = q + (long)l * Q * dm + h * kHeadSize + 16 * g;
    const float* Kb = Kmat + ch * chunk_stride + l * layer_stride + h * kHeadSize + 16 * g;
    float* out = Snew + (((long)ch * L + l) * H + h) * Q * rows;
    for (int ti = wave; ti < QT * RT; ti += 4) {
        const int qt = ti / RT, rt = ti - qt * RT;
        float qa[16];
        const int qrow_a = qt * kQTile + c;
        if (qrow_a < Q) {
            const floatx4* src = reinterpret_cast<const floatx4*>(ql + (long)qrow_a * dm);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const floatx4 t = src[v];
                qa[4 * v] = t.x * scale; qa[4 * v + 1] = t.y * scale; qa[4 * v + 2] = t.z * scale; qa[4 * v + 3] = t.w * scale;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) qa[j] = 0.f;
        }
        floatx4 kb[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        const int krow = rt * 16 + c;
        if (krow < rows) {
            const float* src = Kb + (long)krow * row_stride;
            for (int k = 0; k < splitk; ++k) {
                const floatx4* s4 = reinterpret_cast<const floatx4*>(src + k * split_stride);
#pragma unroll
                for (int v = 0; v < 4; ++v) kb[v] += s4[v];
            }
        }
        floatx4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 16; ++j)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[j], kb[j >> 2][j & 3], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qrow = qt * kQTile + 4 * g + r;
            if (qrow < Q && krow < rows) __builtin_nontemporal_store(acc[r], &out[(long)qrow * rows + krow]);
        }
        if (cq != nullptr && ch == 0 && rt == 0) {
            const float* bk = proj.bk[l] + h * kHeadSize;
            float part = 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) part = fmaf(qa[j], bk[16 * g + j], part);
            part += __shfl_xor(part, 16);
            part += __shfl_xor(part, 32);
            if (g == 0 && qrow_a < Q) cq[((long)l * H + h) * Q + qrow_a] = part;
        }
    }
}

hipError_t launch_new_scores(const float* q, int Q, int H, int n_layers, int n_chunks, int rows, const float* Kmat,
                             long chunk_stride, long row_stride, long layer_stride, int splitk, long split_stride,
                             const ProjPtrs& proj, float* Snew, float* cq, hipStream_t stream) {
    if (rows == 0 || n_chunks == 0) return hipSuccess;
    hipLaunchKernelGGL(new_scores_kernel, dim3(H, n_layers, n_chunks), dim3(256), 0, stream, q, Q, H, rows, Kmat,
                       chunk_stride, row_stride, layer_stride, splitk, split_stride, proj, Snew, cq);
    return hipGetLastError();
}

// ======================================================================================
// the chain kernel: 512-thread workgroups; at the headline shape 96 + 128 + 48 of them, about one per CU.
// Every role issues ALL of its global reads that do not depend on an earlier read first ("prologue").
// ======================================================================================
constexpr int kNT = 512;                 // 8 waves = 2 per SIMD
constexpr int kRowsS = 8;                // query rows per role-S workgroup: wave w <-> row w
constexpr int kBoxesPerU = 4;
constexpr int kMaxN = 256;               // boxes the fast path holds in LDS
constexpr int kMaxTabw = 16;             // slots per box the dense table holds
constexpr int kNIter = kMaxN / 64;       // boxes per lane in a wave-per-row sweep
constexpr int kCRows = 64;               // V' rows staged per read-out pass of role C (keeps the launch's LDS small)

struct ChainSmem {            // role S: offsets (in floats) into dynamic LDS, identical on host and device
    int cdf, sidx, gsum, misc, tab, box_val, box_row, w, bin_box, edge_box, edge_dx, Sprev, Ssm, Snew, Dsm, Msm, total;
};

__host__ __device__ inline ChainSmem chain_smem(int N, int S, int rows, int tabw) {
    ChainSmem m;
    int o = 0;
    auto take = [&](int n) { int r = o; o += (n + 3) & ~3; return r; };
    m.cdf = take(kBins);
    m.sidx = take(S);
    m.gsum = take(2 * kNT);                            // kNT doubles
    m.misc = take(64);                                 // [0,16) cq | [16,32) asum
    m.tab = take(N * tabw);
    m.box_val = take(N);
    m.box_row = take(N);
    m.w = take(N);
    m.bin_box = take(kBins);
    m.edge_box = take(kBins + 4);
    m.edge_dx = take(kBins);
    m.Sprev = take(kRowsS * (N + 4));
    m.Ssm = take(kRowsS * (N + 2));
    m.Snew = take(kRowsS * (rows + 1));
    m.Dsm = take(kRowsS * kDPitch);
    m.Msm = take(kRowsS * kMPitch);
    m.total = o;
    return m;
}

#define STAMP(slot) do { if (a.dbg != nullptr && stamp_me) { a.dbg[slot] = wall_clock64(); if ((slot) == 0 || (slot) == 5) a.dbg[24 + (slot)] = clock64(); } } while (0)

__global__ __launch_bounds__(kNT) void chain_kernel(ChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = a.N, H = a.H, Q = a.Q, QT = a.QT, QS = a.QS;
    const int dm = H * kHeadSize;
    int b = blockIdx.x;
    if (a.debug_noop) return;                                          // dispatch-floor timing experiment
    // latency-critical: win issue arbitration against the throughput kernels of the side stream that share the CU
    __builtin_amdgcn_s_setprio(3);

    if (b < a.s.n_blocks) {
        // ================================================================== role S: wave w <-> query row w
        const ChainRoleS& rs = a.s;
        const int rows = rs.op.rows, tabw = rs.op.tabw;
        const ChainSmem m = chain_smem(N, a.S, rows, tabw);
        const int h = b % H, qs = (b / H) % QS, l = b / (H * QS);
        const int sp = N + 4, sstride = N + 2, sn = rows + 1;
        float* Sprev = lds + m.Sprev;
        float* Ssm = lds + m.Ssm;
        float* Snew = lds + m.Snew;
        float* cqs = lds + m.misc;
        float* asum = lds + m.misc + 16;
        int32_t* sidx = reinterpret_cast<int32_t*>(lds + m.sidx);
        int32_t* tab = reinterpret_cast<int32_t*>(lds + m.tab);
        const long tile = (((long)l * H + h) * Q + qs * kRowsS);       // first row of this tile in [L][H][Q][*] arrays
        const int valid = min(kRowsS, Q - qs * kRowsS);
        const bool writer = (h == 0 && qs == 0);
        const bool stamp_me = (b == 0 && tid == 0);
        STAMP(0);
        // ---- prologue: every global read of this workgroup, back to back ----
        const int n4 = N / 4;
        floatx4 sp_reg = {0.f, 0.f, 0.f, 0.f};
        const int sr = tid / n4, sc4 = tid - sr * n4;                  // N <= 256: at most one float4 per thread
        if (rs.draw_mode != 0 && sr < valid) sp_reg = *reinterpret_cast<const floatx4*>(rs.Sp_prev + (tile + sr) * N + sc4 * 4);
        float sn_reg[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + i * kNT;
            const int r = e / rows, cc = e - r * rows;
            sn_reg[i] = (e < kRowsS * rows && r < valid) ? rs.Snew[(tile + r) * rows + cc] : 0.f;
        }
        const float cq_reg = (tid < valid) ? rs.cq[tile + tid] : 0.f;
        const float t_box_val = (tid < N) ? rs.op.box_val[tid] : 0.f;
        const int t_box_row = (tid < N) ? rs.op.box_row[tid] : -1;
        const float t_w = (tid < N) ? rs.w[tid] : 0.f;
        const int t_bin_box = (tid < kBins) ? a.st.bin_box[tid] : -1;
        const float t_edge_dx = (tid < kBins) ? a.st.edge_dx[tid] : 0.f;
        const int t_edge_box = (tid <= kBins) ? a.st.edge_box[tid] : -1;
        int t_slot[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = tid + i * kNT;
            t_slot[i] = (rs.draw_mode != 0 && e < N * tabw) ? rs.op.slot_tab[e] : -1;
        }
        DrawRegs<1> dr;
        if (rs.draw_mode == 1)
            dr = draw_load<kNT, 1>(rs.part_prev + (long)l * rs.parts * kBins, rs.parts,
                                   rs.acc_prev ? rs.acc_prev + l * kBins : nullptr, rs.probs_override + l * kBins,
                                   (rs.override_mask >> l) & 1u, rs.u + (long)l * a.S, a.S);
        if (writer && tid < kBins) rs.acc_clear[l * kBins + tid] = 0ull;   // ring slot of the NEXT launch: idle now
        int uni = -1;
        if (rs.draw_mode == 2 && tid < a.S) uni = rs.uniform_idx[tid];
        // ---- park the prologue in LDS ----
        if (sr < kRowsS) *reinterpret_cast<floatx4*>(&Sprev[sr * sp + sc4 * 4]) = sp_reg;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + i * kNT;
            if (e < kRowsS * rows) { const int r = e / rows; Snew[r * sn + (e - r * rows)] = sn_reg[i]; }
        }
        if (tid < kRowsS) cqs[tid] = cq_reg;
        if (tid < N) {
            (lds + m.box_val)[tid] = t_box_val;
            reinterpret_cast<int32_t*>(lds + m.box_row)[tid] = t_box_row;
            (lds + m.w)[tid] = t_w;
        }
        if (tid < kBins) {
            reinterpret_cast<int32_t*>(lds + m.bin_box)[tid] = t_bin_box;
            (lds + m.edge_dx)[tid] = t_edge_dx;
        }
        if (tid <= kBins) reinterpret_cast<int32_t*>(lds + m.edge_box)[tid] = t_edge_box;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = tid + i * kNT;
            if (e < N * tabw) tab[e] = t_slot[i];
        }
        if (rs.draw_mode == 2 && tid < a.S) {
            sidx[tid] = uni;
            if (writer) rs.idx_out[(long)l * a.S + tid] = uni;
        }
        __syncthreads();
        STAMP(1);
        // ---- draw, then tab[n][k] := resampled source box of the k-th slot of box n ----
        if (rs.draw_mode == 1)
            draw_finish<kNT, 1>(dr, (rs.override_mask >> l) & 1u, reinterpret_cast<const int32_t*>(lds + m.bin_box),
                                a.S, lds + m.cdf, sidx, reinterpret_cast<double*>(lds + m.gsum),
                                writer ? rs.probs_out + l * kBins : nullptr,
                                writer ? rs.bins_out + (long)l * a.S : nullptr,
                                writer ? rs.idx_out + (long)l * a.S : nullptr, (b == 0) ? a.dbg : nullptr);
        if (rs.draw_mode != 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int e = tid + i * kNT;
                if (e < N * tabw) {
                    const int sl = tab[e];
                    const int src = (sl >= 0) ? sidx[sl] : -1;
                    tab[e] = src;
                    if (writer) rs.tab_out[(long)l * N * tabw + e] = src;
                }
            }
            __syncthreads();
        }
        STAMP(2);
        // ---- score recurrence (gathers are two independent LDS reads deep) ----
        {
            const float* box_val = lds + m.box_val;
            const int32_t* box_row = reinterpret_cast<const int32_t*>(lds + m.box_row);
            const int row = wave;
            const float cqr = cqs[row];
            const float* Sp = Sprev + row * sp;
#pragma unroll
            for (int i = 0; i < kNIter; ++i) {
                const int n = lane + 64 * i;
                if (n < N) {
                    float acc = 0.f;
                    if (rs.draw_mode != 0) {
                        const float val = box_val[n];
                        for (int k0 = 0; k0 < tabw; k0 += 4) {
                            const int4 src = *reinterpret_cast<const int4*>(&tab[n * tabw + k0]);
                            const float v0 = Sp[max(src.x, 0)], v1 = Sp[max(src.y, 0)];
                            const float v2 = Sp[max(src.z, 0)], v3 = Sp[max(src.w, 0)];
                            if (src.x >= 0) acc = fmaf(val, v0, acc);
                            if (src.y >= 0) acc = fmaf(val, v1, acc);
                            if (src.z >= 0) acc = fmaf(val, v2, acc);
                            if (src.w >= 0) acc = fmaf(val, v3, acc);
                        }
                    }
                    const int r = box_row[n];
                    if (r >= 0) acc += Snew[row * sn + r];
                    Ssm[row * sstride + n] = acc + cqr;
                    if (row < valid) rs.Sp_next[(tile + row) * N + n] = acc;
                }
            }
        }
        __syncthreads();
        STAMP(3);
        row_phase_wave(Ssm, sstride, N, valid, lds + m.w, rs.w_out, reinterpret_cast<const int32_t*>(lds + m.edge_box),
                       lds + m.edge_dx, lds + m.Dsm, lds + m.Msm, asum, nullptr, rs.acc_next + l * kBins, kRowsS);
        STAMP(4);
        // alpha_k and its row sums for role C two launches later
        if (wave < valid) {
#pragma unroll
            for (int i = 0; i < kNIter; ++i) {
                const int n = lane + 64 * i;
                if (n < N) rs.alpha_out[(tile + wave) * N + n] = Ssm[wave * sstride + n];
            }
            if (lane == 0) rs.asum_out[tile + wave] = asum[wave];
        }
        STAMP(5);
        return;
    }
    b -= a.s.n_blocks;
    if (b < a.u.n_blocks) {
        // ================================================================== role U: B and V' rows of 4 boxes
        const ChainRoleU& ru = a.u;
        const int tabw = ru.op.tabw;
        const int per_layer = (N + kBoxesPerU - 1) / kBoxesPerU;
        const int l = b / per_layer, n0 = (b - l * per_layer) * kBoxesPerU;
        const int d4 = a.d4, dm4 = a.dm4, kv4 = 2 * dm4;
        const int total4 = d4 + dm4;
        const int nbox = min(kBoxesPerU, N - n0);
        const floatx4* R4 = reinterpret_cast<const floatx4*>(ru.R);
        const floatx4* P4 = reinterpret_cast<const floatx4*>(ru.Pnew);
        int32_t* src_lds = reinterpret_cast<int32_t*>(lds);             // [kBoxesPerU][tabw] resolved source boxes
        float* val_lds = lds + kBoxesPerU * kMaxTabw;                   // [kBoxesPerU]
        const bool stamp_me = (b == 0 && tid == 0);
        STAMP(8);
        // round trip 1: resolved source boxes of this workgroup's boxes, plus the new-row contributions
        if (tid < nbox * tabw)
            src_lds[tid] = ru.gather ? ru.tab[((long)l * N + n0) * tabw + tid] : -1;
        if (tid < nbox) val_lds[tid] = ru.op.box_val[n0 + tid];
        floatx4 newv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            newv[i] = floatx4{0.f, 0.f, 0.f, 0.f};
            const int o = tid + i * kNT;
            if (o < nbox * total4) {
                const int bi = o / total4, c = o - bi * total4;
                const int r = ru.op.box_row[n0 + bi];
                if (r >= 0) {
                    if (c < d4) {
                        newv[i] = R4[(long)r * d4 + c];
                    } else {
                        const long off = ((long)r * a.L + l) * kv4 + dm4 + (c - d4);
                        for (int k = 0; k < ru.splitk; ++k) newv[i] += P4[off + k * ru.split_stride4];
                    }
                }
            }
        }
        __syncthreads();
        STAMP(9);
        // round trip 2: the gathered rows
        const floatx4* Bp = reinterpret_cast<const floatx4*>(ru.B_prev) + (long)l * N * d4;
        const floatx4* Vp = reinterpret_cast<const floatx4*>(ru.KV_prev) + (long)l * N * kv4 + dm4;
        floatx4* Bn = reinterpret_cast<floatx4*>(ru.B_next) + (long)l * N * d4;
        floatx4* Vn = reinterpret_cast<floatx4*>(ru.KV_next) + (long)l * N * kv4 + dm4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int o = tid + i * kNT;
            if (o < nbox * total4) {
                const int bi = o / total4, c = o - bi * total4;
                const int n = n0 + bi;
                const bool isB = c < d4;
                const int cc = isB ? c : c - d4;
                const floatx4* prev = isB ? Bp : Vp;
                const int pitch = isB ? d4 : kv4;
                floatx4 acc = {0.f, 0.f, 0.f, 0.f};
                if (ru.gather) {
                    const float val = val_lds[bi];
                    for (int k0 = 0; k0 < tabw; k0 += 4) {             // 4 gathered rows in flight at a time
                        const int4 src = *reinterpret_cast<const int4*>(&src_lds[bi * tabw + k0]);
                        const int sv[4] = {src.x, src.y, src.z, src.w};
                        floatx4 v[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[k] = prev[(long)max(sv[k], 0) * pitch + cc];
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (sv[k] >= 0) {
                                acc.x = fmaf(val, v[k].x, acc.x); acc.y = fmaf(val, v[k].y, acc.y);
                                acc.z = fmaf(val, v[k].z, acc.z); acc.w = fmaf(val, v[k].w, acc.w);
                            }
                    }
                }
                acc += newv[i];
                if (isB) Bn[(long)n * d4 + cc] = acc; else Vn[(long)n * kv4 + cc] = acc;
            }
        }
        STAMP(11);
        return;
    }
    b -= a.u.n_blocks;
    {
        // ================================================================== role C: read-out of one (head, 16-row tile)
        // 8 waves = 4 column tiles (16 of the head's 64 columns) x 2 halves of the box dimension;
        // the 2 partial accumulators of a column tile are summed through LDS in a fixed order.
        const ChainRoleC& rc = a.c;
        const int h = b % H, qt = (b / H) % QT, l = b / (H * QT);
        const int sstride = N + 2;
        float* Asm = lds;                                              // [16][N+2] alpha
        float* Vsm = lds + ((kQTile * sstride + 3) & ~3);              // [kCRows][80]
        float* red = Vsm + kCRows * kVStride;                          // [8 waves][64 lanes][4]
        const long tile = (((long)l * H + h) * Q + qt * kQTile);
        const int valid = min(kQTile, Q - qt * kQTile);
        const float* Vhead = rc.KV + (long)l * N * 2 * dm + dm + h * kHeadSize;
        const bool stamp_me = (b == 0 && tid == 0);
        STAMP(16);
        // prologue: alpha tile and the first kVRows rows of the V' head slice together
        const int n4 = N / 4;
        floatx4 al[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = tid + i * kNT;
            const int ar = e / n4, ac4 = e - ar * n4;
            al[i] = (ar < valid) ? *reinterpret_cast<const floatx4*>(rc.alpha + (tile + ar) * N + ac4 * 4)
                                 : floatx4{0.f, 0.f, 0.f, 0.f};
        }
        const int c = lane & 15, g = lane >> 4;
        const int ct = wave & 3, ks = wave >> 2;                       // column tile, half of the staged rows
        floatx4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        for (int base = 0; base < N; base += kCRows) {
            const int rows = min(kCRows, N - base);
            floatx4 vreg[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int e = tid + i * kNT;
                const int r = e >> 4, c4 = e & 15;
                vreg[i] = (r < rows) ? *reinterpret_cast<const floatx4*>(Vhead + (long)(base + r) * 2 * dm + c4 * 4)
                                     : floatx4{0.f, 0.f, 0.f, 0.f};
            }
            if (base == 0) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int e = tid + i * kNT;
                    const int ar = e / n4, ac4 = e - ar * n4;
                    if (ar < kQTile) {
                        float* dst = &Asm[ar * sstride + ac4 * 4];
                        dst[0] = al[i].x; dst[1] = al[i].y; dst[2] = al[i].z; dst[3] = al[i].w;
                    }
                }
            } else {
                __syncthreads();                                       // previous pass's reads of Vsm are done
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int e = tid + i * kNT;
                const int r = e >> 4, c4 = e & 15;
                if (r < rows) *reinterpret_cast<floatx4*>(&Vsm[r * kVStride + c4 * 4]) = vreg[i];
            }
            __syncthreads();
            if (base == 0) STAMP(17);
            const int per = rows / 2;                                  // boxes of this wave's half (multiple of 8)
            const int kb = ks * per;
#pragma unroll 4
            for (int t = 0; t < per / 4; t += 2) {
                const float a0 = Asm[c * sstride + base + kb + 4 * t + g];
                const float b0 = Vsm[(kb + 4 * t + g) * kVStride + 16 * ct + c];
                const float a1 = Asm[c * sstride + base + kb + 4 * (t + 1) + g];
                const float b1 = Vsm[(kb + 4 * (t + 1) + g) * kVStride + 16 * ct + c];
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc1, 0, 0, 0);
            }
        }
        const floatx4 accw = acc0 + acc1;
        *reinterpret_cast<floatx4*>(&red[(wave * 64 + lane) * 4]) = accw;
        __syncthreads();
        STAMP(18);
        if (ks == 0) {
            floatx4 tot = accw;
            tot += *reinterpret_cast<const floatx4*>(&red[((4 + ct) * 64 + lane) * 4]);
            const float* bv = rc.bv[l] + h * kHeadSize;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rr = 4 * g + r;
                if (rr < valid) {
                    const int col = 16 * ct + c;
                    rc.ctx_out[((long)l * Q + qt * kQTile + rr) * dm + h * kHeadSize + col] =
                        tot[r] + rc.asum[tile + rr] * bv[col];
                }
            }
        }
        STAMP(19);
    }
}

size_t chain_lds_bytes(int N, int S, int rows, int tabw) {
    const size_t roleS = (size_t)chain_smem(N, S, rows, tabw).total;
    const size_t roleC = (size_t)((kQTile * (N + 2) + 3) & ~3) + kCRows * kVStride + 8 * 64 * 4;
    return (roleS > roleC ? roleS : roleC) * sizeof(float);
}

bool chain_supported(int N, int S, int rows_max, int tabw) {
    return N <= kMaxN && N % 16 == 0 && S <= kNT && rows_max <= kMaxN && tabw <= kMaxTabw && (tabw & 3) == 0 &&
           N * tabw <= 8 * kNT && chain_lds_bytes(N, S, rows_max, tabw) <= 160 * 1024;
}

int chain_s_tiles(int Q) { return (Q + kRowsS - 1) / kRowsS; }

int chain_u_blocks(int N, int n_layers) { return n_layers * ((N + kBoxesPerU - 1) / kBoxesPerU); }

hipError_t launch_chain(const ChainArgs& a, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(chain_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int blocks = a.s.n_blocks + a.u.n_blocks + a.c.n_blocks;
    if (blocks == 0) return hipSuccess;
    int rows = 1, tabw = 4;
    if (a.s.n_blocks) { rows = a.s.op.rows; tabw = a.s.op.tabw; }
    if (a.u.n_blocks && a.u.op.tabw > kMaxTabw) return hipErrorInvalidValue;
    if (!chain_supported(a.N, a.S, rows, tabw)) return hipErrorInvalidValue;
    const size_t lds = chain_lds_bytes(a.N, a.S, rows, tabw);
    hipLaunchKernelGGL(chain_kernel, dim3(blocks), dim3(kNT), lds, stream, a);
    return hipGetLastError();
}

}  // namespace infv

namespace infv {
__global__ void acc_to_part_kernel(const unsigned long long* __restrict__ acc, int parts_pitch, float* __restrict__ part) {
    const int l = blockIdx.x, j = threadIdx.x;
    if (j < kBins) part[((long)l * parts_pitch) * kBins + j] = (j < kBins - 1) ? (float)((double)acc[l * kBins + j] * (1.0 / kMassScale)) : 0.f;
}
hipError_t launch_acc_to_part(const unsigned long long* acc, int n_layers, int parts_pitch, float* part, hipStream_t stream) {
    hipLaunchKernelGGL(acc_to_part_kernel, dim3(n_layers), dim3(128), 0, stream, acc, parts_pitch, part);
    return hipGetLastError();
}
}  // namespace infv
