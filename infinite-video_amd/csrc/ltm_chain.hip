// Whole-video fast path of the LTM memory chain, part 1: the sequential "role S" of one chunk per launch
// (used for the first chunk of a document, and as the fallback when the persistent sub-batch kernel of
// ltm_chain_batch.hip does not apply), plus the batched new-row scores.
//
// The chain  B_c = f(B_{c-1}, S_{c-1}, kbar_c, u_c)  is sequential across chunks, so its cost is
// latency, not bandwidth.  Three restatements take everything but a gather off the critical path
// (DESIGN.md section 2):
//
//  (1) score recurrence.  K'_c[n] = val_n * sum_{s in slots(n)} K'_{c-1}[idx_s] + P_c[row(n)]
//      (projection is linear), hence with a fixed query
//          S'_c[q][n] = val_n * sum_s S'_{c-1}[q][idx_s] + S'new_c[q][row(n)],
//      S'new_c = (q/sqrt(dh)) . P_c^T  is batched over chunks ahead of time (new_scores_kernel).
//  (2) deferred state update and read-out.  B_c, V'_c and ctx_c are not inputs of step c+1's draw;
//      role S only publishes the resolved gather table and the softmax weights, and the UC kernel
//      (ltm_uc.hip) applies them to a whole sub-batch later, slice-parallel.
//
// Role S of chunk k (one workgroup per (head, 8-row query tile, layer)):
//   draw_k -> score recurrence -> alpha_k, sticky bin masses_k (fixed-point atomics), gather table_k.
// Every workgroup repeats the (tiny) Gibbs draw so that no in-launch hand-off is needed.
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
    INFV_LAUNCH(new_scores_kernel, dim3(H, n_layers, n_chunks), dim3(256), 0, stream, q, Q, H, rows, Kmat,
                       chunk_stride, row_stride, layer_stride, splitk, split_stride, proj, Snew, cq);
    return hipGetLastError();
}

// ======================================================================================
// the chain kernel: 512-thread workgroups; at the headline shape 96 + 128 + 48 of them, about one per CU.
// Every role issues ALL of its global reads that do not depend on an earlier read first ("prologue").
// ======================================================================================
constexpr int kNT = 512;                 // 8 waves = 2 per SIMD
constexpr int kRowsS = 8;                // query rows per role-S workgroup: wave w <-> row w
constexpr int kMaxN = 256;               // boxes the fast path holds in LDS
constexpr int kMaxTabw = 16;             // slots per box the dense table holds
constexpr int kNIter = kMaxN / 64;       // boxes per lane in a wave-per-row sweep

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
            const int qr = e & (kRowsS - 1), nr = e >> 3;               // kRowsS == 8
            float v = 0.f;
            if (e < kRowsS * rows && qr < valid)
                for (int k = 0; k < rs.snew_splitk; ++k) v += rs.Snew[(long)nr * rs.snew_ld + tile + qr + k * rs.snew_split_stride];
            sn_reg[i] = v;
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
                                   rs.acc_prev ? rs.acc_prev + (long)l * kAccShards * kBins : nullptr, rs.probs_override + l * kBins,
                                   (rs.override_mask >> l) & 1u, rs.u + (long)l * a.S, a.S);
        if (writer)                                                                  // ring slot of the NEXT launch: idle now
            for (int e = tid; e < kAccShards * kBins; e += kNT) rs.acc_clear[(long)l * kAccShards * kBins + e] = 0ull;
        int uni = -1;
        if (rs.draw_mode == 2 && tid < a.S) uni = rs.uniform_idx[tid];
        // ---- park the prologue in LDS ----
        if (sr < kRowsS) *reinterpret_cast<floatx4*>(&Sprev[sr * sp + sc4 * 4]) = sp_reg;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + i * kNT;
            if (e < kRowsS * rows) Snew[(e & (kRowsS - 1)) * sn + (e >> 3)] = sn_reg[i];
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
                                writer ? rs.idx_out + (long)l * a.S : nullptr, (b == 0) ? a.dbg : nullptr, nullptr,
                                (writer && rs.probs_tr) ? rs.probs_tr + l * kBins : nullptr,
                                (writer && rs.bins_tr) ? rs.bins_tr + (long)l * a.S : nullptr);
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
                       lds + m.edge_dx, lds + m.Dsm, lds + m.Msm, asum, nullptr, rs.acc_next + (long)l * kAccShards * kBins, kRowsS);
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
}

size_t chain_lds_bytes(int N, int S, int rows, int tabw) {
    return (size_t)chain_smem(N, S, rows, tabw).total * sizeof(float);
}

bool chain_supported(int N, int S, int rows_max, int tabw) {
    return N <= kMaxN && N % 16 == 0 && S <= kNT && rows_max <= kMaxN && tabw <= kMaxTabw && (tabw & 3) == 0 &&
           N * tabw <= 8 * kNT && chain_lds_bytes(N, S, rows_max, tabw) <= 160 * 1024;
}

int chain_s_tiles(int Q) { return (Q + kRowsS - 1) / kRowsS; }

hipError_t launch_chain(const ChainArgs& a, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(chain_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int blocks = a.s.n_blocks;
    if (blocks == 0) return hipSuccess;
    const int rows = a.s.op.rows, tabw = a.s.op.tabw;
    if (!chain_supported(a.N, a.S, rows, tabw)) return hipErrorInvalidValue;
    const size_t lds = chain_lds_bytes(a.N, a.S, rows, tabw);
    INFV_LAUNCH(chain_kernel, dim3(blocks), dim3(kNT), lds, stream, a);
    return hipGetLastError();
}

}  // namespace infv

namespace infv {
__global__ void acc_to_part_kernel(const unsigned long long* __restrict__ acc, int parts_pitch, float* __restrict__ part) {
    const int l = blockIdx.x, j = threadIdx.x;
    if (j >= kBins) return;
    unsigned long long tot = 0ull;                                   // integer sum over the replicas: the exact total
    for (int r = 0; r < kAccShards; ++r) tot += acc[((long)l * kAccShards + r) * kBins + acc_word(j)] & kMassMask;
    part[((long)l * parts_pitch) * kBins + j] = (j < kBins - 1) ? (float)mass_of(tot) : 0.f;
}
hipError_t launch_acc_to_part(const unsigned long long* acc, int n_layers, int parts_pitch, float* part, hipStream_t stream) {
    INFV_LAUNCH(acc_to_part_kernel, dim3(n_layers), dim3(128), 0, stream, acc, parts_pitch, part);
    return hipGetLastError();
}
}  // namespace infv
