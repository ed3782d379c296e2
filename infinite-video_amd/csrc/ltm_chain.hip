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
// One launch per chunk, three kinds of workgroups that never talk to each other inside a launch:
//   role S  (head, q-tile, layer): draw -> score recurrence -> alpha_c, sticky partials      [critical]
//   role U  (8 boxes, layer)     : draw -> B_c and V'_c rows (gather + new rows)              [state]
//   role C  (head, q-tile, layer): ctx_{c-1} from alpha_{c-1}, V'_{c-1}                       [deferred]
// Every S/U workgroup repeats the (tiny) Gibbs draw so that no inter-workgroup hand-off is needed.
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
    const int QT = (Q + kQTile - 1) / kQTile, RT = rows / 16;
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
        const float* src = Kb + (long)(rt * 16 + c) * row_stride;
        for (int k = 0; k < splitk; ++k) {
            const floatx4* s4 = reinterpret_cast<const floatx4*>(src + k * split_stride);
#pragma unroll
            for (int v = 0; v < 4; ++v) kb[v] += s4[v];
        }
        floatx4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 16; ++j)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[j], kb[j >> 2][j & 3], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qrow = qt * kQTile + 4 * g + r;
            if (qrow < Q) out[(long)qrow * rows + rt * 16 + c] = acc[r];
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
// the chain kernel
// ======================================================================================
constexpr int kBoxesPerU = 8;

struct ChainSmem {            // offsets (in floats) into dynamic LDS, computed identically on host and device
    int Sprev, Ssm, Snew, old_ptr, old_slot, box_val, box_row, cdf, sidx, Dsm, Msm, misc, total;
};

__host__ __device__ inline ChainSmem chain_smem(int N, int S, int rows_max) {
    ChainSmem m;
    int o = 0;
    auto take = [&](int n) { int r = o; o += (n + 3) & ~3; return r; };
    m.cdf = take(kBins);
    m.sidx = take(S);
    m.old_ptr = take(N + 1);
    m.old_slot = take(S);
    m.box_val = take(N);
    m.box_row = take(N);
    m.misc = take(64);
    m.Sprev = take(kQTile * (N + 4));
    m.Ssm = take(kQTile * (N + 2));
    m.Snew = take(kQTile * (rows_max + 1));
    m.Dsm = take(kQTile * kDPitch);
    m.Msm = take(kQTile * kMPitch);
    m.total = o;
    return m;
}

__device__ inline void load_csr(const ChainArgs& a, float* lds, const ChainSmem& m) {
    const int tid = threadIdx.x;
    int32_t* old_ptr = reinterpret_cast<int32_t*>(lds + m.old_ptr);
    int32_t* old_slot = reinterpret_cast<int32_t*>(lds + m.old_slot);
    float* box_val = lds + m.box_val;
    int32_t* box_row = reinterpret_cast<int32_t*>(lds + m.box_row);
    for (int i = tid; i < a.N; i += 256) { box_val[i] = a.op.box_val[i]; box_row[i] = a.op.box_row[i]; }
    if (a.op.old_ptr != nullptr) {
        for (int i = tid; i <= a.N; i += 256) old_ptr[i] = a.op.old_ptr[i];
        const int nnz = a.op.old_ptr[a.N];
        for (int i = tid; i < nnz; i += 256) old_slot[i] = a.op.old_slot[i];
    } else {
        for (int i = tid; i <= a.N; i += 256) old_ptr[i] = 0;
    }
}

// draw (or uniform resample) of layer l into sidx (LDS); `writer` also publishes the diagnostics.
__device__ inline void chain_draw(const ChainArgs& a, int l, float* lds, const ChainSmem& m, bool writer) {
    int32_t* sidx = reinterpret_cast<int32_t*>(lds + m.sidx);
    if (a.draw_mode == 1) {
        double* scratch = reinterpret_cast<double*>(lds + m.misc);          // 4 doubles
        float* total = lds + m.misc + 8;
        draw_core(a.part_prev + (long)l * a.parts * kBins, a.parts, a.probs_override + l * kBins,
                  (a.override_mask >> l) & 1u, a.st, a.u + (long)l * a.S, a.S, lds + m.cdf, sidx, scratch, total,
                  writer ? a.probs_out + l * kBins : nullptr, writer ? a.bins_out + (long)l * a.S : nullptr,
                  writer ? a.idx_out + (long)l * a.S : nullptr);
    } else if (a.draw_mode == 2) {
        for (int s = threadIdx.x; s < a.S; s += 256) sidx[s] = a.uniform_idx[s];
        __syncthreads();
    } else {
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void chain_kernel(ChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int N = a.N, H = a.H, Q = a.Q, QT = a.QT;
    const int dm = H * kHeadSize;
    int b = blockIdx.x;

    if (b < a.nS) {
        // ------------------------------------------------------------------ role S
        const ChainSmem m = chain_smem(N, a.S, a.rows_max);
        const int h = b % H, qt = (b / H) % QT, l = b / (H * QT);
        const int sp = N + 4, sstride = N + 2, sn = a.rows + 1;
        float* Sprev = lds + m.Sprev;
        float* Ssm = lds + m.Ssm;
        float* Snew = lds + m.Snew;
        float* cqs = lds + m.misc + 16;
        float* asum = lds + m.misc + 32;
        const long tile = (((long)l * H + h) * Q + qt * kQTile);       // first row of this tile in [L][H][Q][*] arrays
        const int valid = min(kQTile, Q - qt * kQTile);
        // stage previous bias-free scores, this chunk's new-row scores, the operator tables
        if (a.draw_mode != 0) {
            const int n4 = N / 4;
            for (int i = tid; i < kQTile * n4; i += 256) {
                const int r = i / n4, c4 = i - r * n4;
                floatx4 v = {0.f, 0.f, 0.f, 0.f};
                if (r < valid) v = *reinterpret_cast<const floatx4*>(a.Sp_prev + (tile + r) * N + c4 * 4);
                *reinterpret_cast<floatx4*>(&Sprev[r * sp + c4 * 4]) = v;
            }
        }
        for (int i = tid; i < kQTile * a.rows; i += 256) {
            const int r = i / a.rows, cc = i - r * a.rows;
            Snew[r * sn + cc] = (r < valid) ? a.Snew[(tile + r) * a.rows + cc] : 0.f;
        }
        if (tid < kQTile) cqs[tid] = (tid < valid) ? a.cq[tile + tid] : 0.f;
        load_csr(a, lds, m);
        chain_draw(a, l, lds, m, h == 0 && qt == 0);                  // ends with a barrier
        // score recurrence: 16 threads per row
        {
            const int32_t* old_ptr = reinterpret_cast<const int32_t*>(lds + m.old_ptr);
            const int32_t* old_slot = reinterpret_cast<const int32_t*>(lds + m.old_slot);
            const int32_t* sidx = reinterpret_cast<const int32_t*>(lds + m.sidx);
            const float* box_val = lds + m.box_val;
            const int32_t* box_row = reinterpret_cast<const int32_t*>(lds + m.box_row);
            const int row = tid >> 4, sub = tid & 15;
            const float cqr = cqs[row];
            for (int n = sub; n < N; n += 16) {
                float acc = 0.f;
                if (a.draw_mode != 0) {
                    const float val = box_val[n];
                    for (int s = old_ptr[n]; s < old_ptr[n + 1]; ++s) {
                        const int src = sidx[old_slot[s]];
                        if (src >= 0) acc = fmaf(val, Sprev[row * sp + src], acc);
                    }
                }
                const int r = box_row[n];
                if (r >= 0) acc += Snew[row * sn + r];
                Ssm[row * sstride + n] = acc + cqr;
                if (row < valid) {
                    a.Sp_next[(tile + row) * N + n] = acc;
                    if (a.scores_out != nullptr) a.scores_out[(tile + row) * N + n] = acc + cqr;
                }
            }
        }
        __syncthreads();
        row_phase(Ssm, sstride, N, valid, a.w, a.w_out, a.st, lds + m.Dsm, lds + m.Msm, asum,
                  a.part_next + (((long)l * H + h) * QT + qt) * kBins);
        // alpha_c and its row sums for the next launch's role C
        {
            const int row = tid >> 4, sub = tid & 15;
            if (row < valid) {
                for (int n = sub; n < N; n += 16) a.alpha_next[(tile + row) * N + n] = Ssm[row * sstride + n];
                if (sub == 0) a.asum_next[tile + row] = asum[row];
            }
        }
        return;
    }
    b -= a.nS;
    if (b < a.nU) {
        // ------------------------------------------------------------------ role U
        const ChainSmem m = chain_smem(N, a.S, a.rows_max);
        const int per_layer = (N + kBoxesPerU - 1) / kBoxesPerU;
        const int l = b / per_layer, n0 = (b - l * per_layer) * kBoxesPerU;
        load_csr(a, lds, m);
        chain_draw(a, l, lds, m, false);
        const int32_t* old_ptr = reinterpret_cast<const int32_t*>(lds + m.old_ptr);
        const int32_t* old_slot = reinterpret_cast<const int32_t*>(lds + m.old_slot);
        const int32_t* sidx = reinterpret_cast<const int32_t*>(lds + m.sidx);
        const float* box_val = lds + m.box_val;
        const int32_t* box_row = reinterpret_cast<const int32_t*>(lds + m.box_row);
        const int d4 = a.d4, dm4 = a.dm4, kv4 = 2 * dm4;
        const floatx4* Bp = reinterpret_cast<const floatx4*>(a.B_prev) + (long)l * N * d4;
        const floatx4* Vp = reinterpret_cast<const floatx4*>(a.KV_prev) + (long)l * N * kv4 + dm4;
        floatx4* Bn = reinterpret_cast<floatx4*>(a.B_next) + (long)l * N * d4;
        floatx4* Vn = reinterpret_cast<floatx4*>(a.KV_next) + (long)l * N * kv4 + dm4;
        const floatx4* R4 = reinterpret_cast<const floatx4*>(a.R);
        const floatx4* P4 = reinterpret_cast<const floatx4*>(a.Pnew);
        const int total4 = d4 + dm4;
        for (int c = tid; c < total4; c += 256) {
            const bool isB = c < d4;
            const int cc = isB ? c : c - d4;
            const floatx4* prev = isB ? Bp : Vp;
            const int pitch = isB ? d4 : kv4;
#pragma unroll 2
            for (int bi = 0; bi < kBoxesPerU; ++bi) {
                const int n = n0 + bi;
                if (n >= N) break;
                floatx4 acc = {0.f, 0.f, 0.f, 0.f};
                if (a.draw_mode != 0) {
                    const float val = box_val[n];
                    for (int s = old_ptr[n]; s < old_ptr[n + 1]; ++s) {
                        const int src = sidx[old_slot[s]];
                        if (src >= 0) {
                            const floatx4 v = prev[(long)src * pitch + cc];
                            acc.x = fmaf(val, v.x, acc.x); acc.y = fmaf(val, v.y, acc.y);
                            acc.z = fmaf(val, v.z, acc.z); acc.w = fmaf(val, v.w, acc.w);
                        }
                    }
                }
                const int r = box_row[n];
                if (r >= 0) {
                    if (isB) {
                        acc += R4[(long)r * d4 + cc];
                    } else {
                        const long off = ((long)r * a.L + l) * kv4 + dm4 + cc;
                        for (int k = 0; k < a.splitk; ++k) acc += P4[off + k * a.split_stride4];
                    }
                }
                if (isB) Bn[(long)n * d4 + cc] = acc; else Vn[(long)n * kv4 + cc] = acc;
            }
        }
        return;
    }
    b -= a.nU;
    {
        // ------------------------------------------------------------------ role C (previous chunk)
        const int h = b % H, qt = (b / H) % QT, l = b / (H * QT);
        const int sstride = N + 2;
        float* Asm = lds;
        float* Vsm = lds + ((kQTile * sstride + 3) & ~3);
        const long tile = (((long)l * H + h) * Q + qt * kQTile);
        const int valid = min(kQTile, Q - qt * kQTile);
        const int row = tid >> 4, sub = tid & 15;
        for (int n = sub; n < N; n += 16)
            Asm[row * sstride + n] = (row < valid) ? a.alpha_cur[(tile + row) * N + n] : 0.f;
        const floatx4 acc = readout_tile(Asm, sstride, N, a.KV_prev + (long)l * N * 2 * dm + dm + h * kHeadSize,
                                         2L * dm, Vsm);
        const int lane = tid & 63, wave = tid >> 6, c = lane & 15, g = lane >> 4;
        const float* bv = a.bv[l] + h * kHeadSize;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rr = 4 * g + r;
            if (rr < valid) {
                const int col = 16 * wave + c;
                a.ctx_out[((long)l * Q + qt * kQTile + rr) * dm + h * kHeadSize + col] =
                    acc[r] + a.asum_cur[tile + rr] * bv[col];
            }
        }
    }
}

size_t chain_lds_bytes(int N, int S, int rows_max) {
    const size_t roleSU = (size_t)chain_smem(N, S, rows_max).total;
    const size_t roleC = (size_t)((kQTile * (N + 2) + 3) & ~3) + kVRows * kVStride;
    return (roleSU > roleC ? roleSU : roleC) * sizeof(float);
}

int chain_u_blocks(int N, int n_layers) { return n_layers * ((N + kBoxesPerU - 1) / kBoxesPerU); }

hipError_t launch_chain(const ChainArgs& a, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(chain_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int blocks = a.nS + a.nU + a.nC;
    if (blocks == 0) return hipSuccess;
    const size_t lds = chain_lds_bytes(a.N, a.S, a.rows_max);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(chain_kernel, dim3(blocks), dim3(256), lds, stream, a);
    return hipGetLastError();
}

}  // namespace infv
