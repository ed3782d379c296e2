// Role S of a whole sub-batch of chunks in ONE launch (persistent over the sub-batch).
//
// The per-chunk launch of chain_kernel pays, per step, a kernel boundary (~3.5 us of dispatch + cache
// write-back) and a cold reload of its state.  Here the H*QS*L workgroups of role S stay resident for all
// chunks of a sub-batch and keep their score tile, tables and gather machinery in LDS; the only
// inter-workgroup dependency of a step -- the 127 sticky bin masses summed over the (head, query) rows of a
// layer -- is exchanged in place:
//   * every workgroup adds its masses into a fixed-point accumulator with integer atomics (performed at
//     the memory side: exact, order-independent, no L2 coherence involved); the same atomic adds 1 to an
//     arrival count kept in the upper bits of every word,
//   * a workgroup starts step i's draw once the words it polls (returning atomics) show that all workgroups
//     of its layer have added their share of step i-1: the complete total arrives in the same round trip.
// No fence / L2 write-back is needed: the exchanged words are only ever touched by device-scope atomics.
// All workgroups of the launch must be co-resident (96 at the headline shape, one per CU of 256); waits are
// bounded and raise an error flag instead of hanging.
#include "ltm_device.h"

namespace infv {

constexpr int kBNT = 512;
constexpr int kBRows = 8;
constexpr int kBMaxN = 256;
constexpr int kBNIter = kBMaxN / 64;

struct BatchSmem { int cdf, sidx, gsum, misc, tab0, tab, box_val, box_row, w, bin_box, edge_box, edge_dx, Sp0, Sp1, Ssm, Snew, Dsm, Msm, total; };

__host__ __device__ inline BatchSmem batch_smem(int N, int S, int rows, int tabw) {
    BatchSmem m;
    int o = 0;
    auto take = [&](int n) { int r = o; o += (n + 3) & ~3; return r; };
    m.cdf = take(kBins);
    m.sidx = take(S);
    m.gsum = take(2 * kBNT);
    m.misc = take(64);                                 // [0,16) cq | [16,32) asum | [32] wait flag
    m.tab0 = take(N * tabw);                           // static slot ids
    m.tab = take(N * tabw);                            // resolved source boxes of the current step
    m.box_val = take(N);
    m.box_row = take(N);
    m.w = take(N);
    m.bin_box = take(kBins);
    m.edge_box = take(kBins + 4);
    m.edge_dx = take(kBins);
    m.Sp0 = take(kBRows * (N + 4));
    m.Sp1 = take(kBRows * (N + 4));
    m.Ssm = take(kBRows * (N + 2));
    m.Snew = take(kBRows * (rows + 1));
    m.Dsm = take(kBRows * kDPitch);
    m.Msm = take(kBRows * kMPitch);
    m.total = o;
    return m;
}

__device__ inline unsigned long long coherent_read(unsigned long long* p) { return atomicAdd(p, 0ull); }

__global__ __launch_bounds__(kBNT) void chain_batch_kernel(ChainBatchArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = a.N, H = a.H, Q = a.Q, QS = a.QS;
    const int rows = a.op.rows, tabw = a.op.tabw;
    const BatchSmem m = batch_smem(N, a.S, rows, tabw);
    const int b = blockIdx.x;
    const int h = b % H, qs = (b / H) % QS, l = b / (H * QS);
    const int blocks_per_layer = H * QS;
    const int sp = N + 4, sstride = N + 2, sn = rows + 1;
    float* Spc = lds + m.Sp0;                           // current bias-free scores of this tile
    float* Spn = lds + m.Sp1;
    float* Ssm = lds + m.Ssm;
    float* Snew = lds + m.Snew;
    float* cqs = lds + m.misc;
    float* asum = lds + m.misc + 16;
    int32_t* sidx = reinterpret_cast<int32_t*>(lds + m.sidx);
    int32_t* tab0 = reinterpret_cast<int32_t*>(lds + m.tab0);
    int32_t* tab = reinterpret_cast<int32_t*>(lds + m.tab);
    const long tile = (((long)l * H + h) * Q + qs * kBRows);
    const int valid = min(kBRows, Q - qs * kBRows);
    const bool writer = (h == 0 && qs == 0);
    const long tile_snew = (long)rows * a.snew_ld;       // floats of one step's rows

    // ---- one-time set-up: tables, slot table, score tile ----
    {
        const int n4 = N / 4;
        const int sr = tid / n4, sc4 = tid - sr * n4;
        floatx4 v = {0.f, 0.f, 0.f, 0.f};
        if (sr < valid) v = *reinterpret_cast<const floatx4*>(a.Sp_in + (tile + sr) * N + sc4 * 4);
        if (sr < kBRows) *reinterpret_cast<floatx4*>(&Spc[sr * sp + sc4 * 4]) = v;
        if (tid < kBRows) cqs[tid] = (tid < valid) ? a.cq[tile + tid] : 0.f;
        if (tid < N) {
            (lds + m.box_val)[tid] = a.op.box_val[tid];
            reinterpret_cast<int32_t*>(lds + m.box_row)[tid] = a.op.box_row[tid];
            (lds + m.w)[tid] = a.w[tid];
        }
        if (tid < kBins) {
            reinterpret_cast<int32_t*>(lds + m.bin_box)[tid] = a.st.bin_box[tid];
            (lds + m.edge_dx)[tid] = a.st.edge_dx[tid];
        }
        if (tid <= kBins) reinterpret_cast<int32_t*>(lds + m.edge_box)[tid] = a.st.edge_box[tid];
        for (int e = tid; e < N * tabw; e += kBNT) tab0[e] = a.op.slot_tab[e];
        if (a.draw_mode == 2)
            for (int s = tid; s < a.S; s += kBNT) sidx[s] = a.uniform_idx[s];
    }
    // prefetch of step 0: S'new tile elements and this thread's uniform
    float sn_reg[4];
    double u_reg = 2.0;
    int sn_off[4], sn_lds[4];                            // per-thread element offsets, computed once
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int e = tid + k * kBNT;
        const int qr = e & (kBRows - 1), nr = e >> 3;    // kBRows == 8: 8 consecutive columns of S'new per new row
        sn_off[k] = (e < kBRows * rows && qr < valid) ? (int)((long)nr * a.snew_ld + tile + qr) : -1;
        sn_lds[k] = (e < kBRows * rows) ? qr * sn + nr : -1;
    }
    const double* u_base = a.u + (long)l * a.S + tid;
    const long u_step = (long)a.L * a.S;
    const bool has_u = a.draw_mode == 1 && tid < a.S;
    auto prefetch = [&](int i) {
        const float* sb = a.Snew + i * tile_snew;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float v = 0.f;
            if (sn_off[k] >= 0) {
                v = sb[sn_off[k]];
                for (int x = 1; x < a.snew_splitk; ++x) v += sb[sn_off[k] + x * a.snew_split_stride];
            }
            sn_reg[k] = v;
        }
        if (has_u) u_reg = u_base[i * u_step];
    };
    prefetch(0);
    __syncthreads();

#define BSTAMP(k) do { if (a.dbg != nullptr && b == 0 && tid == 0 && i == 5) { a.dbg[k] = wall_clock64(); a.dbg[8 + k] = clock64(); } } while (0)
    // running ring indices (global step g = a.step0 + i): g % 3 and g % ring without per-step 64-bit division
    int g3 = (int)(a.step0 % 3), slot_run = (int)(a.step0 % a.ring);
    const int ring_n = (int)a.ring;
    for (int i = 0; i < a.n_steps; ++i) {
        BSTAMP(0);
        unsigned long long* acc_prev = a.acc[g3 == 0 ? 2 : g3 - 1] + l * kBins;      // (g + 2) % 3
        unsigned long long* acc_cur = a.acc[g3] + l * kBins;
        unsigned long long* acc_clr = a.acc[g3 == 2 ? 0 : g3 + 1] + l * kBins;      // (g + 1) % 3
        const long slot = slot_run;
        if (++g3 == 3) g3 = 0;
        if (++slot_run == ring_n) slot_run = 0;
        // park the prefetched S'new tile
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (sn_lds[k] >= 0) Snew[sn_lds[k]] = sn_reg[k];
        const double my_u = u_reg;
        // ---- totals of the previous step: every word carries the number of workgroups that have added their
        // share; the lanes that need a word poll it until the count is complete (mass and arrival in ONE round trip)
        double mass_prev = 0.0;
        const bool steady = a.draw_mode == 1 && !((i == 0) && (((a.override_mask >> l) & 1u) || a.first_from_parts));
        if (steady && tid < kBins - 1) {
            unsigned long long v = coherent_read(acc_prev + tid);
            if (i > 0) {
                int spins = 0;
                while ((v >> kArriveShift) < (unsigned long long)(blocks_per_layer + a.expect_extra)) {
                    __builtin_amdgcn_s_sleep(1);
                    // give up loudly: the word lives in host-visible memory, every later call on the handle reports it
                    if (++spins > a.spin_limit) { __hip_atomic_store(a.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
                    v = coherent_read(acc_prev + tid);
                }
            }
            mass_prev = mass_of(v);
        }
        BSTAMP(1);
        if (writer && tid < kBins) atomicExch(acc_clr + tid, 0ull);      // slot of the NEXT step: idle until then
        // ---- draw ----
        if (a.draw_mode == 1) {
            const bool ovr = (i == 0) && ((a.override_mask >> l) & 1u);
            DrawRegs<1> dr;
            dr.us[0] = my_u; dr.acc = 0.0; dr.ovr[0] = dr.ovr[1] = 0.f;
            if (ovr) {
                if (tid < 64) {
                    dr.ovr[0] = a.probs_override[l * kBins + tid];
                    if (tid + 64 < kBins - 1) dr.ovr[1] = a.probs_override[l * kBins + tid + 64];
                }
            } else if (i == 0 && a.first_from_parts) {
                const int j = tid & (kBins - 1), grp = tid / kBins;
                if (j < kBins - 1)
                    for (int p = grp; p < a.parts; p += kBNT / kBins) dr.acc += (double)a.part_prev[((long)l * a.parts + p) * kBins + j];
            } else {
                dr.acc = mass_prev;
            }
            const bool last = (i == a.n_steps - 1);
            draw_finish<kBNT, 1>(dr, ovr, reinterpret_cast<const int32_t*>(lds + m.bin_box), a.S, lds + m.cdf, sidx,
                                 reinterpret_cast<double*>(lds + m.gsum),
                                 (writer && last) ? a.probs_out + l * kBins : nullptr,
                                 (writer && last) ? a.bins_out + (long)l * a.S : nullptr,
                                 (writer && last) ? a.idx_out + (long)l * a.S : nullptr, nullptr, nullptr,
                                 (writer && a.probs_tr && i < a.trace_steps) ? a.probs_tr + ((long)i * a.L + l) * kBins : nullptr,
                                 (writer && a.bins_tr && i < a.trace_steps) ? a.bins_tr + ((long)i * a.L + l) * a.S : nullptr);
        } else {
            __syncthreads();
        }
        BSTAMP(2);
        // resolved gather table of this step (published for the UC kernel by the layer's writer)
        for (int e = tid; e < N * tabw; e += kBNT) {
            const int sl = tab0[e];
            tab[e] = (sl >= 0) ? sidx[sl] : -1;
        }
        if (i + 1 < a.n_steps) prefetch(i + 1);
        __syncthreads();
        BSTAMP(3);
        // ---- score recurrence ----
        {
            const float* box_val = lds + m.box_val;
            const int32_t* box_row = reinterpret_cast<const int32_t*>(lds + m.box_row);
            const int row = wave;
            const float cqr = cqs[row];
            const float* Sp = Spc + row * sp;
#pragma unroll
            for (int k = 0; k < kBNIter; ++k) {
                const int n = lane + 64 * k;
                if (n < N) {
                    float acc = 0.f;
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
                    const int r = box_row[n];
                    if (r >= 0) acc += Snew[row * sn + r];
                    Spn[row * sp + n] = acc;
                    Ssm[row * sstride + n] = acc + cqr;
                }
            }
        }
        __syncthreads();
        BSTAMP(4);
        row_phase_wave<false>(Ssm, sstride, N, valid, lds + m.w, a.w_out, reinterpret_cast<const int32_t*>(lds + m.edge_box),
                              lds + m.edge_dx, lds + m.Dsm, lds + m.Msm, asum, nullptr, acc_cur, kBRows,
                              1ull << kArriveShift);
        BSTAMP(5);
        BSTAMP(6);
        // outputs for the UC kernel go out after the arrival, off the other workgroups' critical path:
        // the resolved gather table (layer's writer) and this step's SCORES (alpha_rows_kernel turns them into the
        // softmax weights and row sums later, off the chain)
        if (writer) {
            int32_t* tab_out = a.tab_ring + slot * a.tab_slot + (long)l * N * tabw;
            for (int e = tid; e < N * tabw; e += kBNT) tab_out[e] = tab[e];
        }
        if (wave < valid) {
            float* al = a.alpha_ring + slot * a.alpha_slot + (tile + wave) * N;
#pragma unroll
            for (int k = 0; k < kBNIter; ++k) {
                const int n = lane + 64 * k;
                if (n < N) al[n] = Ssm[wave * sstride + n];
            }
        }
        { float* t = Spc; Spc = Spn; Spn = t; }
        BSTAMP(7);
        // (next iteration's first LDS writes go to Snew, whose last readers sit before the barrier above)
    }
    // ---- hand the score tile to the next launch ----
    __syncthreads();
    {
        const int n4 = N / 4;
        const int sr = tid / n4, sc4 = tid - sr * n4;
        if (sr < valid) *reinterpret_cast<floatx4*>(a.Sp_out + (tile + sr) * N + sc4 * 4) =
                            *reinterpret_cast<const floatx4*>(&Spc[sr * sp + sc4 * 4]);
    }
}

// ------------------------------------------------------------------------------------------------------
// Scores -> softmax weights of `n_steps` ring slots, in place:  alpha[n] = w_n e^{S_n} / (sum_m w_m e^{S_m} + w_out)
// (LTM.py:247-248,269-282 in closed form) and asum = sum_n alpha[n].  One wave per (step, layer, head, query) row;
// same arithmetic as row_phase_wave<true>.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void alpha_rows_kernel(float* __restrict__ alpha_ring, long alpha_slot,
                                                         float* __restrict__ asum_ring, long asum_slot, long slot0, int ring,
                                                         int n_steps, int rows_per_step, int N, const float* __restrict__ w,
                                                         float w_out) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= (long)n_steps * rows_per_step) return;
    const int i = (int)(r / rows_per_step), row = (int)(r - (long)i * rows_per_step);
    const long slot = (slot0 + i) % ring;
    float* a = alpha_ring + slot * alpha_slot + (long)row * N;
    float sv[4];
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int n = lane + 64 * k;
        sv[k] = (n < N) ? a[n] : -INFINITY;
        m = fmaxf(m, sv[k]);
    }
    m = wave_max(m);
    float e[4];
    float esum = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int n = lane + 64 * k;
        e[k] = (n < N) ? w[n] * __expf(sv[k] - m) : 0.f;
        esum += e[k];
    }
    esum = wave_sum(esum);
    const float inv = 1.0f / (esum + w_out * __expf(-m));
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int n = lane + 64 * k;
        if (n < N) a[n] = e[k] * inv;
    }
    if (lane == 0) asum_ring[slot * asum_slot + row] = esum * inv;
}

hipError_t launch_alpha_rows(float* alpha_ring, long alpha_slot, float* asum_ring, long asum_slot, long slot0, int ring,
                             int n_steps, int rows_per_step, int N, const float* w, float w_out, hipStream_t stream) {
    if (n_steps <= 0) return hipSuccess;
    if (N > 256) return hipErrorInvalidValue;
    const long rows = (long)n_steps * rows_per_step;
    hipLaunchKernelGGL(alpha_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, alpha_ring, alpha_slot,
                       asum_ring, asum_slot, slot0, ring, n_steps, rows_per_step, N, w, w_out);
    return hipGetLastError();
}

size_t chain_batch_lds_bytes(int N, int S, int rows, int tabw) { return (size_t)batch_smem(N, S, rows, tabw).total * sizeof(float); }

bool chain_batch_supported(int N, int S, int rows, int tabw, int n_blocks) {
    return N <= kBMaxN && N % 16 == 0 && S <= kBNT && rows <= kBMaxN && tabw <= 16 && (tabw & 3) == 0 &&
           n_blocks <= 384 && chain_batch_lds_bytes(N, S, rows, tabw) <= 100 * 1024;
}

static hipError_t chain_batch_attr() {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(chain_batch_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    return hipSuccess;
}

// The persistent kernel spin-waits on its own workgroups: all of them must fit on the device at once, with this
// kernel's registers and LDS, even if nothing else left room (other kernels may still delay residency; the waits are
// bounded and report through the error word).  One workgroup per CU less than the API's answer: the occupancy
// query reads one high for some SGPR counts (MI355X_MICROARCH.md, residency).
bool chain_batch_resident(int N, int S, int rows, int tabw, int n_blocks) {
    if (chain_batch_attr() != hipSuccess) return false;
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, chain_batch_kernel, kBNT, chain_batch_lds_bytes(N, S, rows, tabw)) != hipSuccess) return false;
    const int safe = per_cu > 1 ? per_cu - 1 : per_cu;
    return (long)safe * cus >= n_blocks;
}

hipError_t launch_chain_batch(const ChainBatchArgs& a, hipStream_t stream) {
    if (hipError_t e = chain_batch_attr()) return e;
    if (a.n_steps <= 0) return hipSuccess;
    const int blocks = a.H * a.QS * a.L;
    if (!chain_batch_supported(a.N, a.S, a.op.rows, a.op.tabw, blocks)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(chain_batch_kernel, dim3(blocks), dim3(kBNT), chain_batch_lds_bytes(a.N, a.S, a.op.rows, a.op.tabw),
                       stream, a);
    return hipGetLastError();
}

}  // namespace infv
