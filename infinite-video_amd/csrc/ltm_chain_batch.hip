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

// Poll of a layer's accumulator words by one wave: lane i reads words (2 i, 2 i + 1) = bins (i, i + 64) (acc_word) with ONE
// 16-byte buffer load that bypasses the vector L1 and is served past the XCD's L2 (sc1).  The words are only ever written
// by device-scope atomics, which execute at the memory side and leave no line behind in any L2, so the load sees every add
// that has completed; an aligned 8-byte half never tears.  (Round 2 polled with returning atomicAdd(p, 0): correct too,
// but every poll then serialises at the memory side with the deposits and the other workgroups' polls of the same word --
// ~12 ns each, 24-48 workgroups, several polls per step.)  A stale or torn read could only delay the arrival count: the
// waits are bounded and report through the error word.
typedef unsigned int uintx4_t __attribute__((ext_vector_type(4)));
__device__ inline void poll_pair(const unsigned long long* layer_words, int lane, unsigned long long& w0, unsigned long long& w1) {
#ifdef INFV_POLL_ATOMIC                                      /* A/B builds only: round 2's returning-atomic poll */
    unsigned long long* p = const_cast<unsigned long long*>(layer_words);
    w0 = atomicAdd(p + 2 * lane, 0ull);
    w1 = atomicAdd(p + 2 * lane + 1, 0ull);
    return;
#endif
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned long long*>(layer_words), 0, kBins * 8, 0x00020000);
    const uintx4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, 0, 16 /* sc1 */);
    w0 = ((unsigned long long)v.y << 32) | v.x;
    w1 = ((unsigned long long)v.w << 32) | v.z;
}

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
            unsigned long long v = coherent_read(acc_prev + acc_word(tid));
            if (i > 0) {
                int spins = 0;
                while ((v >> kArriveShift) < (unsigned long long)(blocks_per_layer + a.expect_extra)) {
                    __builtin_amdgcn_s_sleep(1);
                    // give up loudly: the word lives in host-visible memory, every later call on the handle reports it
                    if (++spins > a.spin_limit) { __hip_atomic_store(a.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
                    v = coherent_read(acc_prev + acc_word(tid));
                }
            }
            mass_prev = mass_of(v);
        }
        BSTAMP(1);
        if (writer && tid < kBins) atomicExch(acc_clr + acc_word(tid), 0ull);      // slot of the NEXT step: idle until then
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

// ======================================================================================================
// chain_batch2_kernel: the same role S with only the truly sequential work on the chain.
//
// A step's draw needs the bin masses of the previous step, which need the scores at the 127 interior edges
// of the sticky histogram, which are the scores of the boxes holding the bins' left edges -- the same boxes the
// resampling reads (LTM.py:207-208: ts = bins[b]).  So the recurrence only ever feeds back through the 128
// "points" j = 0..127 (left edge of bin j, box pb[j] = bin_box[j]); every other box is a pure output, rebuilt
// later, chunk-parallel, by alpha_rows2_kernel from what this kernel publishes per step: the point scores and the
// table of drawn bins.  Eight waves, wave r owns query row r.  Per step:
//   wave 0      poll the previous totals (the poll is issued right behind this workgroup's own deposit, one step
//               earlier), normalise, sequential fp32 cdf (systolic DPP scan)                      -> barrier 1
//   all waves   one lower-bound search per thread (fp32 compares against the round-up of the f64 uniform:
//               equivalent to the f64 compare), result written straight into the gather table;
//               wave 7 also parks the NEXT step's inputs (new-row scores, uniforms), loaded two steps ago -> barrier 2
//   wave = row  recurrence of the row's 128 point scores (2 per lane, state private to the wave), edge densities,
//               trapezoid masses: all in registers, neighbours through DPP shifts                    -> barrier 3
//   tid < 127   8-row sum, fixed-point deposit (+ arrival count); wave 0 re-arms its poll and goes round
//   waves 1-6   publish the step (point scores of all 8 rows; the layer's writer: drawn-bin table, source-box table)
//   wave 7      requests the inputs of step i+3.
// vmcnt counts loads, stores and atomics in issue order, so a wave that waits for a load also waits for the
// acknowledgement of every store it issued before -- hence the split: the poller issues no store between arming and
// reading its poll, the storing waves load nothing, the loading wave stores nothing.  While a streaming kernel shares
// the CU every vector-memory INSTRUCTION also queues ~0.2 us at issue: each role issues a handful per step.
// Requires the plan's edges to be the bins' left edges (StickyView.points_ok), the sticky draw (mode 1), rows <= 128.
// ======================================================================================================
constexpr int kB2Ld = 4;                  // float4 pieces of the new-row score tile the loader holds per lane (2 * rows <= 64 * kB2Ld)
struct Batch2Smem { int cdf, coarse, pos, tabb, box_val, box_row, pb, sc0, sc1, Snew, uf, Msm, total; };
constexpr int kScPitch = kBins + 4;

// rpw: query rows per wave (the workgroup's tile is 8 * rpw rows of one head)
__host__ __device__ inline Batch2Smem batch2_smem(int N, int S, int rows, int tabw, int rpw = 1) {
    Batch2Smem m;
    int o = 0;
    auto take = [&](int n) { int r = o; o += (n + 3) & ~3; return r; };
    m.cdf = take(kBins);
    m.coarse = take(16);
    m.pos = take(S);
    m.tabb = take(N * tabw);
    m.box_val = take(N);
    m.box_row = take(N);
    m.pb = take(kBins);
    m.sc0 = take(rpw * kBRows * kScPitch);
    m.sc1 = take(rpw * kBRows * kScPitch);
    m.Snew = take(2 * rpw * kBRows * (rows + 1));
    m.uf = take(2 * S);
    m.Msm = take(rpw * kBRows * kMPitch);
    m.total = o;
    return m;
}

// lane i <- lane i+1; lane 63 <- fill
__device__ inline float wave_shl1(float v, float fill) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), 0x130, 0xf, 0xf, false));
}

// RPW query rows per wave: wave w owns rows w, w + 8, ... of the workgroup's 8*RPW-row tile.  More rows per wave = fewer
// workgroups on the chip (H * ceil(Q / (8 RPW)) per layer; each owns a whole CU by its registers) and fewer arrivals per
// exchange; the rows of a wave are independent dependency chains that share the table reads and overlap each other's
// LDS / DPP latencies.  Same arithmetic per row for every RPW: the results do not depend on it.
template <int RPW>
__global__ __launch_bounds__(kBNT) void chain_batch2_kernel(ChainBatchArgs a) {
    constexpr int TR = kBRows * RPW;                                   // rows of the tile
    constexpr int PPR = TR / 4;                                        // float4 pieces per new row of the S'new tile
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (!(a.exp_flags & 1)) __builtin_amdgcn_s_setprio(3);
    wg_stamp_begin(a.wg_stamps);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = a.N, H = a.H, Q = a.Q, QS = a.QS;
    const int rows = a.op.rows, tabw = a.op.tabw;
    const Batch2Smem m = batch2_smem(N, a.S, rows, tabw, RPW);
    const int b = blockIdx.x;
    const int h = b % H, qs = (b / H) % QS, l = b / (H * QS);
    const int blocks_per_layer = H * QS;
    const int sn = rows + 1, sn_tile = TR * sn;
    float* cdf = lds + m.cdf;
    float* coarse = lds + m.coarse;
    int32_t* tabb = reinterpret_cast<int32_t*>(lds + m.tabb);          // bin of the k-th resampled slot of box n, -1 = none
    int32_t* pb = reinterpret_cast<int32_t*>(lds + m.pb);
    const float* box_val = lds + m.box_val;
    const int32_t* box_row = reinterpret_cast<const int32_t*>(lds + m.box_row);
    float* Msm = lds + m.Msm;
    const long tile = (((long)l * H + h) * Q + qs * TR);
    const int valid = min(TR, Q - qs * TR);
    const bool writer = (h == 0 && qs == 0);
    const bool loader = wave == kBRows - 1;
    const long tile_snew = (long)rows * a.snew_ld;

    // ---- one-time set-up ----
    {
        int32_t* pos = reinterpret_cast<int32_t*>(lds + m.pos);
        if (tid < N) {
            (lds + m.box_val)[tid] = a.op.box_val[tid];
            reinterpret_cast<int32_t*>(lds + m.box_row)[tid] = a.op.box_row[tid];
        }
        if (tid < kBins) pb[tid] = a.st.bin_box[tid];
        if (tid < a.S) pos[tid] = -1;
        __syncthreads();
        for (int e = tid; e < N * tabw; e += kBNT) {
            const int sl = a.op.slot_tab[e];
            tabb[e] = -1;
            if (sl >= 0) pos[sl] = e;
        }
        __syncthreads();
    }
    const int my_pos = (tid < a.S) ? reinterpret_cast<const int32_t*>(lds + m.pos)[tid] : -1;
    bool row_ok[RPW];
#pragma unroll
    for (int j = 0; j < RPW; ++j) row_ok[j] = wave + kBRows * j < valid;
    float* scc = lds + m.sc0 + wave * kScPitch;                        // this wave's first row of point scores (bias-free), current
    float* scn = lds + m.sc1 + wave * kScPitch;                        // ... and next  (row j of the wave: + j * kBRows * kScPitch)
    // the two points of this lane: boxes, operator entries, edge validity and spacings (static)
    const int n0 = pb[lane], n1 = pb[lane + 64];
    const float val0 = box_val[n0], val1 = box_val[n1];
    const int br0 = box_row[n0], br1 = box_row[n1];
    const bool e0ok = a.st.edge_box[lane] >= 0, e1ok = a.st.edge_box[lane + 64] >= 0;     // edge 0 lies left of every box
    const float dx0 = a.st.edge_dx[lane], dx1 = a.st.edge_dx[lane + 64];
    const float dxa = a.st.edge_dx[lane + 1], dxb = (lane + 65 < kBins) ? a.st.edge_dx[lane + 65] : 0.f;
    float cqr[RPW];
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const int row = wave + kBRows * j;
        cqr[j] = row_ok[j] ? a.cq[tile + row] : 0.f;
        const float i0 = row_ok[j] ? a.Sp_in[(tile + row) * N + n0] : 0.f;
        const float i1 = row_ok[j] ? a.Sp_in[(tile + row) * N + n1] : 0.f;
        scc[j * kBRows * kScPitch + lane] = i0;
        scc[j * kBRows * kScPitch + lane + 64] = i1;
        if (a.publish_init && row_ok[j]) {
            // first launch of a call: the state before its first step, for the rows alpha_rows2_kernel rebuilds of that step
            const long prev = (a.step0 % a.ring == 0) ? a.ring - 1 : a.step0 % a.ring - 1;
            float* cr = a.crit_ring + prev * a.crit_slot + (tile + row) * kBins;
            cr[lane] = i0;
            cr[lane + 64] = i1;
        }
    }

    // ---- loader (wave 7): the S'new tile and the uniforms of a step in registers, two sets (steps in flight: i+1, i+2).
    // Wide loads only.   S'new tile: new row nr holds this tile's TR scores contiguously -> float4 e4 = lane + 64 k:
    // row e4 / PPR, piece e4 % PPR;   uniforms: S float64 -> double2 per lane
    struct LdSet { floatx4 sn[kB2Ld]; double2 u[4]; };
    LdSet ldA, ldB;
    auto ld_request = [&](int i, LdSet& r) {
        const float* sb = a.Snew + (long)i * tile_snew + tile;
        const int splitk = a.snew_splitk;
        const long split_stride = a.snew_split_stride;
#pragma unroll
        for (int k = 0; k < kB2Ld; ++k) {
            const int e4 = lane + 64 * k;
            const int nr = e4 / PPR, hf = e4 % PPR;
            floatx4 v = {0.f, 0.f, 0.f, 0.f};
            if (nr < rows && 4 * hf < valid) {
                const float* src = sb + (long)nr * a.snew_ld + 4 * hf;
                v = *reinterpret_cast<const floatx4*>(src);
                for (int x = 1; x < splitk; ++x) v += *reinterpret_cast<const floatx4*>(src + x * split_stride);
            }
            r.sn[k] = v;
        }
        const double2* ub = reinterpret_cast<const double2*>(a.u + ((long)i * a.L + l) * a.S);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int s2 = lane + 64 * k;
            r.u[k] = (2 * s2 < a.S) ? ub[s2] : make_double2(2.0, 2.0);
        }
    };
    auto round_up = [](double u) {
        // (double)c < u  <=>  c < uf for every float c, with uf = the smallest float >= u
        float f = (float)u;
        if ((double)f < u) f = __int_as_float(__float_as_int(f) + 1);
        return f;
    };
    auto ld_park = [&](int i, const LdSet& r) {                         // into the tiles of parity i & 1
        float* st = lds + m.Snew + (i & 1) * sn_tile;
        float* uf = lds + m.uf + (i & 1) * a.S;
#pragma unroll
        for (int k = 0; k < kB2Ld; ++k) {
            const int e4 = lane + 64 * k;
            const int nr = e4 / PPR, hf = e4 % PPR;
            if (nr < rows) {
                st[(4 * hf + 0) * sn + nr] = r.sn[k].x; st[(4 * hf + 1) * sn + nr] = r.sn[k].y;
                st[(4 * hf + 2) * sn + nr] = r.sn[k].z; st[(4 * hf + 3) * sn + nr] = r.sn[k].w;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int s2 = lane + 64 * k;
            if (2 * s2 < a.S) { uf[2 * s2] = round_up(r.u[k].x); uf[2 * s2 + 1] = round_up(r.u[k].y); }
        }
    };
    if (loader) {
        ld_request(0, ldA);
        ld_park(0, ldA);
        if (a.n_steps > 1) ld_request(1, ldB);                         // odd steps live in set B, even steps in set A
        if (a.n_steps > 2) ld_request(2, ldA);
    }

    int g3 = (int)(a.step0 % 3), slot_run = (int)(a.step0 % a.ring);
    const int ring_n = (int)a.ring;
    const unsigned long long need_full = (unsigned long long)(blocks_per_layer + a.expect_extra);
    // wave 0: totals of the previous step, polled one step ahead (lane j: bins j and j+64)
    unsigned long long pv0 = 0ull, pv1 = 0ull;
    const bool ovr0 = ((a.override_mask >> l) & 1u) != 0;
    const bool special0 = ovr0 || a.first_from_parts;                 // step 0 takes its probabilities from elsewhere
    if (wave == 0 && !special0) {
        poll_pair(a.acc[g3 == 0 ? 2 : g3 - 1] + l * kBins, lane, pv0, pv1);
    }
    __syncthreads();

#define B2STAMP(k) do { if (a.dbg != nullptr && b == 0 && tid == 0 && i == 5) { a.dbg[k] = wall_clock64(); a.dbg[8 + k] = clock64(); } } while (0)
    for (int i = 0; i < a.n_steps; ++i) {
        B2STAMP(0);
        unsigned long long* acc_prev = a.acc[g3 == 0 ? 2 : g3 - 1] + l * kBins;
        unsigned long long* acc_cur = a.acc[g3] + l * kBins;
        unsigned long long* acc_clr = a.acc[g3 == 2 ? 0 : g3 + 1] + l * kBins;
        const long slot = slot_run;
        if (++g3 == 3) g3 = 0;
        if (++slot_run == ring_n) slot_run = 0;
        const bool last = (i == a.n_steps - 1);
        const float* Snew = lds + m.Snew + (i & 1) * sn_tile;
        // ---- wave 0: probabilities -> cdf ----
        if (wave == 0) {
            constexpr int nb = kBins - 1;
            const int j1 = lane + 64;
            float p0, p1;
            if (i == 0 && ovr0) {
                p0 = a.probs_override[l * kBins + lane];
                p1 = (j1 < nb) ? a.probs_override[l * kBins + j1] : 0.f;
            } else {
                double a0 = 0.0, a1 = 0.0;
                if (i == 0 && a.first_from_parts) {
                    for (int p = 0; p < a.parts; ++p) {
                        a0 += (double)a.part_prev[((long)l * a.parts + p) * kBins + lane];
                        if (j1 < nb) a1 += (double)a.part_prev[((long)l * a.parts + p) * kBins + j1];
                    }
                } else {
                    // the poll was issued one step ago; step 0 reads totals completed by an earlier launch (no count)
                    const unsigned long long need = (i > 0 && !(a.exp_flags & 16)) ? need_full : 0ull;
                    int spins = 0;
                    while ((pv0 >> kArriveShift) < need || (j1 < nb && (pv1 >> kArriveShift) < need)) {
                        if (a.exp_flags & 2) __builtin_amdgcn_s_sleep(16); else __builtin_amdgcn_s_sleep(1);
                        if (++spins > a.spin_limit) { __hip_atomic_store(a.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
                        poll_pair(acc_prev, lane, pv0, pv1);
                    }
                    a0 = mass_of(pv0);
                    a1 = (j1 < nb) ? mass_of(pv1) : 0.0;
                }
                const float raw0 = (float)a0;
                const float raw1 = (j1 < nb) ? (float)a1 : 0.f;
                const float tot1 = (float)wave_sum_f64((double)raw0 + (double)raw1);
                const float q0 = raw0 / tot1, q1 = raw1 / tot1;                       // LTM.py:203
                const float tot2 = (float)wave_sum_f64((double)q0 + (double)q1);
                p0 = q0 / tot2; p1 = q1 / tot2;                                       // Categorical's own normalisation
            }
            if (j1 >= nb) p1 = 0.f;
            B2STAMP(1);
            if (writer) {
                if (last) { a.probs_out[l * kBins + lane] = p0; if (j1 < nb) a.probs_out[l * kBins + j1] = p1; }
                if (a.probs_tr != nullptr && i < a.trace_steps) {
                    float* pt = a.probs_tr + ((long)i * a.L + l) * kBins;
                    pt[lane] = p0; if (j1 < nb) pt[j1] = p1;
                }
            }
            // sequential fp32 running sum in bin order (torch.multinomial, CPU) as a systolic scan over the lanes
            float c0 = p0;
#pragma unroll
            for (int t = 0; t < 63; ++t) c0 = dpp_f32<0x138>(c0) + p0;          // bins 0..63
            const float carry = readlane_f32(c0, 63);
            const float q1s = (lane == 0) ? carry + p1 : p1;                    // bin 64 continues the chain
            float c1 = q1s;
#pragma unroll
            for (int t = 0; t < 62; ++t) c1 = dpp_f32<0x138>(c1) + q1s;         // bins 64..126 (lane 63: padding)
            const float run = readlane_f32(c1, 62);
            const float f0 = c0 / run;
            const float f1 = (j1 < nb) ? ((j1 == nb - 1) ? 1.f : c1 / run) : 2.f;   // last bucket forced to 1; pad never below a uniform
            cdf[lane] = f0;
            cdf[j1] = f1;
            if ((lane & 7) == 7) { coarse[lane >> 3] = f0; coarse[8 + (lane >> 3)] = f1; }
        }
        __syncthreads();                                                         // barrier 1
        B2STAMP(2);
        // Clear the accumulator slot of step i+1 (it held the totals of step i-2).  Wave 0 has just seen every arrival of
        // step i-1, and a workgroup arrives only after its own poll of step i-2's totals, so nobody reads the slot any
        // more; nobody adds to it before having seen all arrivals of step i, this workgroup's included -- and that arrival
        // (behind barrier 3) is held back until the clear has been acknowledged: the clearing waves drain vmcnt before
        // they join barrier 3.  (Round 2 issued the clear behind barrier 3, beside the arrival, with nothing ordering the two.)
        if (writer && tid >= 128 && tid < 128 + kBins) atomicExch(acc_clr + acc_word(tid - 128), 0ull);
        if (tid < a.S) {
            // ---- lower bound of this thread's uniform in the cdf == number of entries below it ----
            const float my_uf = (lds + m.uf + (i & 1) * a.S)[tid];
            const floatx4* c4 = reinterpret_cast<const floatx4*>(coarse);
            int grp = 0;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const floatx4 f = c4[v];
                grp += (f.x < my_uf) + (f.y < my_uf) + (f.z < my_uf) + (f.w < my_uf);
            }
            int lo = 8 * grp;
            if (grp < 16) {
                const floatx4 f0 = *reinterpret_cast<const floatx4*>(&cdf[8 * grp]);
                const floatx4 f1 = *reinterpret_cast<const floatx4*>(&cdf[8 * grp + 4]);
                lo += (f0.x < my_uf) + (f0.y < my_uf) + (f0.z < my_uf) + (f0.w < my_uf) +
                      (f1.x < my_uf) + (f1.y < my_uf) + (f1.z < my_uf);
            }
            lo = min(lo, kBins - 2);
            if (my_pos >= 0) tabb[my_pos] = lo;
            if (writer && !loader) {
                if (last) { a.bins_out[(long)l * a.S + tid] = lo; a.idx_out[(long)l * a.S + tid] = pb[lo]; }
                if (a.bins_tr != nullptr && i < a.trace_steps) a.bins_tr[((long)i * a.L + l) * a.S + tid] = lo;
            }
            if (writer && loader) (lds + m.pos)[tid] = __int_as_float(lo);       // wave 7 stores nothing: wave 6 writes its 64 bins out
        }
        // the next step's inputs were requested two steps ago: park them (tiles of the other parity: their last readers,
        // recurrence and search of step i-1, are behind barrier 3 of that step)
        if (loader && !last) { if ((i + 1) & 1) ld_park(i + 1, ldB); else ld_park(i + 1, ldA); }
        __syncthreads();                                                         // barrier 2
        B2STAMP(3);
        // ---- wave = rows: recurrence of the 128 point scores of each of its rows, edge densities, bin masses (registers + DPP) ----
        float acc0[RPW], acc1[RPW];
        {
#pragma unroll
            for (int j = 0; j < RPW; ++j) acc0[j] = acc1[j] = 0.f;
            for (int k0 = 0; k0 < tabw; k0 += 4) {
                const int4 s0 = *reinterpret_cast<const int4*>(&tabb[n0 * tabw + k0]);
                const int4 s1 = *reinterpret_cast<const int4*>(&tabb[n1 * tabw + k0]);
#pragma unroll
                for (int j = 0; j < RPW; ++j) {
                    const float* sc = scc + j * kBRows * kScPitch;
                    const float v0 = sc[max(s0.x, 0)], v1 = sc[max(s0.y, 0)], v2 = sc[max(s0.z, 0)], v3 = sc[max(s0.w, 0)];
                    const float w0 = sc[max(s1.x, 0)], w1 = sc[max(s1.y, 0)], w2 = sc[max(s1.z, 0)], w3 = sc[max(s1.w, 0)];
                    if (s0.x >= 0) acc0[j] = fmaf(val0, v0, acc0[j]);
                    if (s0.y >= 0) acc0[j] = fmaf(val0, v1, acc0[j]);
                    if (s0.z >= 0) acc0[j] = fmaf(val0, v2, acc0[j]);
                    if (s0.w >= 0) acc0[j] = fmaf(val0, v3, acc0[j]);
                    if (s1.x >= 0) acc1[j] = fmaf(val1, w0, acc1[j]);
                    if (s1.y >= 0) acc1[j] = fmaf(val1, w1, acc1[j]);
                    if (s1.z >= 0) acc1[j] = fmaf(val1, w2, acc1[j]);
                    if (s1.w >= 0) acc1[j] = fmaf(val1, w3, acc1[j]);
                }
            }
            // the rows of a wave are independent dependency chains: written phase by phase so that their LDS reads, DPP
            // reductions and transcendentals interleave (same arithmetic per row as one row at a time)
            float es0[RPW], es1[RPW], mx[RPW];
#pragma unroll
            for (int j = 0; j < RPW; ++j) {
                const int row = wave + kBRows * j;
                if (br0 >= 0) acc0[j] += Snew[row * sn + br0];
                if (br1 >= 0) acc1[j] += Snew[row * sn + br1];
                scn[j * kBRows * kScPitch + lane] = acc0[j];
                scn[j * kBRows * kScPitch + lane + 64] = acc1[j];
                // densities at the 129 edges: edge j (1..127) sits in the box of point j, edges 0 and 128 in none (score 0)
                es0[j] = e0ok ? acc0[j] + cqr[j] : 0.f;
                es1[j] = e1ok ? acc1[j] + cqr[j] : 0.f;
                mx[j] = fmaxf(e0ok ? es0[j] : -INFINITY, e1ok ? es1[j] : -INFINITY);
            }
#pragma unroll
            for (int j = 0; j < RPW; ++j) mx[j] = fmaxf(wave_max(mx[j]), 0.f);
            float d0[RPW], d1[RPW], d128[RPW];
#pragma unroll
            for (int j = 0; j < RPW; ++j) { d0[j] = expf(es0[j] - mx[j]); d1[j] = expf(es1[j] - mx[j]); d128[j] = expf(0.f - mx[j]); }
            float d0n[RPW], d1n[RPW], d0nn[RPW], d1nn[RPW], zz[RPW];
#pragma unroll
            for (int j = 0; j < RPW; ++j) {
                d0n[j] = wave_shl1(d0[j], readlane_f32(d1[j], 0));                  // D[lane + 1]
                d1n[j] = wave_shl1(d1[j], d128[j]);                                 // D[lane + 65]
                d0nn[j] = wave_shl1(d0n[j], readlane_f32(d1[j], 1));                // D[lane + 2]
                d1nn[j] = wave_shl1(d1n[j], 0.f);                                   // D[lane + 66]  (lane 63: unused)
                zz[j] = (d0[j] + d0n[j]) * dx0 + (d1[j] + d1n[j]) * dx1;
            }
#pragma unroll
            for (int j = 0; j < RPW; ++j) zz[j] = wave_sum(zz[j]) * 0.5f;
#pragma unroll
            for (int j = 0; j < RPW; ++j) {
                const int row = wave + kBRows * j;
                const float inv_z = 1.0f / zz[j];
                // mass of interval j+1 -> bin j (cum[j+1]-cum[j], LTM.py:201-202): lanes take j = lane and lane+64 (< 127)
                Msm[row * kMPitch + lane] = row_ok[j] ? ((d0n[j] * inv_z + d0nn[j] * inv_z) * dxa) * 0.5f : 0.f;
                if (lane + 64 < kBins - 1) Msm[row * kMPitch + lane + 64] = row_ok[j] ? ((d1n[j] * inv_z + d1nn[j] * inv_z) * dxb) * 0.5f : 0.f;
            }
        }
        if (writer && (wave == 2 || wave == 3)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clear above is complete
        __syncthreads();                                                         // barrier 3
        B2STAMP(4);
        if (tid < kBins - 1) {
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < TR; ++r) t += Msm[r * kMPitch + tid];
            if (!(a.exp_flags & 16)) atomicAdd(&acc_cur[acc_word(tid)], (unsigned long long)((double)t * kMassScale + 0.5) + (1ull << kArriveShift));
        }
        if (loader) {
            // request the inputs of step i+3 (set of its parity: parked during this step) in the shadow of wave 0's poll
            if (i + 3 < a.n_steps && !(a.exp_flags & 8)) { if ((i + 3) & 1) ld_request(i + 3, ldB); else ld_request(i + 3, ldA); }
        } else if (wave == 0) {
            // re-arm the poll right behind the deposit and go round: this wave issues nothing else until it has read it
            if (!last && !(a.exp_flags & 16)) {
                poll_pair(acc_cur, lane, pv0, pv1);
            }
            B2STAMP(5);
        } else {
            // ---- waves 1-6, stores only: publish the step for alpha_rows2_kernel / the UC kernel ----
            if (writer) {
                int32_t* tab_out = a.tab_ring + slot * a.tab_slot + (long)l * N * tabw;          // source BOX of every slot (UC kernel)
                int32_t* tabb_out = a.tabb_ring + slot * a.tab_slot + (long)l * N * tabw;        // drawn BIN of every slot
                for (int e = tid - 64; e < N * tabw; e += kBNT - 128) {
                    const int bb = tabb[e];
                    tabb_out[e] = bb;
                    tab_out[e] = (bb >= 0) ? pb[bb] : -1;
                }
                if (wave == 6 && a.S > 448) {                                                    // wave 7's share of the draw diagnostics
                    const int s7 = 448 + lane;
                    const int lo = __float_as_int((lds + m.pos)[s7]);
                    if (s7 < a.S) {
                        if (last) { a.bins_out[(long)l * a.S + s7] = lo; a.idx_out[(long)l * a.S + s7] = pb[lo]; }
                        if (a.bins_tr != nullptr && i < a.trace_steps) a.bins_tr[((long)i * a.L + l) * a.S + s7] = lo;
                    }
                }
            }
            // point scores after this step: own rows from registers; wave 1 also wave 0's rows, wave 6 also wave 7's (from LDS)
            float* cr = a.crit_ring + slot * a.crit_slot + tile * kBins;
#pragma unroll
            for (int j = 0; j < RPW; ++j) {
                const int row = wave + kBRows * j;
                if (row_ok[j] && !(a.exp_flags & 4)) { cr[row * kBins + lane] = acc0[j]; cr[row * kBins + lane + 64] = acc1[j]; }
                const int extra = (wave == 1) ? kBRows * j : ((wave == 6) ? kBRows * j + kBRows - 1 : -1);
                if (extra >= 0 && extra < valid) {
                    const float* sx = lds + ((i & 1) ? m.sc0 : m.sc1) + extra * kScPitch;        // that row's NEXT buffer
                    cr[extra * kBins + lane] = sx[lane];
                    cr[extra * kBins + lane + 64] = sx[lane + 64];
                }
            }
        }
        { float* t = scc; scc = scn; scn = t; }
        B2STAMP(6);
        B2STAMP(7);
        // LDS reuse: cdf / coarse are rewritten by wave 0 after barrier 3, i.e. after every search of this step; the
        // S'new / uniform tiles of parity i are rewritten by wave 7 behind barrier 1 of step i+1; tabb and the bins parked in
        // `pos` behind barrier 1 as well (their readers, the publishing waves, reach that barrier after their stores were
        // issued); Msm behind barrier 2; the rows read by waves 1 and 6 are rewritten behind barrier 2 of step i+1.
    }
    // ---- hand the point scores to the next launch (its set-up reads them back through pb) ----
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const int row = wave + kBRows * j;
        if (row_ok[j]) {
            a.Sp_out[(tile + row) * N + n0] = scc[j * kBRows * kScPitch + lane];
            a.Sp_out[(tile + row) * N + n1] = scc[j * kBRows * kScPitch + lane + 64];
        }
    }
    wg_stamp_end(a.wg_stamps);
}

// ------------------------------------------------------------------------------------------------------
// alpha_rows2_kernel: the chunk-parallel other half of chain_batch2_kernel.  For every step of a sub-batch and every
// (layer, head, query) row it rebuilds the full score row from the point scores of the PREVIOUS step and the step's
// drawn-bin table -- S'_c[n] = val_n * sum_k S'_{c-1}[point tabb_c[n][k]] + S'new_c[row(n)], the same fma chain in
// the same order as the chain kernel's, so the two agree bit for bit on the boxes both compute -- then the
// count-weighted softmax alpha[n] = w_n e^{S_n} / (sum_m w_m e^{S_m} + w_out)  (LTM.py:247-248,269-282 in closed
// form) and asum = sum_n alpha[n].  One workgroup per (step, layer, head), a wave per query row in turn.
// ------------------------------------------------------------------------------------------------------
constexpr int kA2NT = 256;
constexpr int kA2Q = 32;                  // query rows staged per pass

__global__ __launch_bounds__(kA2NT) void alpha_rows2_kernel(AlphaRows2Args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = a.N, Q = a.Q, H = a.H, rows = a.rows, tabw = a.tabw;
    const int i = blockIdx.x, lh = blockIdx.y, l = lh / H;
    const int snp = kA2Q + 1;
    wg_stamp_begin(a.wg_stamps);
#ifdef INFV_EXPERIMENTS
    if (a.prio == 3) __builtin_amdgcn_s_setprio(3);          // (experiment INFV_ALPHA_PRIO)
    else if (a.prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (a.prio == 1) __builtin_amdgcn_s_setprio(1);
#endif
    int32_t* tabb = reinterpret_cast<int32_t*>(lds);                       // [N * tabw]
    float* prev = lds + ((N * tabw + 3) & ~3);                             // [kA2Q][kScPitch]
    float* snew = prev + kA2Q * kScPitch;                                  // [rows][kA2Q + 1]
    const long slot = (a.slot0 + i) % a.ring;
    const long pslot = (a.slot0 + i + a.ring - 1) % a.ring;
    const int32_t* tb = a.tabb_ring + slot * a.tab_slot + (long)l * N * tabw;
    for (int e = tid; e < N * tabw / 4; e += kA2NT) reinterpret_cast<int4*>(tabb)[e] = reinterpret_cast<const int4*>(tb)[e];
    // static operator entries of this lane's boxes
    float val[4]; int brow[4]; float wn[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int n = lane + 64 * k;
        val[k] = (n < N) ? a.box_val[n] : 0.f;
        brow[k] = (n < N) ? a.box_row[n] : -1;
        wn[k] = (n < N) ? a.w[n] : 0.f;
    }
    const bool write_sp = a.Sp_out != nullptr && i == a.n_steps - 1;
    for (int q0 = 0; q0 < Q; q0 += kA2Q) {
        const int qn = min(kA2Q, Q - q0);
        const long row0 = (long)lh * Q + q0;                               // first row of the pass in [L][H][Q] order
        __syncthreads();
        // previous point scores of the pass's rows (contiguous in the ring) and the S'new tile [rows][qn]
        const float* cp = a.crit_ring + pslot * a.crit_slot + row0 * kBins;
        for (int e = tid; e < qn * (kBins / 4); e += kA2NT) {
            const int r = e / (kBins / 4), c4 = e - r * (kBins / 4);
            *reinterpret_cast<floatx4*>(&prev[r * kScPitch + 4 * c4]) = reinterpret_cast<const floatx4*>(cp)[e];
        }
        const float* sb = a.Snew + (long)i * rows * a.snew_ld + row0;
        for (int e = tid; e < rows * kA2Q; e += kA2NT) {
            const int r = e / kA2Q, qq = e - r * kA2Q;
            float v = 0.f;
            if (qq < qn) {
                v = sb[(long)r * a.snew_ld + qq];
                for (int x = 1; x < a.snew_splitk; ++x) v += sb[(long)r * a.snew_ld + qq + x * a.snew_split_stride];
            }
            snew[r * snp + qq] = v;
        }
        __syncthreads();
        for (int qq = wave; qq < qn; qq += kA2NT / 64) {
            const float* pr = prev + qq * kScPitch;
            const float cqv = a.cq[row0 + qq];
            float sv[4];
            float mx = -INFINITY;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int n = lane + 64 * k;
                sv[k] = -INFINITY;
                if (n < N) {
                    float acc = 0.f;
                    for (int k0 = 0; k0 < tabw; k0 += 4) {
                        const int4 src = *reinterpret_cast<const int4*>(&tabb[n * tabw + k0]);
                        const float v0 = pr[max(src.x, 0)], v1 = pr[max(src.y, 0)];
                        const float v2 = pr[max(src.z, 0)], v3 = pr[max(src.w, 0)];
                        if (src.x >= 0) acc = fmaf(val[k], v0, acc);
                        if (src.y >= 0) acc = fmaf(val[k], v1, acc);
                        if (src.z >= 0) acc = fmaf(val[k], v2, acc);
                        if (src.w >= 0) acc = fmaf(val[k], v3, acc);
                    }
                    if (brow[k] >= 0) acc += snew[brow[k] * snp + qq];
                    if (write_sp) a.Sp_out[(row0 + qq) * N + n] = acc;     // bias-free scores of the call's last step (diagnostics)
                    sv[k] = acc + cqv;
                }
                mx = fmaxf(mx, sv[k]);
            }
            mx = wave_max(mx);
            float e[4];
            float esum = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int n = lane + 64 * k;
                e[k] = (n < N) ? wn[k] * __expf(sv[k] - mx) : 0.f;
                esum += e[k];
            }
            esum = wave_sum(esum);
            const float inv = 1.0f / (esum + a.w_out * __expf(-mx));
            float* al = a.alpha_ring + slot * a.alpha_slot + (row0 + qq) * N;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int n = lane + 64 * k;
                if (n < N) al[n] = e[k] * inv;
            }
            if (lane == 0) a.asum_ring[slot * a.asum_slot + row0 + qq] = esum * inv;
        }
    }
    wg_stamp_end(a.wg_stamps);
}

hipError_t launch_alpha_rows2(const AlphaRows2Args& a_, hipStream_t stream) {
    if (a_.n_steps <= 0) return hipSuccess;
    if (a_.N > 256 || (a_.N * a_.tabw) % 4) return hipErrorInvalidValue;
    AlphaRows2Args a = a_;
    a.wg_stamps = exp_stamps_reserve(WG_ALPHA, (long)a.n_steps * a.L * a.H);
    static const int prio = [] { const char* e = exp_env("INFV_ALPHA_PRIO"); return e ? atoi(e) : 0; }();
    a.prio = prio;
    const size_t lds = (size_t)(((a.N * a.tabw + 3) & ~3) + kA2Q * kScPitch + a.rows * (kA2Q + 1)) * sizeof(float);
    hipLaunchKernelGGL(alpha_rows2_kernel, dim3(a.n_steps, a.L * a.H), dim3(kA2NT), lds, stream, a);
    return hipGetLastError();
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
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(chain_batch2_kernel<1>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(chain_batch2_kernel<2>),
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
bool chain_batch2_shape_ok(int draw_mode, int points_ok, int rows, int S, int Q) {
    static const bool want_v2 = [] { const char* e = exp_env("INFV_CHAIN_V1"); return !e || atoi(e) == 0; }();
    return want_v2 && draw_mode == 1 && points_ok && 2 * rows <= 64 * kB2Ld && S <= 512 && S % 2 == 0 && Q % 4 == 0;
}

// Query rows per wave of chain_batch2_kernel for this shape: 2 (16-row tiles, half the workgroups -- each of which owns a
// whole CU by its registers -- and half the arrivals per exchange) when the loader's register tile holds 16 scores per new
// row; INFV_CHAIN_RPW=1 restores 8-row tiles.
int chain_batch2_rpw(int rows, int Q) {
    static const int want = [] { const char* e = exp_env("INFV_CHAIN_RPW"); return e ? atoi(e) : 2; }();
    if (want >= 2 && 4 * rows <= 64 * kB2Ld && Q > kBRows) return 2;
    return 1;
}

bool chain_batch2_applies(const ChainBatchArgs& a) {
    return chain_batch2_shape_ok(a.draw_mode, a.st.points_ok, a.op.rows, a.S, a.Q) && a.crit_ring != nullptr;
}

// Workgroups of the persistent role-S launch for this shape (the kernel launch_chain_batch will choose).
int chain_batch_blocks(int H, int Q, int L, int draw_mode, int points_ok, int rows, int S) {
    const int rpw = chain_batch2_shape_ok(draw_mode, points_ok, rows, S, Q) ? chain_batch2_rpw(rows, Q) : 1;
    return H * ((Q + kBRows * rpw - 1) / (kBRows * rpw)) * L;
}

static size_t chain_batch2_launch_lds(int N, int S, int rows, int tabw, int rpw) {
    // padding LDS keeps the workgroup's CU footprint what the stream layout of consolidate() was tuned for
    static const int pad = [] { const char* e = exp_env("INFV_S_LDS"); return e ? atoi(e) : 0; }();
    size_t lds = (size_t)batch2_smem(N, S, rows, tabw, rpw).total * sizeof(float);
    if ((size_t)pad > lds) lds = pad;
    return lds;
}

// Asks about the kernel and the dynamic LDS size that launch_chain_batch will really use for this shape.
bool chain_batch_resident(int N, int S, int rows, int tabw, int n_blocks, int draw_mode, int points_ok, int Q) {
    if (chain_batch_attr() != hipSuccess) return false;
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
    hipError_t e;
    if (chain_batch2_shape_ok(draw_mode, points_ok, rows, S, Q)) {
        const int rpw = chain_batch2_rpw(rows, Q);
        const size_t lds = chain_batch2_launch_lds(N, S, rows, tabw, rpw);
        e = rpw == 2 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, chain_batch2_kernel<2>, kBNT, lds)
                     : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, chain_batch2_kernel<1>, kBNT, lds);
    } else {
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, chain_batch_kernel, kBNT, chain_batch_lds_bytes(N, S, rows, tabw));
    }
    if (e != hipSuccess) return false;
    const int safe = per_cu > 1 ? per_cu - 1 : per_cu;
    return (long)safe * cus >= n_blocks;
}

hipError_t launch_chain_batch(const ChainBatchArgs& a_in, hipStream_t stream) {
    if (hipError_t e = chain_batch_attr()) return e;
    if (a_in.n_steps <= 0) return hipSuccess;
    ChainBatchArgs a = a_in;
    if (!chain_batch_supported(a.N, a.S, a.op.rows, a.op.tabw, a.H * a.QS * a.L)) return hipErrorInvalidValue;
    if (chain_batch2_applies(a)) {
        const int rpw = chain_batch2_rpw(a.op.rows, a.Q);
        a.QS = (a.Q + kBRows * rpw - 1) / (kBRows * rpw);             // tiles of 8 * rpw query rows
        const int blocks = a.H * a.QS * a.L;
        const size_t lds = chain_batch2_launch_lds(a.N, a.S, a.op.rows, a.op.tabw, rpw);
        a.wg_stamps = exp_stamps_reserve(WG_CHAIN, blocks);
        if (rpw == 2) hipLaunchKernelGGL(chain_batch2_kernel<2>, dim3(blocks), dim3(kBNT), lds, stream, a);
        else hipLaunchKernelGGL(chain_batch2_kernel<1>, dim3(blocks), dim3(kBNT), lds, stream, a);
        return hipGetLastError();
    }
    a.wg_stamps = nullptr;
    hipLaunchKernelGGL(chain_batch_kernel, dim3(a.H * a.QS * a.L), dim3(kBNT), chain_batch_lds_bytes(a.N, a.S, a.op.rows, a.op.tabw),
                       stream, a);
    return hipGetLastError();
}

}  // namespace infv
