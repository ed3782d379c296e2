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
        // this kernel adds into replica 0 of a layer's accumulators (ltm_internal.h: kAccShards)
        unsigned long long* acc_prev = a.acc[g3 == 0 ? 2 : g3 - 1] + (long)l * kAccShards * kBins;      // (g + 2) % 3
        unsigned long long* acc_cur = a.acc[g3] + (long)l * kAccShards * kBins;
        unsigned long long* acc_clr = a.acc[g3 == 2 ? 0 : g3 + 1] + (long)l * kAccShards * kBins;      // (g + 1) % 3
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
            // totals completed by an earlier launch may sit in the other replicas (chain_batch3_kernel)
            if (i == 0)
                for (int r = 1; r < kAccShards; ++r) mass_prev += mass_of(coherent_read(acc_prev + r * kBins + acc_word(tid)));
        }
        BSTAMP(1);
        if (writer)                                                                  // slot of the NEXT step: idle until then
            for (int e = tid; e < kAccShards * kBins; e += kBNT) atomicExch(acc_clr + e, 0ull);
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
// chain_batch3_kernel: the same role S with only the truly sequential work on the chain.
//
// A step's draw needs the bin masses of the previous step, which need the scores at the 127 interior edges
// of the sticky histogram, which are the scores of the boxes holding the bins' left edges -- the same boxes the
// resampling reads (LTM.py:207-208: ts = bins[b]).  So the recurrence only ever feeds back through the 128
// "points" j = 0..127 (left edge of bin j, box pb[j] = bin_box[j]); every other box is a pure output, rebuilt
// later, chunk-parallel, by alpha_rows2_kernel from what this kernel publishes per step: the point scores and the
// table of drawn bins.  Eight waves, wave r owns query rows r, r + 8, ...  Per step:
//   wave 0      poll the previous totals (the poll is issued right behind this workgroup's own deposit, one step
//               earlier), normalise, sequential fp32 cdf                                          -> barrier 1
//   all waves   one lower-bound search per thread (fp32 compares against the round-up of the f64 uniform:
//               equivalent to the f64 compare), result written straight into the gather table    -> barrier 2
//   wave = row  recurrence of the row's 128 point scores (2 per lane, state private to the wave), edge densities,
//               trapezoid masses: all in registers, neighbours through DPP shifts                    -> barrier 3
//   tid < 127   row sum, fixed-point deposit (+ arrival count); wave 0 re-arms its poll and goes round
//   waves 1-6   publish the step (point scores of all rows; the layer's writer: drawn-bin table, source-box table)
//   wave 7      parks the inputs of step i+1 (new-row scores, uniforms), requests those of step i+3.
// vmcnt counts loads, stores and atomics in issue order, so a wave that waits for a load also waits for the
// acknowledgement of every store it issued before -- hence the split: the poller issues no store between arming and
// reading its poll, the storing waves load nothing, the loading wave stores nothing.  While a streaming kernel shares
// the CU every vector-memory INSTRUCTION also queues ~0.2 us at issue: each role issues a handful per step.
// Requires the plan's edges to be the bins' left edges (StickyView.points_ok), the sticky draw (mode 1), rows <= 128.
// (Round 2-3: chain_batch2_kernel -- unsharded exchange, systolic DPP scan, f64 uniforms converted by the loader,
// ping-pong score rows: 5.4-5.7 us per step.)
// ======================================================================================================
constexpr int kB2Ld = 4;                  // float4 pieces of the new-row score tile the loader holds per lane (2 * rows <= 64 * kB2Ld)
constexpr int kScPitch = kBins + 4;
typedef unsigned int uintx4_t __attribute__((ext_vector_type(4)));
__device__ inline floatx4 as_floatx4(uintx4_t v) { return __builtin_bit_cast(floatx4, v); }

// LDS written by some lanes of this wave is read by others: release / barrier / acquire at wavefront scope (no instruction
// beyond the waits the hardware needs anyway; it pins the compiler's ordering across divergent code)
__device__ inline void wave_lds_handover() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// lane i <- lane i+1; lane 63 <- fill
__device__ inline float wave_shl1(float v, float fill) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), 0x130, 0xf, 0xf, false));
}

// Round 4 -- what shortened the step (measured with tools/ubench/exchange.hip and the in-kernel stamps):
//  (1) The exchange stays inside ONE XCD's L2.  Every cross-XCD form of the 127-value all-reduce over a layer's 24 workgroups
//      costs 1.8-1.9 us per step on an idle chip -- fixed-point atomics (one word or 2-8 replicas per bin), or mailboxes
//      written and read with sc1 -- because each hop is a trip to the memory side.  A layer's workgroups are now launched
//      with block ids that are equal mod 8 (round-robin placement puts them on one XCD; blocks of the other residues exit
//      at once) and exchange through MAILBOXES: every workgroup stores its 127 row sums as 8-byte {mass, step tag} granules
//      into its own 1-KB slot with PLAIN stores (write-through L1, the line stays in that XCD's L2), every workgroup reads
//      all slots with sc1 loads (past its L1, served by that same L2), re-reads a slot until its tags match, and adds the
//      slots in a fixed order: 0.85 us per step in the micro-benchmark.  Placement is never assumed: at launch the
//      workgroups of a layer publish their HW_REG_XCC_ID through sc1 granules (valid at any placement) and only if all are
//      equal use plain stores; otherwise the same mailboxes are written with sc1 stores (cross-XCD coherent, 1.8 us).  The
//      totals are the same fp32 sums in the same order either way, on every workgroup: placement changes speed only.
//  (2) The uniforms arrive as fp32 round-ups (round_up_uniforms_kernel, once per call, off the chain): the loader holds
//      24 instead of 32 registers per set and parks with two b128 stores instead of 8 conversions; it parks BEHIND
//      barrier 3 (in the shadow of the exchange), so the search window holds nothing but the search (0.84 -> 0.52 us).
//  (3) Row phase: the gather table holds BYTE OFFSETS into a score row, empty slots point at a zero word inside the
//      row (fma(val, 0, acc) == acc exactly), boxes without a new row read a zero column of the S'new tile: no clamp,
//      no compare, no select per gathered value.  The point scores are updated in place (a row is private to its
//      wave and LDS executes a wave's operations in order), no ping-pong buffer.
//  Not kept: the cdf as one lane's register chain through LDS (a dependent v_add_f32 costs 9.3 clocks, not 4: 127 of them
//  plus the LDS round trips took 1970 clocks against 1670 for the systolic DPP scan, whose floor is those 9.3 + a hazard nop).
// ======================================================================================================
struct Batch3Smem { int cdf, coarse, part, pos, tabb, box_val, box_row, pb, sc, Snew, uf, Msm, total; };
constexpr int kZeroPoint = kBins;          // word 128 of every score row holds 0.0f (rows have kScPitch = 132 words)
constexpr int kPollWaves = 6;              // waves 0..5 read the mailboxes (wave 7's long-latency input requests must not sit in
                                           // front of a poll in its vmcnt queue; wave 6 carries the writer's extra stores)
constexpr int kPollRound = 4;              // slots per wave and round: 24 workgroups per layer = one round

constexpr int kDmaPar = 4;                // DMA loader: LDS buffers of the S'new tile / the uniforms (steps i .. i+3)
constexpr int kDmaSnew = 64 * kB2Ld * 4;  // floats of one S'new buffer: kB2Ld load instructions of 1 KiB
constexpr int kDmaUf = 64 * 2 * 4;        // floats of one uniforms buffer: two load instructions
__host__ __device__ inline Batch3Smem batch3_smem(int N, int S, int rows, int tabw, int rpw, bool dma = false) {
    Batch3Smem m;
    int o = 0;
    auto take = [&](int n) { int r = o; o += (n + 3) & ~3; return r; };
    m.cdf = take(kBins);
    m.coarse = take(16);
    m.part = take(kPollWaves * kBins);
    m.pos = take(S);
    m.tabb = take(N * tabw);
    m.box_val = take(N);
    m.box_row = take(N);
    m.pb = take(kBins);
    m.sc = take(rpw * kBRows * kScPitch);
    m.Snew = dma ? take(kDmaPar * kDmaSnew) : take(2 * rpw * kBRows * (rows + 1));
    m.uf = dma ? take(kDmaPar * kDmaUf) : take(2 * S);
    m.Msm = take(rpw * kBRows * kMPitch);
    m.total = o;
    return m;
}

// uf[i] = the smallest float >= u[i]:  (double)c < u  <=>  c < uf  for every float c (the search compares in fp32)
__global__ void round_up_uniforms_kernel(const double* __restrict__ u, float* __restrict__ uf, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = u[i];
    float f = (float)x;
    if ((double)f < x) f = __int_as_float(__float_as_int(f) + 1);
    uf[i] = f;
}

hipError_t launch_round_up_uniforms(const double* u, float* uf, long n, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    INFV_LAUNCH(round_up_uniforms_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, u, uf, n);
    return hipGetLastError();
}

// ---- mailboxes of the exchange (ChainBatchArgs.mbox): [2 parities][L][G slots][128 granules of {float mass, uint tag}], then the
// placement handshake [L][G] 16-byte granules {xcc, tag, xcc, tag}.  Granule acc_word(j) of a slot holds bin j, so lane i's
// 16-byte load at byte 16 i returns bins (i, i + 64).  The total of a bin is DEFINED as
//     sum over w = 0..kPollWaves-1 (in order) of [ sum over slots g = w, w + kPollWaves, ... (in order) of mass[g] ]     (fp32)
// -- the order the kernel's waves add in; mailbox_total() restates it for the hand-over kernels.
__host__ __device__ inline long mbox_slot_granules(int L, int G, int parity, int l, int g) { return (((long)parity * L + l) * G + g) * kBins; }
__host__ __device__ inline long mbox_handshake_granules(int L, int G) { return 2L * L * G * kBins; }     // 16-byte units follow at this 8-byte offset
size_t chain_mailbox_bytes(int L, int G) { return (size_t)mbox_handshake_granules(L, G) * 8 + (size_t)L * G * 16; }

__device__ inline float mailbox_total(const unsigned long long* mbox, int L, int G, int parity, int l, int j) {
    float t = 0.f;
    for (int w = 0; w < kPollWaves; ++w) {
        float s = 0.f;
        for (int g = w; g < G; g += kPollWaves) s += __uint_as_float((unsigned int)(mbox[mbox_slot_granules(L, G, parity, l, g) + acc_word(j)] & 0xffffffffull));
        t += s;
    }
    return t;
}

#ifdef INFV_EXPERIMENTS
// part[l][0][j] = total of bin j of the step whose mailboxes have parity `parity` (fast path -> per-call path hand-over)
__global__ void mailbox_to_part_kernel(const unsigned long long* __restrict__ mbox, int L, int G, int parity, int parts_pitch, float* __restrict__ part) {
    const int l = blockIdx.x, j = threadIdx.x;
    if (j >= kBins) return;
    part[((long)l * parts_pitch) * kBins + j] = (j < kBins - 1) ? mailbox_total(mbox, L, G, parity, l, j) : 0.f;
}
hipError_t launch_mailbox_to_part(const unsigned long long* mbox, int n_layers, int G, int parity, int parts_pitch, float* part, hipStream_t stream) {
    INFV_LAUNCH(mailbox_to_part_kernel, dim3(n_layers), dim3(128), 0, stream, mbox, n_layers, G, parity, parts_pitch, part);
    return hipGetLastError();
}
#else
hipError_t launch_mailbox_to_part(const unsigned long long*, int, int, int, int, float*, hipStream_t) { return hipErrorNotSupported; }
#endif

__device__ inline int xcc_id() { int v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 15; }

// MBOX = false: the exchange of rounds 1-3 (fixed-point u64 atomics at the memory side, arrival count in the word, one
//                poller), valid at any placement, linear grid: what the per-sub-batch launches of consolidate() use;
// MBOX = true:  mailboxes inside one XCD's L2 (XCD-aware grid + placement handshake): 0.6 us per step less and no slower
//                beside the other kernels -- but a layer then needs 24 FREE CUs on one XCD at every launch, which in the shared
//                pipeline takes longer than the launch saves (21.5 against 14.7 ms per video): experiments build only
//                (INFV_CHAIN_XCD=1), for a future call-long launch on dedicated CUs.
// DMA = true (round 6): the loader wave moves the S'new tile and the uniforms global -> LDS with `buffer_load_dwordx4 ... lds`
//   (no register sets: the two sets of the register loader and their address arithmetic were 86 of the kernel's 214 registers),
//   four LDS buffers deep (requested three steps ahead).  The kernel then fits 128 registers: two of these workgroups -- or one
//   and a pooling workgroup -- share a CU.  Needs splitk == 1 (a DMA cannot add slabs), per-sub-batch launches, atomics exchange.
#ifndef INFV_CHAIN_WPE
#define INFV_CHAIN_WPE 4
#endif
// (one kernel template, DMA a parameter: the waves-per-SIMD attribute takes a template-dependent argument -- as an inlined body
//  function behind two kernels the register-loader form carried 20 bytes of unused private segment)
template <int RPW, bool MBOX, bool DMA>
__global__ __launch_bounds__(kBNT) __attribute__((amdgpu_waves_per_eu(DMA ? INFV_CHAIN_WPE : 2, DMA ? INFV_CHAIN_WPE : 2)))
void chain_batch3_kernel(ChainBatchArgs a) {
    constexpr int TR = kBRows * RPW;                                   // rows of the tile
    constexpr int PPR = TR / 4;                                        // float4 pieces per new row of the S'new tile
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int N = a.N, H = a.H, Q = a.Q, QS = a.QS;
    const int blocks_per_layer = H * QS;
    // ---- block -> (layer, workgroup of the layer).  XCD-aware launches (a.xcd_grid): grid = 8 * blocks_per_layer, block b sits
    // in placement class b % 8 (blocks of one class share an XCD under round-robin placement); layer l is served by class
    // (8 l) / L, the other classes have nothing to do.
    int l, blk;
    if (MBOX && a.xcd_grid) {
        const int cls = blockIdx.x & 7;
        l = -1;
        for (int ll = 0; ll < a.L; ++ll) if ((8 * ll) / a.L == cls) l = ll;
        if (l < 0) return;
        blk = blockIdx.x >> 3;
    } else {
        l = blockIdx.x / blocks_per_layer;
        blk = blockIdx.x - l * blocks_per_layer;
    }
    if (!(a.exp_flags & 1)) __builtin_amdgcn_s_setprio(3);
    wg_stamp_begin(a.wg_stamps);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (a.dbg != nullptr && l == 0 && blk == 0 && tid == 0) a.dbg[18] = wall_clock64();      // (stamps: kernel entry of workgroup 0)
    const int rows = a.op.rows, tabw = a.op.tabw;
    const Batch3Smem m = batch3_smem(N, a.S, rows, tabw, RPW, DMA);
    const int h = blk % H, qs = blk / H;
    const int b = l * blocks_per_layer + blk;                          // (stamps: workgroup 0 = layer 0, head 0, tile 0)
    const int G = blocks_per_layer;
    const int sn = rows + 1, sn_tile = TR * sn;
    float* cdf = lds + m.cdf;
    float* coarse = lds + m.coarse;
    float* part = lds + m.part;
    int32_t* tabb = reinterpret_cast<int32_t*>(lds + m.tabb);          // byte offset (into a score row) of the k-th resampled slot of box n
    int32_t* pb = reinterpret_cast<int32_t*>(lds + m.pb);
    const float* box_val = lds + m.box_val;
    const int32_t* box_row = reinterpret_cast<const int32_t*>(lds + m.box_row);
    float* Msm = lds + m.Msm;
    const long tile = (((long)l * H + h) * Q + qs * TR);
    const int valid = min(TR, Q - qs * TR);
    const bool writer = (h == 0 && qs == 0);
    const bool loader = wave == kBRows - 1;
    const bool poller = wave < kPollWaves;
    const long tile_snew = (long)rows * a.snew_ld;

    // ---- placement handshake, first half: publish this workgroup's XCD (sc1: visible at any placement) ----
    const unsigned int launch_tag = (unsigned int)(a.step0 + 1);
    uintx4_t* hs = MBOX ? reinterpret_cast<uintx4_t*>(a.mbox + mbox_handshake_granules(a.L, G)) + (long)l * G : nullptr;
    if (MBOX && wave == 0 && lane == 0) {
        const unsigned int x = (unsigned int)xcc_id();
        const uintx4_t v = {x, launch_tag, x, launch_tag};
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(hs, 0, G * 16, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, blk * 16, 0, 16 /* sc1 */);
    }

    // ---- one-time set-up ----
    {
        int32_t* pos = reinterpret_cast<int32_t*>(lds + m.pos);
        if (tid < N) {
            (lds + m.box_val)[tid] = a.op.box_val[tid];
            reinterpret_cast<int32_t*>(lds + m.box_row)[tid] = a.op.box_row[tid];
        }
        if (tid < kBins) pb[tid] = a.st.bin_box[tid];
        if (tid < a.S) pos[tid] = -1;
        for (int e = tid; e < (DMA ? kDmaPar * kDmaSnew : 2 * sn_tile); e += kBNT) (lds + m.Snew)[e] = 0.f;     // incl. the zero column (index rows) of every row
        if (DMA) for (int e = tid; e < kDmaPar * kDmaUf; e += kBNT) (lds + m.uf)[e] = 2.f;                       // (a uniform nobody draws with)
        for (int e = tid; e < TR * kScPitch; e += kBNT) (lds + m.sc)[e] = 0.f;     // incl. the zero point of every row
        __syncthreads();
        for (int e = tid; e < N * tabw; e += kBNT) {
            const int sl = a.op.slot_tab[e];
            tabb[e] = 4 * kZeroPoint;
            if (sl >= 0) pos[sl] = e;
        }
        __syncthreads();
    }
    const int my_pos = (tid < a.S) ? reinterpret_cast<const int32_t*>(lds + m.pos)[tid] : -1;
    bool row_ok[RPW];
#pragma unroll
    for (int j = 0; j < RPW; ++j) row_ok[j] = wave + kBRows * j < valid;
    float* scw = lds + m.sc + wave * kScPitch;                         // this wave's first row of point scores (row j: + j * kBRows * kScPitch)
    // the two points of this lane: boxes, operator entries, edge validity and spacings (static)
    const int n0 = pb[lane], n1 = pb[lane + 64];
    const float val0 = box_val[n0], val1 = box_val[n1];
    const int br0 = box_row[n0] >= 0 ? box_row[n0] : rows, br1 = box_row[n1] >= 0 ? box_row[n1] : rows;   // `rows` = the zero column
    // DMA layout of the S'new tile: float4 slot (hf, nr) = hf * rows + nr holds tile rows 4 hf .. 4 hf + 3 of new row nr
    const bool has_r0 = br0 < rows, has_r1 = br1 < rows;
    const int sna0 = has_r0 ? 4 * br0 : 0, sna1 = has_r1 ? 4 * br1 : 0;
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
        scw[j * kBRows * kScPitch + lane] = i0;
        scw[j * kBRows * kScPitch + lane + 64] = i1;
        if (a.publish_init && row_ok[j]) {
            // first launch of a call: the state before its first step, for the rows alpha_rows2_kernel rebuilds of that step
            const long prev = (a.step0 % a.ring == 0) ? a.ring - 1 : a.step0 % a.ring - 1;
            float* cr = a.crit_ring + prev * a.crit_slot + (tile + row) * kBins;
            cr[lane] = i0;
            cr[lane + 64] = i1;
        }
    }

    // ---- loader (wave 7): the S'new tile and the rounded-up uniforms of a step in registers, two sets (steps i+1, i+2 in
    // flight).  Wide loads only.  S'new tile: new row nr holds this tile's TR scores contiguously -> float4 e4 = lane + 64 k:
    // row e4 / PPR, piece e4 % PPR;   uniforms: S floats -> two float4 per lane
    struct LdSet { floatx4 sn[DMA ? 1 : kB2Ld]; floatx4 u[DMA ? 1 : 2]; };
    LdSet ldA, ldB;
    // ---- DMA loader: per-lane global byte offsets of this lane's float4 of each load instruction (an offset beyond the buffer's
    // range reads as zero), the same for every step; the step selects the buffer through the scalar offset
    typedef int v4i_t __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    int dvo[kB2Ld] = {-1, -1, -1, -1}, dvu[2] = {-1, -1};
    v4i_t drs_sn = {0, 0, 0, 0}, drs_uf = {0, 0, 0, 0};
    unsigned dl_sn = 0, dl_uf = 0;
    if constexpr (DMA) {
#pragma unroll
        for (int k = 0; k < kB2Ld; ++k) {
            const int e4 = lane + 64 * k;
            const int hf = e4 / rows, nr = e4 - hf * rows;
            dvo[k] = (hf < PPR && 4 * hf < valid) ? (nr * a.snew_ld + 4 * hf) * 4 : -1;
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) { const int s4 = lane + 64 * k; dvu[k] = (4 * s4 < a.S) ? 16 * s4 : -1; }
        const unsigned long ps = reinterpret_cast<unsigned long>(a.Snew + tile), pu = reinterpret_cast<unsigned long>(a.uf + (long)l * a.S);
        drs_sn = v4i_t{(int)(unsigned)ps, (int)(unsigned)((ps >> 32) & 0xffffu), 0x7fffffff, 0x00020000};
        drs_uf = v4i_t{(int)(unsigned)pu, (int)(unsigned)((pu >> 32) & 0xffffu), 0x7fffffff, 0x00020000};
        dl_sn = (unsigned)(unsigned long)(lds_ptr_t)(lds + m.Snew);
        dl_uf = (unsigned)(unsigned long)(lds_ptr_t)(lds + m.uf);
    }
    auto dma16 = [](const v4i_t& rsrc, unsigned lds_addr, int voff, int soff) {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                     :: "s"(__builtin_amdgcn_readfirstlane(lds_addr)), "v"(voff), "s"(rsrc), "s"(__builtin_amdgcn_readfirstlane(soff)) : "memory");
    };
    // request step i into buffer i % kDmaPar: six load instructions, no registers
    auto dma_request = [&](int i) {
        const int buf = i & (kDmaPar - 1);
        const int so_sn = (int)((long)i * tile_snew * 4), so_uf = (int)((long)i * a.L * a.S * 4);     // (below 2^31: launcher)
#pragma unroll
        for (int k = 0; k < kB2Ld; ++k) dma16(drs_sn, dl_sn + (buf * kDmaSnew + k * 256) * 4, dvo[k], so_sn);
#pragma unroll
        for (int k = 0; k < 2; ++k) dma16(drs_uf, dl_uf + (buf * kDmaUf + k * 256) * 4, dvu[k], so_uf);
    };
    // Call-long launch: step i belongs to sub-batch i / call_sub, whose S'new rows sit in workspace set (sub-batch % call_sets) once
    // the GEMM stream has raised `ready` past it.  The rows were written by ANOTHER kernel while this one was running, possibly
    // over lines this XCD's L2 still holds from the set's previous use: every load of them is an sc1 load (served by the
    // memory side; the producer's end-of-kernel release has written them back before its flag kernel ran).
    const bool call_long = a.call != nullptr;
    unsigned int ready_seen = 0;                                              // (loader wave) sub-batches known to be projected
    auto ld_request = [&](int i, LdSet& r) {
        const float* sb;
        int splitk = a.snew_splitk;
        long split_stride = a.snew_split_stride;
        if (call_long) {
            const ChainCallDesc* cd = a.call;                                 // (uniform address: scalar loads)
            const int cb = i / a.call_sub, li = i - cb * a.call_sub;
            if ((unsigned int)cb >= ready_seen) {
                // sub-batch cb is projected: its S' tiles are counted in (call-long GEMM), or the per-sub-batch GEMM's flag has passed it
                const bool tiled = cd->tiles_s != nullptr && cb < cd->n_tiled;
                const unsigned int* ready = tiled ? cd->tiles_s + cb : cd->ready;
                const unsigned int need = tiled ? (unsigned int)(cb == cd->n_tiled - 1 ? cd->tiles_last : cd->tiles_full) : (unsigned int)cb + 1u;
                long long* stats = (b == 0) ? cd->stats : nullptr;
                long long t0 = 0;
                int spins = 0;
                for (;;) {
                    const unsigned int v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    if (v >= need) { ready_seen = tiled ? (unsigned int)cb + 1u : v; break; }
                    if (spins == 0 && stats != nullptr) t0 = wall_clock64();
                    __builtin_amdgcn_s_sleep(8);
                    if (++spins > a.spin_limit) { __hip_atomic_store(a.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); ready_seen = 0xffffffffu; break; }   // (failed: no further waits)
                }
                if (spins > 0 && stats != nullptr && lane == 0) { stats[0] += wall_clock64() - t0; stats[1] += 1; }
            }
            sb = cd->snew_set[cb % cd->n_sets] + (long)li * tile_snew + tile;
            if (cb == cd->n_batches - 1) { splitk = cd->sk_last; split_stride = cd->ss_last; }
        } else {
            sb = a.Snew + (long)i * tile_snew + tile;
        }
        // one buffer resource per split-K slab, its 64-bit base at the slab (a 32-bit slab offset overflowed for short sub-batches of
        // large models: the hardware then returns zeros, silently); the offset inside a slab, rows * snew_ld * 4 bytes, is bounded by
        // the launcher.  sc1 only on the call-long launch, whose rows another kernel writes while this one is resident.
        // (slab 0 on its own: its loads stay in flight behind the request; the further slabs of a short sub-batch are added behind them)
        {
            __amdgpu_buffer_rsrc_t rsn = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sb), 0, 0x7fffffff, 0x00020000);
#pragma unroll
            for (int k = 0; k < kB2Ld; ++k) {
                const int e4 = lane + 64 * k;
                const int nr = e4 / PPR, hf = e4 % PPR;
                floatx4 v = {0.f, 0.f, 0.f, 0.f};
                if (nr < rows && 4 * hf < valid) {
                    const int off = (nr * a.snew_ld + 4 * hf) * 4;
                    v = call_long ? as_floatx4(__builtin_amdgcn_raw_buffer_load_b128(rsn, off, 0, 16 /* sc1 */))
                                  : as_floatx4(__builtin_amdgcn_raw_buffer_load_b128(rsn, off, 0, 0));
                }
                r.sn[k] = v;
            }
        }
        for (int x = 1; x < splitk; ++x) {
            __amdgpu_buffer_rsrc_t rsn = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sb + (long)x * split_stride), 0, 0x7fffffff, 0x00020000);
#pragma unroll
            for (int k = 0; k < kB2Ld; ++k) {
                const int e4 = lane + 64 * k;
                const int nr = e4 / PPR, hf = e4 % PPR;
                if (nr < rows && 4 * hf < valid) {
                    const int off = (nr * a.snew_ld + 4 * hf) * 4;
                    r.sn[k] += call_long ? as_floatx4(__builtin_amdgcn_raw_buffer_load_b128(rsn, off, 0, 16 /* sc1 */))
                                         : as_floatx4(__builtin_amdgcn_raw_buffer_load_b128(rsn, off, 0, 0));
                }
            }
        }
        const floatx4* ub = reinterpret_cast<const floatx4*>(a.uf + ((long)i * a.L + l) * a.S);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int s4 = lane + 64 * k;
            const floatx4 two = {2.f, 2.f, 2.f, 2.f};
            r.u[k] = (4 * s4 < a.S) ? ub[s4] : two;
        }
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
        for (int k = 0; k < 2; ++k) {
            const int s4 = lane + 64 * k;
            if (4 * s4 < a.S) *reinterpret_cast<floatx4*>(&uf[4 * s4]) = r.u[k];
        }
    };
    if (loader) {
        if constexpr (DMA) {
            // steps 0, 1, 2 in flight; step 0 has landed before the barrier below hands it to the other waves
            dma_request(0);
            if (a.n_steps > 1) dma_request(1);
            if (a.n_steps > 2) dma_request(2);
            if (a.n_steps > 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if (a.n_steps > 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
        ld_request(0, ldA);
        ld_park(0, ldA);
        if (a.n_steps > 1) ld_request(1, ldB);                         // odd steps live in set B, even steps in set A
        if (a.n_steps > 2) ld_request(2, ldA);
        }
    }

    // ---- placement handshake, second half: one XCD for the whole layer?  (bounded wait; needs every workgroup resident, as
    // the chain itself does) ----
    bool plain = false;                                                       // plain mailbox stores: the layer's L2 is one
    if constexpr (MBOX) {
    if (wave == 0) {
        bool same = false;
        if (G <= 64 && a.xcd_grid && !(a.exp_flags & 32)) {
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(hs, 0, G * 16, 0x00020000);
            uintx4_t v = {0u, 0u, 0u, 0u};
            int spins = 0;
            for (;;) {
                if (lane < G) v = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, 0, 16 /* sc1 */);
                if (!__any(lane < G && (v.y != launch_tag || v.w != launch_tag))) break;
                __builtin_amdgcn_s_sleep(8);
                if (++spins > a.spin_limit) { __hip_atomic_store(a.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
            }
            const unsigned int mine = __builtin_amdgcn_readfirstlane(v.x);
            same = !__any(lane < G && (v.x != mine || v.z != mine));
        }
        if (lane == 0) reinterpret_cast<int32_t*>(coarse)[0] = same ? 1 : 0;
    }
    __syncthreads();
    plain = reinterpret_cast<const int32_t*>(coarse)[0] != 0;
    __syncthreads();                                                          // (coarse is rewritten by wave 0 in step 0)
    if (a.xcc_report != nullptr && tid == 0) a.xcc_report[b] = (plain ? 0x100 : 0) | xcc_id();
    }

    const unsigned long long* mbox = a.mbox;
    // poll registers of waves 0..5: this wave's slots of the first round (slot g = wave + kPollWaves k)
    uintx4_t pv[kPollRound];
#pragma unroll
    for (int k = 0; k < kPollRound; ++k) pv[k] = uintx4_t{0u, 0u, 0u, 0u};
    auto poll_issue = [&](int parity, int round) {
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned long long*>(mbox + mbox_slot_granules(a.L, G, parity, l, 0)), 0,
                                                                      G * kBins * 8, 0x00020000);
#pragma unroll
        for (int k = 0; k < kPollRound; ++k) {
            const int g = wave + kPollWaves * (kPollRound * round + k);
            if (g < G) pv[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, g * kBins * 8, 16 /* sc1 */);
        }
    };
    // wait for this wave's slots to carry `tag`, add them in slot order, park the partial sums for wave 0
    auto poll_collect = [&](int parity, unsigned int tag, bool wait) {
        float s0 = 0.f, s1 = 0.f;
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned long long*>(mbox + mbox_slot_granules(a.L, G, parity, l, 0)), 0,
                                                                      G * kBins * 8, 0x00020000);
        for (int round = 0; wave + kPollWaves * kPollRound * round < G; ++round) {
            if (round > 0) poll_issue(parity, round);
#pragma unroll
            for (int k = 0; k < kPollRound; ++k) {
                const int g = wave + kPollWaves * (kPollRound * round + k);
                if (g < G) {
                    int spins = 0;
                    while (wait && __any(pv[k].y != tag || pv[k].w != tag)) {
                        if (a.exp_flags & 2) __builtin_amdgcn_s_sleep(16); else __builtin_amdgcn_s_sleep(1);
                        if (++spins > a.spin_limit) { __hip_atomic_store(a.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
                        pv[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, g * kBins * 8, 16 /* sc1 */);
                    }
                    s0 += __uint_as_float(pv[k].x);
                    s1 += __uint_as_float(pv[k].z);
                }
            }
        }
        part[wave * kBins + lane] = s0;
        part[wave * kBins + lane + 64] = s1;
    };

    // ---- publish step `pi` (ring slot `pslot`) for alpha_rows2_kernel / the UC kernel: waves 1-6, stores only.  Runs in the shadow
    // of wave 0's scan of the NEXT step (behind its barrier 0), or after the loop for the last step.  pa0 / pa1: the publishing
    // wave's own point scores of that step.
    auto publish_step = [&](int pi, long pslot, const float (&pa0)[RPW], const float (&pa1)[RPW]) {
        const bool plast = (pi == a.n_steps - 1);
        if (writer) {
            int32_t* tab_out = a.tab_ring + pslot * a.tab_slot + (long)l * N * tabw;          // source BOX of every slot (UC kernel)
            int32_t* tabb_out = a.tabb_ring + pslot * a.tab_slot + (long)l * N * tabw;        // drawn BIN of every slot
            for (int e = tid - 64; e < N * tabw; e += kBNT - 128) {
                const int bb = tabb[e] >> 2;
                const bool none = bb == kZeroPoint;
                tabb_out[e] = none ? -1 : bb;
                tab_out[e] = none ? -1 : pb[bb];
            }
            if (wave == 6 && a.S > 448) {                                                    // wave 7's share of the draw diagnostics
                const int s7 = 448 + lane;
                const int lo = __float_as_int((lds + m.pos)[s7]);
                if (s7 < a.S) {
                    if (plast) { a.bins_out[(long)l * a.S + s7] = lo; a.idx_out[(long)l * a.S + s7] = pb[lo]; }
                    if (a.bins_tr != nullptr && pi < a.trace_steps) a.bins_tr[((long)pi * a.L + l) * a.S + s7] = lo;
                }
            }
        }
        // point scores after the step: own rows from registers; wave 1 also wave 0's rows, wave 6 also wave 7's (from LDS)
        float* cr = a.crit_ring + pslot * a.crit_slot + tile * kBins;
#pragma unroll
        for (int j = 0; j < RPW; ++j) {
            const int row = wave + kBRows * j;
            if constexpr (DMA) {
                // (own rows from LDS as well: they hold the step's scores until barrier 2 of the next step; no registers live across the step)
                if (row_ok[j]) { cr[row * kBins + lane] = scw[j * kBRows * kScPitch + lane]; cr[row * kBins + lane + 64] = scw[j * kBRows * kScPitch + lane + 64]; }
            } else
            if (row_ok[j] && !(a.exp_flags & 4)) { cr[row * kBins + lane] = pa0[j]; cr[row * kBins + lane + 64] = pa1[j]; }
            const int extra = (wave == 1) ? kBRows * j : ((wave == 6) ? kBRows * j + kBRows - 1 : -1);
            if (extra >= 0 && extra < valid) {
                const float* sx = lds + m.sc + extra * kScPitch;
                cr[extra * kBins + lane] = sx[lane];
                cr[extra * kBins + lane + 64] = sx[lane + 64];
            }
        }
    };
    float acc0[RPW], acc1[RPW];                                            // this wave's point scores of the step (kept for publish_step)
#pragma unroll
    for (int j = 0; j < RPW; ++j) acc0[j] = acc1[j] = 0.f;
    long prev_slot = 0;

    int slot_run = (int)(a.step0 % a.ring);
    const int ring_n = (int)a.ring;
    int cb_next = call_long ? a.call_sub : 0;                              // first step of the next sub-batch (call-long launch)
    bool sig_pending = false;                                              // (wave 6) an L2 write-back of a finished sub-batch is in flight
    const bool ovr0 = ((a.override_mask >> l) & 1u) != 0;
    // step 0 takes its totals from elsewhere than the exchange (mailboxes: also when the step before ran in a per-chunk launch)
    const bool special0 = ovr0 || a.first_from_parts || (MBOX && a.first_from_acc);
    // atomics exchange: ring of three accumulator slots (read g-1 / add g / clear g+1), wave 0 polls one 16-byte word pair per lane
    int g3 = (int)(a.step0 % 3);
    const long layer_words = (long)kAccShards * kBins;
    const unsigned long long need_full = (unsigned long long)(blocks_per_layer + a.expect_extra);
    uintx4_t pw = {0u, 0u, 0u, 0u};
    auto poll_pair = [&](const unsigned long long* layer_words_p) {
        // lane i reads words (2 i, 2 i + 1) = bins (i, i + 64) with ONE 16-byte load that bypasses the vector L1 (sc1); the words
        // are only ever written by device-scope atomics, which execute at the memory side and leave no line behind in any L2
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned long long*>(layer_words_p), 0, kBins * 8, 0x00020000);
        pw = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, 0, 16 /* sc1 */);
    };
    if constexpr (MBOX) {
        if (poller && !special0) poll_issue((int)((a.step0 + 1) & 1), 0);   // the mailboxes of global step step0 - 1
    } else {
        if (wave == 0 && !special0) poll_pair(a.acc[g3 == 0 ? 2 : g3 - 1] + l * layer_words);
    }
    __syncthreads();

#define B3STAMP(k) do { if (a.dbg != nullptr && b == 0 && tid == 0 && i == 5) { a.dbg[k] = wall_clock64(); a.dbg[8 + k] = clock64(); } } while (0)
    if (a.dbg != nullptr && b == 0 && tid == 0) a.dbg[19] = wall_clock64();                 // (stamps: set-up done)
    for (int i = 0; i < a.n_steps; ++i) {
        B3STAMP(0);
        // (stamps 16 / 17: top of step 1 and of the last step -- the launch's average step without the per-phase stamps' own cost)
        if (a.dbg != nullptr && b == 0 && tid == 0 && (i == 1 || i == a.n_steps - 1)) a.dbg[i == 1 ? 16 : 17] = wall_clock64();
        const long gstep = a.step0 + i;                                         // global index of this step in the call
        const long slot = slot_run;
        if (++slot_run == ring_n) slot_run = 0;
        const bool last = (i == a.n_steps - 1);
        const float* Snew = lds + m.Snew + (DMA ? (i & (kDmaPar - 1)) * kDmaSnew : (i & 1) * sn_tile);
        unsigned long long* acc_prev = a.acc[g3 == 0 ? 2 : g3 - 1] + l * layer_words;
        unsigned long long* acc_cur = a.acc[g3] + l * layer_words;
        unsigned long long* acc_clr = a.acc[g3 == 2 ? 0 : g3 + 1] + l * layer_words;
        if (++g3 == 3) g3 = 0;
        const bool from_box = MBOX && !(i == 0 && special0);
        // ---- waves 0..5: the previous step's mailboxes (polled one step ahead) -> partial sums; fault injection: a slot nobody fills
        if (from_box) {
            if (poller) poll_collect((int)((gstep + 1) & 1), (unsigned int)gstep, i > 0 && !(a.exp_flags & 16));
            if (a.expect_extra > 0 && i > 0 && wave == 0) {
                for (int spins = 0; spins <= a.spin_limit; ++spins) __builtin_amdgcn_s_sleep(1);
                __hip_atomic_store(a.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            __syncthreads();                                                     // barrier 0
        }
        // call-long launch: step i opens a new sub-batch -> the steps of the one before are all published once step i-1 is
        const bool batch_open = call_long && i > 0 && i == cb_next;
        if (batch_open) cb_next += a.call_sub;
        if (i > 0) {
            // in the shadow of wave 0's scan: the previous step goes out, the loader asks for the inputs of step i+2
            if (loader) {
                if constexpr (DMA) { if (i + 2 < a.n_steps) dma_request(i + 2); }      // (buffer of step i - 2: its last readers ran before barrier 3 of that step)
                else if (i + 2 < a.n_steps && !(a.exp_flags & 8)) { if ((i + 2) & 1) ld_request(i + 2, ldB); else ld_request(i + 2, ldA); }
            } else if (wave != 0) {
                publish_step(i - 1, prev_slot, acc0, acc1);
                // hand a finished sub-batch to the UC stream.  Every publishing wave drains its stores (they are then in this XCD's
                // L2); behind barrier 1 wave 6 starts the L2 write-back and, a step later, waits for it and counts this workgroup in.
                if (wave == 6 && sig_pending) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // the write-back issued one step ago has completed
                    if (lane == 0) __hip_atomic_fetch_add(a.call->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    sig_pending = false;
                }
                if (batch_open) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        prev_slot = slot;
        // ---- wave 0: probabilities -> cdf ----
        if (wave == 0) {
            constexpr int nb = kBins - 1;
            const int j1 = lane + 64;
            float p0, p1;
            if (i == 0 && ovr0) {
                p0 = a.probs_override[l * kBins + lane];
                p1 = (j1 < nb) ? a.probs_override[l * kBins + j1] : 0.f;
            } else {
                float raw0, raw1;
                if (i == 0 && a.first_from_parts) {
                    double a0 = 0.0, a1 = 0.0;
                    for (int p = 0; p < a.parts; ++p) {
                        a0 += (double)a.part_prev[((long)l * a.parts + p) * kBins + lane];
                        if (j1 < nb) a1 += (double)a.part_prev[((long)l * a.parts + p) * kBins + j1];
                    }
                    raw0 = (float)a0; raw1 = (float)a1;
                } else if (MBOX && i == 0 && a.first_from_acc) {
                    // the previous step ran in a per-chunk launch: its fixed-point totals (complete: kernel boundary)
                    raw0 = (float)mass_of(acc_prev[acc_word(lane)]);
                    raw1 = (j1 < nb) ? (float)mass_of(acc_prev[acc_word(j1)]) : 0.f;
                } else if (!MBOX) {
                    // the poll was issued one step ago; step 0 reads totals completed by an earlier launch (no count)
                    const unsigned long long need = (i > 0 && !(a.exp_flags & 16)) ? need_full : 0ull;
                    int spins = 0;
                    unsigned long long w0, w1;
                    for (;;) {
                        w0 = ((unsigned long long)pw.y << 32) | pw.x;
                        w1 = ((unsigned long long)pw.w << 32) | pw.z;
                        if (!((w0 >> kArriveShift) < need || (j1 < nb && (w1 >> kArriveShift) < need))) break;
                        if (a.exp_flags & 2) __builtin_amdgcn_s_sleep(16); else __builtin_amdgcn_s_sleep(1);
                        if (++spins > a.spin_limit) { __hip_atomic_store(a.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
                        poll_pair(acc_prev);
                    }
                    raw0 = (float)mass_of(w0);
                    raw1 = (j1 < nb) ? (float)mass_of(w1) : 0.f;
                } else {
                    float t0 = 0.f, t1 = 0.f;
#pragma unroll
                    for (int w = 0; w < kPollWaves; ++w) { t0 += part[w * kBins + lane]; t1 += part[w * kBins + j1]; }
                    raw0 = t0; raw1 = t1;
                }
                if (j1 >= nb) raw1 = 0.f;
                const float tot1 = (float)wave_sum_f64((double)raw0 + (double)raw1);
                const float q0 = raw0 / tot1, q1 = raw1 / tot1;                       // LTM.py:203
                const float tot2 = (float)wave_sum_f64((double)q0 + (double)q1);
                p0 = q0 / tot2; p1 = q1 / tot2;                                       // Categorical's own normalisation
            }
            if (j1 >= nb) p1 = 0.f;
            B3STAMP(1);
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
        B3STAMP(2);
        if (batch_open && wave == 6) { asm volatile("buffer_wbl2 sc1" ::: "memory"); sig_pending = true; }
        if constexpr (!MBOX) {
            // Clear the accumulator slot of step i+1 (it held the totals of step i-2).  Wave 0 has just seen every arrival of
            // step i-1, and a workgroup arrives only after its own poll of step i-2's totals, so nobody reads the slot any more;
            // nobody adds to it before having seen all arrivals of step i, this workgroup's included -- and that arrival (behind
            // barrier 3) is held back until the clear has been acknowledged: the clearing waves drain vmcnt before barrier 3.
            if (writer && tid >= 128 && tid < 128 + kBins) atomicExch(acc_clr + (tid - 128), 0ull);
        }
        if (tid < a.S) {
            // ---- lower bound of this thread's uniform in the cdf == number of entries below it ----
            const float my_uf = (lds + m.uf + (DMA ? (i & (kDmaPar - 1)) * kDmaUf : (i & 1) * a.S))[tid];
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
            if (my_pos >= 0) tabb[my_pos] = 4 * lo;
            if (writer && !loader) {
                if (last) { a.bins_out[(long)l * a.S + tid] = lo; a.idx_out[(long)l * a.S + tid] = pb[lo]; }
                if (a.bins_tr != nullptr && i < a.trace_steps) a.bins_tr[((long)i * a.L + l) * a.S + tid] = lo;
            }
            if (writer && loader) (lds + m.pos)[tid] = __int_as_float(lo);       // wave 7 stores nothing: wave 6 writes its 64 bins out
        }
        __syncthreads();                                                         // barrier 2
        B3STAMP(3);
        // ---- wave = rows: recurrence of the 128 point scores of each of its rows, edge densities, bin masses (registers + DPP) ----
        {
#pragma unroll
            for (int j = 0; j < RPW; ++j) acc0[j] = acc1[j] = 0.f;
            const char* scb = reinterpret_cast<const char*>(scw);
            for (int k0 = 0; k0 < tabw; k0 += 4) {
                const int4 s0 = *reinterpret_cast<const int4*>(&tabb[n0 * tabw + k0]);
                const int4 s1 = *reinterpret_cast<const int4*>(&tabb[n1 * tabw + k0]);
#pragma unroll
                for (int j = 0; j < RPW; ++j) {
                    const char* sc = scb + j * kBRows * kScPitch * 4;
                    const float v0 = *reinterpret_cast<const float*>(sc + s0.x), v1 = *reinterpret_cast<const float*>(sc + s0.y);
                    const float v2 = *reinterpret_cast<const float*>(sc + s0.z), v3 = *reinterpret_cast<const float*>(sc + s0.w);
                    const float w0 = *reinterpret_cast<const float*>(sc + s1.x), w1 = *reinterpret_cast<const float*>(sc + s1.y);
                    const float w2 = *reinterpret_cast<const float*>(sc + s1.z), w3 = *reinterpret_cast<const float*>(sc + s1.w);
                    acc0[j] = fmaf(val0, v0, acc0[j]); acc0[j] = fmaf(val0, v1, acc0[j]);
                    acc0[j] = fmaf(val0, v2, acc0[j]); acc0[j] = fmaf(val0, v3, acc0[j]);
                    acc1[j] = fmaf(val1, w0, acc1[j]); acc1[j] = fmaf(val1, w1, acc1[j]);
                    acc1[j] = fmaf(val1, w2, acc1[j]); acc1[j] = fmaf(val1, w3, acc1[j]);
                }
            }
            // the rows of a wave are independent dependency chains: written phase by phase so that their LDS reads, DPP
            // reductions and transcendentals interleave (same arithmetic per row as one row at a time)
            float es0[RPW], es1[RPW], mx[RPW];
#pragma unroll
            for (int j = 0; j < RPW; ++j) {
                const int row = wave + kBRows * j;
                if constexpr (DMA) {
                    const float* sr = Snew + (row >> 2) * rows * 4 + (row & 3);
                    const float x0 = sr[sna0], x1 = sr[sna1];
                    acc0[j] += has_r0 ? x0 : 0.f;
                    acc1[j] += has_r1 ? x1 : 0.f;
                } else {
                acc0[j] += Snew[row * sn + br0];
                acc1[j] += Snew[row * sn + br1];
                }
                // in place: every gather of this wave is issued before these stores (the lanes read each other's old values),
                // and LDS runs a wave's operations in order
                if (j == 0) wave_lds_handover();
                scw[j * kBRows * kScPitch + lane] = acc0[j];
                scw[j * kBRows * kScPitch + lane + 64] = acc1[j];
                // densities at the 129 edges: edge j (1..127) sits in the box of point j, edges 0 and 128 in none (score 0)
                es0[j] = e0ok ? acc0[j] + cqr[j] : 0.f;
                es1[j] = e1ok ? acc1[j] + cqr[j] : 0.f;
                mx[j] = fmaxf(e0ok ? es0[j] : -INFINITY, e1ok ? es1[j] : -INFINITY);
            }
#pragma unroll
            for (int j = 0; j < RPW; ++j) mx[j] = fmaxf(wave_max(mx[j]), 0.f);
            float d0[RPW], d1[RPW], d128[RPW];
#pragma unroll
            // (hardware exp2 / reciprocal: the masses feed the draw through a sum over 384 rows and two normalisations; their
            //  ~1e-6 relative error is a twentieth of the 2e-5 the probabilities are held to, and 45 of the row phase's ~200
            //  instructions per row go -- the phase is issue-bound, two waves per SIMD)
            for (int j = 0; j < RPW; ++j) { d0[j] = __expf(es0[j] - mx[j]); d1[j] = __expf(es1[j] - mx[j]); d128[j] = __expf(0.f - mx[j]); }
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
                const float inv_z = __builtin_amdgcn_rcpf(zz[j]);
                // mass of interval j+1 -> bin j (cum[j+1]-cum[j], LTM.py:201-202): lanes take j = lane and lane+64 (< 127); the
                // two masses of a lane go out as one 8-byte LDS word (what the row sum below reads)
                float2 mm;
                mm.x = row_ok[j] ? ((d0n[j] * inv_z + d0nn[j] * inv_z) * dxa) * 0.5f : 0.f;
                mm.y = (row_ok[j] && lane + 64 < kBins - 1) ? ((d1n[j] * inv_z + d1nn[j] * inv_z) * dxb) * 0.5f : 0.f;
                *reinterpret_cast<float2*>(&Msm[row * kMPitch + 2 * lane]) = mm;
            }
        }
        if (!MBOX && writer && (wave == 2 || wave == 3)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clear above is complete
        __syncthreads();                                                         // barrier 3
        B3STAMP(4);
        if constexpr (!MBOX) {
            if (tid < kBins - 1) {
                float mr[TR];
#pragma unroll
                for (int r = 0; r < TR; ++r) mr[r] = Msm[r * kMPitch + acc_word(tid)];
#pragma unroll
                for (int w = TR / 2; w >= 1; w >>= 1)                   // pairwise: log2(TR) dependent adds instead of TR
#pragma unroll
                    for (int r = 0; r < w; ++r) mr[r] += mr[r + w];
                const float t = mr[0];
                if (!(a.exp_flags & 16))
                    atomicAdd(&acc_cur[acc_word(tid)], (unsigned long long)((double)t * kMassScale + 0.5) + (1ull << kArriveShift));
            }
            // re-arm the poll behind the deposit and go round: this wave issues nothing else until it has read it.  A poll that
            // reaches the memory side before the slowest workgroup's adds returns an incomplete count and costs a second round
            // trip, so the first poll is held back by poll_delay x 64 clocks (tuned in situ: chain_ab.sh)
            if (wave == 0 && !last && !(a.exp_flags & 16)) {
                for (int dly = 0; dly < a.poll_delay; ++dly) __builtin_amdgcn_s_sleep(1);
                poll_pair(acc_cur);
            }
        } else if (wave == 0) {
            // ---- this workgroup's row sums of the step -> its mailbox: one 16-byte store per lane = two {mass, tag} granules
            float t0 = 0.f, t1 = 0.f;
#pragma unroll
            for (int r = 0; r < TR; ++r) {
                const float2 mm = *reinterpret_cast<const float2*>(&Msm[r * kMPitch + 2 * lane]);
                t0 += mm.x; t1 += mm.y;
            }
            if (!(a.exp_flags & 16)) {
                const unsigned int tag = (unsigned int)(gstep + 1);
                const uintx4_t v = {__float_as_uint(t0), tag, __float_as_uint(t1), tag};
                unsigned long long* dst = a.mbox + mbox_slot_granules(a.L, G, (int)(gstep & 1), l, blk);
                if (plain) {
                    reinterpret_cast<uintx4_t*>(dst)[lane] = v;                 // stays in this XCD's L2, where the whole layer reads it
                } else {
                    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(dst, 0, kBins * 8, 0x00020000);
                    __builtin_amdgcn_raw_buffer_store_b128(v, rs, lane * 16, 0, 16 /* sc1: write-through, valid at any placement */);
                }
            }
        }
        // re-arm the polls right behind the deposit.  Nothing else touches memory between here and barrier 0 of the next step:
        // the publishing stores and the loader's requests wait for the window behind that barrier (wave 0's scan), so neither a
        // poll nor a re-poll queues behind them in this CU's memory pipe
        if (MBOX && poller && !last) poll_issue((int)(gstep & 1), 0);
        if constexpr (DMA) {
            // step i+1's tile and uniforms have landed (the barriers of step i+1 hand them to the other waves); step i+2 may still fly
            if (loader && !last) {
                if (i + 2 < a.n_steps) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        } else
        if (loader && !last) { if ((i + 1) & 1) ld_park(i + 1, ldB); else ld_park(i + 1, ldA); }   // LDS only (requested two steps ago)
        if (wave == 0) B3STAMP(5);
        B3STAMP(6);
        B3STAMP(7);
        // LDS reuse: `part` is rewritten by the polling waves at the top of step i+1, its reader (wave 0) is past barrier 1 of
        // step i by then... and every polling wave passes barriers 1-3 of step i in between; cdf / coarse are rewritten by wave 0
        // after barrier 0 of step i+1, i.e. after every search of this step; the S'new / uniform tiles of parity i+1 are rewritten
        // by wave 7 behind barrier 3 (their readers ran before barrier 3 of step i-1); tabb and the bins parked in `pos` behind
        // barrier 1 (their readers, the publishing waves, reach barrier 0 after their loads); Msm behind barrier 2; the score rows
        // read by waves 1 and 6 behind barrier 2 of step i+1.
    }
    if (a.n_steps > 0 && wave != 0 && !loader) publish_step(a.n_steps - 1, prev_slot, acc0, acc1);
    if (call_long) {
        // the last sub-batch (and a write-back still in flight): drain, write back, count this workgroup in
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (wave == 6 && sig_pending && lane == 0) __hip_atomic_fetch_add(a.call->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (wave == 6) {
            asm volatile("buffer_wbl2 sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_fetch_add(a.call->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // ---- hand the point scores to the next launch (its set-up reads them back through pb) ----
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const int row = wave + kBRows * j;
        if (row_ok[j]) {
            a.Sp_out[(tile + row) * N + n0] = scw[j * kBRows * kScPitch + lane];
            a.Sp_out[(tile + row) * N + n1] = scw[j * kBRows * kScPitch + lane + 64];
        }
    }
    // (the last step's totals stay in the mailboxes: the next launch's step 0 reads them there; whoever continues without
    // mailboxes -- the next call, a per-chunk launch -- gets them from launch_mailbox_to_part)
    if (a.dbg != nullptr && b == 0 && tid == 0) { a.dbg[21] = a.dbg[20]; a.dbg[20] = wall_clock64(); }   // (stamps: this / the previous launch's end)
    wg_stamp_end(a.wg_stamps);
}

// the DMA-loader form (atomics exchange, 16-row tiles): 128 registers, four of its waves per SIMD -- two workgroups, or one and
// a pooling workgroup (168 registers x 3 waves per SIMD), fit a CU.  Bit-identical to the register loader and 2 % faster alone
// (10.45 against 10.7 ms of chain per 2048-chunk video) -- and measured in situ (round 6, docs/NOTEBOOK.md): sharing its CUs with
// pooling workgroups gives the pooling 48 more seats (12.5 -> 11.6 ms, 0.56 of the HBM peak), the GEMM and UC kernels 8-15 %,
// and costs role S 3 us per step (chain 12.5 -> 14.9 ms, call 13.4 -> 16.3 ms; shorter pooling bursts do not help); padded to
// 77 KB (two of these per CU, no pooling workgroup) the chain is 13.0-13.6 ms; padded to 84 KB (a CU each) it is the shipped
// pipeline to the noise.  Role S stays a CU's only tenant: experiments build only (INFV_CHAIN_DMA=1).

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

// (TW4 = int4 entries per box of the gather table as a compile-time constant, 0 = read it from the arguments: with a run-time trip
//  count hipcc unrolls the gather loop four ways with remainder loops for every one of the 32 (row, box) positions -- 12 000
//  instructions, 16 us of arithmetic per unit; with the constant, 1.)
template <int TW4>
#ifndef INFV_ALPHA_WPE
#define INFV_ALPHA_WPE 2
#endif
__global__ __launch_bounds__(kA2NT) __attribute__((amdgpu_waves_per_eu(INFV_ALPHA_WPE))) void alpha_rows2_kernel(AlphaRows2Args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = a.N, Q = a.Q, H = a.H, rows = a.rows, tabw = a.tabw;
    const int snp = kA2Q + 1;
    // (Round 5 tried occupancy shaping by registers here -- naming v255 / a7 in an asm clobber makes the allocation 264, so that ONE of
    //  these workgroups fits a CU and a pooling workgroup always fits beside it: they then shared a CU 81 % of the time instead of
    //  18-49 %, this kernel's workgroups lived 34 us instead of 20 beside the streaming loads, the alpha stage took three rounds,
    //  and the call went from 13.5 to 14.5 ms.  Three of them on a CU the pooling has just left is the better deal.)
    wg_stamp_begin(a.wg_stamps);
#ifdef INFV_EXPERIMENTS
    if (a.prio == 3) __builtin_amdgcn_s_setprio(3);          // (experiment INFV_ALPHA_PRIO)
    else if (a.prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (a.prio == 1) __builtin_amdgcn_s_setprio(1);
#endif
    int4* tabb = reinterpret_cast<int4*>(lds);                             // [tabw / 4][4][N / 4]: box n's j-th int4 at (4 j + (n & 3)) * (N / 4) + (n >> 2)
    const int nq = N / 4;                                                  // lanes in use
    float* prev = lds + ((N * tabw + 3) & ~3);                             // [kA2Q][kScPitch]
    float* snew = prev + kA2Q * kScPitch;                                  // [rows][kA2Q + 1]
    // Work unit = (step i, layer-head lh, pass of kA2Q query rows); a workgroup walks units blockIdx.x, + gridDim.x, ... and holds
    // the NEXT unit's inputs in registers while it computes the present one.  Lane = boxes 4 lane .. 4 lane + 3, so that a row's
    // weights leave as ONE 16-byte store per lane, the S'new tile arrives as 16-byte loads and the bias terms as one load per wave:
    // beside a streaming pooling workgroup every vector-memory instruction costs its wave ~0.2 us at issue (round 3, GEMM stamps),
    // and this kernel issued 55 of them per wave and unit (18.5 us per unit for ~3 us of arithmetic); now 18.
    const int n_pass = (Q + kA2Q - 1) / kA2Q, LH = a.L * H;
    const long n_units = (long)a.n_steps * LH * n_pass;
    const int tw4 = TW4 > 0 ? TW4 : tabw / 4, n_tb = N * tw4;                  // int4s per box / per table
    constexpr int kRpw = kA2Q / (kA2NT / 64);                                  // query rows per wave and pass
    // (r_tb as two named values, each written unconditionally: as a conditionally written array hipcc kept it in scratch -- 48 bytes
    //  of private memory and an s_waitcnt vmcnt(0) right behind the "prefetch" of the next unit's table, rounds 4-5)
    int4 r_tb0 = make_int4(-1, -1, -1, -1), r_tb1 = make_int4(-1, -1, -1, -1); floatx4 r_prev[4]; floatx4 r_sn[2]; float r_cq;
    struct Unit { int i, lh, l, q0, qn; long slot, pslot, row0; };
    auto unit_of = [&](long u) {
        Unit t;
        const int pass = (int)(u % n_pass); const long il = u / n_pass;
        t.lh = (int)(il % LH); t.i = (int)(il / LH); t.l = t.lh / H;
        t.q0 = pass * kA2Q; t.qn = min(kA2Q, Q - t.q0);
        t.slot = (a.slot0 + t.i) % a.ring; t.pslot = (a.slot0 + t.i + a.ring - 1) % a.ring;
        t.row0 = (long)t.lh * Q + t.q0;                                      // first row of the pass in [L][H][Q] order
        return t;
    };
    // (launcher: N % 4 == 0, n_tb <= 512, rows * kA2Q / 4 <= 512, Q % 4 == 0 and 16-byte aligned S'new rows)
    auto load_unit = [&](const Unit& t) {                                   // global -> registers: 9 load instructions per wave
        const int32_t* tb = a.tabb_ring + t.slot * a.tab_slot + (long)t.l * N * tabw;
        {
            // (in range for every thread: an index past the table is clamped to its last entry and never parked)
            const int4* tb4 = reinterpret_cast<const int4*>(tb);
            r_tb0 = tb4[min(tid, n_tb - 1)];
            r_tb1 = tb4[min(tid + kA2NT, n_tb - 1)];
        }
        const float* cp = a.crit_ring + t.pslot * a.crit_slot + t.row0 * kBins;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = tid + j * kA2NT;
            if (e < t.qn * (kBins / 4)) r_prev[j] = reinterpret_cast<const floatx4*>(cp)[e];
        }
        const float* sb = a.Snew + (long)t.i * rows * a.snew_ld + t.row0;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int e = tid + j * kA2NT;                                   // (row e / 8, query rows 4 (e % 8) ..)
            const int r = e / (kA2Q / 4), c4 = e - r * (kA2Q / 4);
            floatx4 v = {0.f, 0.f, 0.f, 0.f};
            if (r < rows && 4 * c4 < t.qn) {
                v = *reinterpret_cast<const floatx4*>(sb + (long)r * a.snew_ld + 4 * c4);
                for (int x = 1; x < a.snew_splitk; ++x) v += *reinterpret_cast<const floatx4*>(sb + (long)r * a.snew_ld + 4 * c4 + x * a.snew_split_stride);
            }
            r_sn[j] = v;
        }
        r_cq = (lane < t.qn) ? a.cq[t.row0 + lane] : 0.f;                   // bias term of pass row `lane`
    };
    auto store_unit = [&](const Unit& t) {                                  // registers -> LDS
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int e = tid + j * kA2NT;
            if (e < n_tb) { const int n = e / tw4, jj = e - n * tw4; tabb[(4 * jj + (n & 3)) * nq + (n >> 2)] = j == 0 ? r_tb0 : r_tb1; }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = tid + j * kA2NT;
            if (e < t.qn * (kBins / 4)) {
                const int r = e / (kBins / 4), c4 = e - r * (kBins / 4);
                *reinterpret_cast<floatx4*>(&prev[r * kScPitch + 4 * c4]) = r_prev[j];
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int e = tid + j * kA2NT;
            const int r = e / (kA2Q / 4), c4 = e - r * (kA2Q / 4);
            if (r < rows) {
                snew[r * snp + 4 * c4] = r_sn[j].x; snew[r * snp + 4 * c4 + 1] = r_sn[j].y;
                snew[r * snp + 4 * c4 + 2] = r_sn[j].z; snew[r * snp + 4 * c4 + 3] = r_sn[j].w;
            }
        }
    };
    auto stage_unit_direct = [&](const Unit& t) {                           // shapes beyond the register stage: global -> LDS in loops
        const int32_t* tb = a.tabb_ring + t.slot * a.tab_slot + (long)t.l * N * tabw;
        for (int e = tid; e < n_tb; e += kA2NT) { const int n = e / tw4, jj = e - n * tw4; tabb[(4 * jj + (n & 3)) * nq + (n >> 2)] = reinterpret_cast<const int4*>(tb)[e]; }
        const float* cp = a.crit_ring + t.pslot * a.crit_slot + t.row0 * kBins;
        for (int e = tid; e < t.qn * (kBins / 4); e += kA2NT) {
            const int r = e / (kBins / 4), c4 = e - r * (kBins / 4);
            *reinterpret_cast<floatx4*>(&prev[r * kScPitch + 4 * c4]) = reinterpret_cast<const floatx4*>(cp)[e];
        }
        const float* sb = a.Snew + (long)t.i * rows * a.snew_ld + t.row0;
        for (int e = tid; e < rows * kA2Q; e += kA2NT) {
            const int r = e / kA2Q, qq = e - r * kA2Q;
            float v = 0.f;
            if (qq < t.qn) {
                v = sb[(long)r * a.snew_ld + qq];
                for (int x = 1; x < a.snew_splitk; ++x) v += sb[(long)r * a.snew_ld + qq + x * a.snew_split_stride];
            }
            snew[r * snp + qq] = v;
        }
        r_cq = (lane < t.qn) ? a.cq[t.row0 + lane] : 0.f;
    };
    const bool regs_ok = a.regs_ok != 0;                                    // (launcher: the register stage fits and the rows are 16-byte aligned)
    if (tid < snp) snew[rows * snp + tid] = 0.f;                          // the zero row (read by boxes without a new row)
    if (tid < kA2Q * (kScPitch - kBins)) prev[(tid / (kScPitch - kBins)) * kScPitch + kBins + tid % (kScPitch - kBins)] = 0.f;   // zero words behind every row's points (never rewritten)
    long u = blockIdx.x;
    if (u >= n_units) { wg_stamp_end(a.wg_stamps); return; }
    Unit cur = unit_of(u);
    if (regs_ok) load_unit(cur);
    // static operator entries of this lane's four boxes
    float val[4] = {0.f, 0.f, 0.f, 0.f}, wn[4] = {0.f, 0.f, 0.f, 0.f};
    int brow[4] = {-1, -1, -1, -1};
    const bool lane_ok = 4 * lane < N;
    if (lane_ok) {
        const floatx4 v4 = reinterpret_cast<const floatx4*>(a.box_val)[lane], w4 = reinterpret_cast<const floatx4*>(a.w)[lane];
        const int4 b4 = reinterpret_cast<const int4*>(a.box_row)[lane];
        val[0] = v4.x; val[1] = v4.y; val[2] = v4.z; val[3] = v4.w;
        wn[0] = w4.x; wn[1] = w4.y; wn[2] = w4.z; wn[3] = w4.w;
        brow[0] = b4.x; brow[1] = b4.y; brow[2] = b4.z; brow[3] = b4.w;
    }
#ifdef INFV_EXPERIMENTS
    const bool dbg_on = TW4 != 0 && a.dbg != nullptr;   // (the generic instantiation keeps the shipped register count: test_host_cpu)
    long long t_wg0 = dbg_on ? wall_clock64() : 0, t_stage = 0, t_comp = 0, t_issue = 0; int n_done = 0;
#endif
    for (; u < n_units; u += gridDim.x) {
        cur = unit_of(u);
#ifdef INFV_EXPERIMENTS
        const long long t0 = dbg_on ? wall_clock64() : 0;
#endif
        __syncthreads();                                                    // the previous unit's rows are done with the LDS tiles
        if (regs_ok) store_unit(cur); else stage_unit_direct(cur);
        const float cq_lane = r_cq;
        __syncthreads();
#ifdef INFV_EXPERIMENTS
        const long long t1 = dbg_on ? wall_clock64() : 0;
#endif
        if (regs_ok && u + gridDim.x < n_units) load_unit(unit_of(u + gridDim.x));     // in flight behind this unit's arithmetic
#ifdef INFV_EXPERIMENTS
        if (dbg_on) { const long long t1b = wall_clock64(); t_issue += t1b - t1; }
#endif
        const bool write_sp = a.Sp_out != nullptr && cur.i == a.n_steps - 1;
        const int qn = cur.qn;
        const long row0 = cur.row0, slot = cur.slot;
        float asum_mine = 0.f;                                              // lane j keeps the weight sum of this wave's j-th row
        // The gather table is the same for all rows of the unit: resolve this lane's 4 x tabw entries ONCE into LDS addresses of
        // the wave's first row (row j of the wave is a constant 4 kScPitch floats further: an immediate offset of the read) and
        // into coefficients that are 0 for an empty slot (fma(0, finite, acc) = acc: the same bits as skipping it); a box without
        // a new row reads the zero row behind the S'new tile.  Per row that leaves the reads, the fma chain and the softmax.
        constexpr int kSl = TW4 > 0 ? 4 * TW4 : 4;                          // resolved slots per box held in registers
        const float* gp[4][kSl]; const float* sp[4];
        if (TW4 > 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
#pragma unroll
                for (int jj = 0; jj < TW4; ++jj) {
                    int4 src = make_int4(-1, -1, -1, -1);
                    if (lane_ok) src = tabb[(4 * jj + k) * nq + lane];
                    const int sx[4] = {src.x, src.y, src.z, src.w};
                    // an empty slot reads the zero word behind the row's 128 points (fma(val, 0, acc) == acc: the same bits as skipping it)
#pragma unroll
                    for (int c = 0; c < 4; ++c) gp[k][4 * jj + c] = prev + wave * kScPitch + (sx[c] >= 0 ? sx[c] : kZeroPoint);
                }
                sp[k] = snew + (brow[k] >= 0 ? brow[k] : rows) * snp + wave;
            }
        }
#pragma unroll
        for (int j = 0; j < kRpw; ++j) {
            const int qq = wave + j * (kA2NT / 64);
            if (qq >= qn) break;
            const float* pr = prev + qq * kScPitch;
            const float cqv = __shfl(cq_lane, qq);
            float sv[4];
            float mx = -INFINITY;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                sv[k] = -INFINITY;
                if (lane_ok) {
                    float acc = 0.f;
                    if (TW4 > 0) {
#pragma unroll
                        for (int c = 0; c < kSl; ++c) acc = fmaf(val[k], gp[k][c][j * (kA2NT / 64) * kScPitch], acc);
                        acc += sp[k][j * (kA2NT / 64)];
                    } else {
#pragma unroll 1
                        for (int jj = 0; jj < tw4; ++jj) {
                            const int4 src = tabb[(4 * jj + k) * nq + lane];
                            const float v0 = pr[max(src.x, 0)], v1 = pr[max(src.y, 0)];
                            const float v2 = pr[max(src.z, 0)], v3 = pr[max(src.w, 0)];
                            if (src.x >= 0) acc = fmaf(val[k], v0, acc);
                            if (src.y >= 0) acc = fmaf(val[k], v1, acc);
                            if (src.z >= 0) acc = fmaf(val[k], v2, acc);
                            if (src.w >= 0) acc = fmaf(val[k], v3, acc);
                        }
                        if (brow[k] >= 0) acc += snew[brow[k] * snp + qq];
                    }
                    if (write_sp) a.Sp_out[(row0 + qq) * N + 4 * lane + k] = acc;   // bias-free scores of the call's last step (diagnostics)
                    sv[k] = acc + cqv;
                }
                mx = fmaxf(mx, sv[k]);
            }
            mx = wave_max(mx);
            floatx4 e = {0.f, 0.f, 0.f, 0.f};
            if (lane_ok) { e.x = wn[0] * __expf(sv[0] - mx); e.y = wn[1] * __expf(sv[1] - mx); e.z = wn[2] * __expf(sv[2] - mx); e.w = wn[3] * __expf(sv[3] - mx); }
            float esum = (e.x + e.y) + (e.z + e.w);
            esum = wave_sum(esum);
            const float inv = 1.0f / (esum + a.w_out * __expf(-mx));
            if (lane_ok) {
                e.x *= inv; e.y *= inv; e.z *= inv; e.w *= inv;
                *reinterpret_cast<floatx4*>(a.alpha_ring + slot * a.alpha_slot + (row0 + qq) * N + 4 * lane) = e;
            }
            if (lane == j) asum_mine = esum * inv;
        }
        if (lane < kRpw && wave + lane * (kA2NT / 64) < qn) a.asum_ring[slot * a.asum_slot + row0 + wave + lane * (kA2NT / 64)] = asum_mine;
#ifdef INFV_EXPERIMENTS
        if (dbg_on) { const long long t2 = wall_clock64(); t_stage += t1 - t0; t_comp += t2 - t1; ++n_done; }
#endif
    }
#ifdef INFV_EXPERIMENTS
    if (dbg_on && tid == 0) {
        atomicAdd(reinterpret_cast<unsigned long long*>(a.dbg + 0), (unsigned long long)n_done);
        atomicAdd(reinterpret_cast<unsigned long long*>(a.dbg + 1), (unsigned long long)t_stage);
        atomicAdd(reinterpret_cast<unsigned long long*>(a.dbg + 2), (unsigned long long)t_comp);
        atomicAdd(reinterpret_cast<unsigned long long*>(a.dbg + 3), (unsigned long long)(wall_clock64() - t_wg0));
        atomicAdd(reinterpret_cast<unsigned long long*>(a.dbg + 4), 1ull);
        atomicAdd(reinterpret_cast<unsigned long long*>(a.dbg + 5), (unsigned long long)t_issue);
    }
#endif
    wg_stamp_end(a.wg_stamps);
}

hipError_t launch_alpha_rows2(const AlphaRows2Args& a_, hipStream_t stream) {
    if (a_.n_steps <= 0) return hipSuccess;
    // (one lane per four boxes: 16-byte stores of the weight rows)
    if (a_.N > 256 || a_.N % 4 || a_.tabw % 4 || (reinterpret_cast<unsigned long>(a_.alpha_ring) & 15) || a_.alpha_slot % 4) return hipErrorInvalidValue;
    AlphaRows2Args a = a_;
    static const int prio = [] { const char* e = exp_env("INFV_ALPHA_PRIO"); return e ? atoi(e) : 0; }();
    a.prio = prio;
    // the register stage of the next unit: the table and the S'new tile fit two 16-byte vectors per thread, rows 16-byte aligned
    static const bool force_direct = [] { const char* e = exp_env("INFV_ALPHA_DIRECT"); return e && atoi(e) != 0; }();   // (tests: the fallback staging)
    a.regs_ok = !force_direct && a.N * (a.tabw / 4) <= 2 * kA2NT && a.rows * (kA2Q / 4) <= 2 * kA2NT && a.Q % 4 == 0 && a.snew_ld % 4 == 0 &&
                a.snew_split_stride % 4 == 0 && (reinterpret_cast<unsigned long>(a.Snew) & 15) == 0;
    const size_t lds = (size_t)(((a.N * a.tabw + 3) & ~3) + kA2Q * kScPitch + (a.rows + 1) * (kA2Q + 1)) * sizeof(float);   // (+ a zero row behind the S'new tile)
    // two work units per workgroup (experiments build: INFV_ALPHA_UPW; 1 / 2 / 4 on one box: 14.1 / 13.8 / 15.1 ms per video): the next unit's loads fly behind the present one's arithmetic
    static const int upw = [] { const char* e = exp_env("INFV_ALPHA_UPW"); const int v = e ? atoi(e) : 2; return v > 0 ? v : 1; }();
    const long n_units = (long)a.n_steps * a.L * a.H * ((a.Q + kA2Q - 1) / kA2Q);
    const unsigned grid = (unsigned)((n_units + upw - 1) / upw);
    a.wg_stamps = exp_stamps_reserve(WG_ALPHA, (long)grid);
    a.dbg = nullptr;
#ifdef INFV_EXPERIMENTS
    {   // INFV_ALPHA_STAMPS=1: phase time sums over all workgroups, printed every 64 launches
        static long long* dbg = [] { long long* p = nullptr; if (exp_env("INFV_ALPHA_STAMPS")) { (void)hipMalloc(&p, 8 * sizeof(long long)); (void)hipMemset(p, 0, 8 * sizeof(long long)); } return p; }();
        a.dbg = dbg;
        static int calls = 0;
        if (dbg && (++calls % 64) == 0) {
            long long hb[8];
            (void)hipStreamSynchronize(stream);
            (void)hipMemcpy(hb, dbg, sizeof(hb), hipMemcpyDeviceToHost);
            (void)hipMemset(dbg, 0, 8 * sizeof(long long));
            if (hb[0] > 0 && hb[4] > 0)
                fprintf(stderr, "[alpha stamps] units %lld workgroups %lld: per unit stage (sync + registers -> LDS, incl. waiting for the loads) %.2f us, issue of the next unit's loads %.2f us, arithmetic + stores %.2f us; per workgroup %.2f us\n",
                        hb[0], hb[4], hb[1] / 100.0 / hb[0], hb[5] / 100.0 / hb[0], (hb[2] - hb[5]) / 100.0 / hb[0], hb[3] / 100.0 / hb[4]);
        }
    }
#endif
    // (experiments: INFV_ALPHA_LDS pads the workgroup's LDS -- who may share a CU with whom is decided by LDS and registers)
    static const size_t lds_pad = [] { const char* e = exp_env("INFV_ALPHA_LDS"); return e ? (size_t)atol(e) : (size_t)0; }();
    size_t lds_launch = lds;
    if (lds_pad > lds_launch && lds_pad <= 160 * 1024) {
        static bool attr = false;
        if (!attr) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(alpha_rows2_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(alpha_rows2_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(alpha_rows2_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            attr = true;
        }
        lds_launch = lds_pad;
    }
    if (a.tabw == 4) INFV_LAUNCH(alpha_rows2_kernel<1>, dim3(grid), dim3(kA2NT), lds_launch, stream, a);
    else if (a.tabw == 8) INFV_LAUNCH(alpha_rows2_kernel<2>, dim3(grid), dim3(kA2NT), lds_launch, stream, a);
    else INFV_LAUNCH(alpha_rows2_kernel<0>, dim3(grid), dim3(kA2NT), lds_launch, stream, a);
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
    INFV_LAUNCH(alpha_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, alpha_ring, alpha_slot,
                       asum_ring, asum_slot, slot0, ring, n_steps, rows_per_step, N, w, w_out);
    return hipGetLastError();
}

size_t chain_batch_lds_bytes(int N, int S, int rows, int tabw) { return (size_t)batch_smem(N, S, rows, tabw).total * sizeof(float); }

bool chain_batch_supported(int N, int S, int rows, int tabw, int n_blocks) {
    return N <= kBMaxN && N % 16 == 0 && S <= kBNT && rows <= kBMaxN && tabw <= 16 && (tabw & 3) == 0 &&
           n_blocks <= 384 && chain_batch_lds_bytes(N, S, rows, tabw) <= 100 * 1024;
}

// Variants of the persistent role S.  The shipped library launches ONE: 16-row tiles, atomics exchange, one launch per sub-batch.
// The experiments build can select 8-row tiles (INFV_CHAIN_RPW=1), the mailbox exchange (INFV_CHAIN_XCD=1: inside one XCD's L2
// with the XCD-aware grid, sc1 mailboxes with INFV_CHAIN_LINEAR=1) and ONE launch per call (INFV_CHAIN_CALL=1) for A/B runs.
// Round 5 built the call-long forms of role S, of the pooling and of the projection GEMM and measured them on one box against
// this form (docs/NOTEBOOK.md, round 5): a resident role S is 12 % faster alone (9.4 against 10.65 ms of chain per video) and
// 10.2 ms without the pooling stream -- but whatever is resident holds its CUs for the whole call, the pooling stream's speed is
// the number of CUs that can host one of its workgroups, and every all-resident combination ended 3-5 % SLOWER end to end than
// launches that give their CUs back (14.0-14.2 against 13.4-13.6 ms); a layer's 24 workgroups on ONE XCD (the mailbox exchange
// in one L2) made it 33 ms: every other launch deals its workgroups round-robin over the XCDs and is paced by the fullest one.
constexpr int kDefRpw = 2;
typedef void (*Chain3Fn)(ChainBatchArgs);
bool chain_batch3_mailboxes() {
    static const bool want = [] { const char* e = exp_env("INFV_CHAIN_XCD"); return e && atoi(e) != 0; }();
    return want;
}
bool chain_call_long() {
    static const bool want = [] { const char* e = exp_env("INFV_CHAIN_CALL"); return e && atoi(e) != 0; }();
    return want;
}
static Chain3Fn chain3_fn(int rpw) {
#ifdef INFV_EXPERIMENTS
    if (chain_batch3_mailboxes()) return rpw == 1 ? chain_batch3_kernel<1, true, false> : chain_batch3_kernel<2, true, false>;
    return rpw == 1 ? chain_batch3_kernel<1, false, false> : chain_batch3_kernel<2, false, false>;
#else
    (void)rpw;
    return chain_batch3_kernel<2, false, false>;
#endif
}

// Round 6 (experiments build, INFV_CHAIN_DMA=1): the DMA-loader form of the default role S (chain_batch3_kernel<2, false, true>, 128 registers).
// It applies to every launch whose new-row scores are one slab (the sub-batches of 16 chunks and more), on the atomics exchange
// with per-sub-batch launches.
static bool chain_dma_wanted() {
    static const bool want = [] { const char* e = exp_env("INFV_CHAIN_DMA"); return e && atoi(e) != 0; }();
    return want;
}
static bool chain_dma_applies(const ChainBatchArgs& a, int rpw) {
    return chain_dma_wanted() && rpw == 2 && !chain_batch3_mailboxes() && a.call == nullptr && a.snew_splitk == 1 && a.exp_flags == 0 &&
           a.snew_ld % 4 == 0 && a.Q % 4 == 0 && (reinterpret_cast<unsigned long>(a.Snew) & 15) == 0 && (reinterpret_cast<unsigned long>(a.uf) & 15) == 0 &&
           a.S % 4 == 0 && a.S <= 512 && 4 * a.op.rows <= 64 * kB2Ld &&
           (long)a.n_steps * a.op.rows * a.snew_ld * 4 < (1l << 31) && (long)a.n_steps * a.L * a.S * 4 < (1l << 31);
}

static hipError_t chain_batch_attr() {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(chain_batch_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        for (int rpw = 1; rpw <= 2 && e == hipSuccess; ++rpw)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(chain3_fn(rpw)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#ifdef INFV_EXPERIMENTS
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(chain_batch3_kernel<2, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#endif
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    return hipSuccess;
}

// The persistent kernel spin-waits on its own workgroups: all of them must fit on the device at once, with this
// kernel's registers and LDS, even if nothing else left room (other kernels may still delay residency; the waits are
// bounded and report through the error word).  One workgroup per CU less than the API's answer: the occupancy
// query reads one high for some SGPR counts (MI355X_MICROARCH.md, residency).
static int chain_batch3_rpw(int rows, int Q);
bool chain_batch3_shape_ok(int draw_mode, int points_ok, int rows, int S, int Q) {
    static const bool want_v3 = [] { const char* e = exp_env("INFV_CHAIN_V1"); return !e || atoi(e) == 0; }();
    return want_v3 && draw_mode == 1 && points_ok && 2 * rows <= 64 * kB2Ld && S <= 512 && S % 4 == 0 && Q % 4 == 0 &&
           chain_batch3_rpw(rows, Q) > 0;
}

// Query rows per wave of chain_batch3_kernel for this shape: 2 (16-row tiles, half the workgroups -- each of which owns a
// whole CU by its registers -- and half the arrivals per exchange) when the loader's register tile holds 16 scores per new
// row; INFV_CHAIN_RPW=1 (experiments build) selects 8-row tiles.
static int chain_batch3_rpw(int rows, int Q) {
    static const int want = [] { const char* e = exp_env("INFV_CHAIN_RPW"); return e ? atoi(e) : kDefRpw; }();
    if (want >= 2 && 4 * rows <= 64 * kB2Ld && Q > kBRows) return 2;
#ifdef INFV_EXPERIMENTS
    return 1;
#else
    return 0;                                          // (8-row tiles are not compiled into the shipped library: v1 kernel instead)
#endif
}

bool chain_batch2_applies(const ChainBatchArgs& a) {
    return chain_batch3_shape_ok(a.draw_mode, a.st.points_ok, a.op.rows, a.S, a.Q) && a.crit_ring != nullptr && a.uf != nullptr &&
           (a.mbox != nullptr || !chain_batch3_mailboxes());
}

// Workgroups of the persistent role-S launch for this shape (the kernel launch_chain_batch will choose).
int chain_batch_blocks(int H, int Q, int L, int draw_mode, int points_ok, int rows, int S) {
    const int rpw = chain_batch3_shape_ok(draw_mode, points_ok, rows, S, Q) ? chain_batch3_rpw(rows, Q) : 1;
    return H * ((Q + kBRows * rpw - 1) / (kBRows * rpw)) * L;
}

static size_t chain_batch3_launch_lds(int N, int S, int rows, int tabw, int rpw, bool dma = false) {
    // padding LDS keeps the workgroup's CU footprint what the stream layout of consolidate() was tuned for
    static const int pad = [] { const char* e = exp_env("INFV_S_LDS"); return e ? atoi(e) : 0; }();
    size_t lds = (size_t)batch3_smem(N, S, rows, tabw, rpw, dma).total * sizeof(float);
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
    if (chain_batch3_shape_ok(draw_mode, points_ok, rows, S, Q)) {
        const int rpw = chain_batch3_rpw(rows, Q);
        const size_t lds = chain_batch3_launch_lds(N, S, rows, tabw, rpw);
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, chain3_fn(rpw), kBNT, lds);
    } else {
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, chain_batch_kernel, kBNT, chain_batch_lds_bytes(N, S, rows, tabw));
    }
    if (e != hipSuccess) return false;
    const int safe = per_cu > 1 ? per_cu - 1 : per_cu;
    return (long)safe * cus >= n_blocks;
}

// flag_wait_kernel (shipped since round 6: the GEMM stream holds on the call-long pooling launch's per-sub-batch completion counts):
// one wave, bounded, latches the handle's error word.
__global__ void flag_wait_kernel(const unsigned int* counter, unsigned int target, int spin_limit, unsigned int* error) {
    if (threadIdx.x != 0) return;
    for (int spins = 0;; ++spins) {
        if (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return;
        __builtin_amdgcn_s_sleep(32);
        if ((spins & 1023) == 1023 && __hip_atomic_load(error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) return;   // the chain has already failed
        if (spins > spin_limit) { __hip_atomic_store(error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); return; }
    }
}
hipError_t launch_flag_wait(const unsigned int* counter, unsigned int target, int spin_limit, unsigned int* error, hipStream_t stream) {
    INFV_LAUNCH(flag_wait_kernel, dim3(1), dim3(64), 0, stream, counter, target, spin_limit, error);
    return hipGetLastError();
}
bool launch_flag_wait_available() { return true; }

#ifdef INFV_EXPERIMENTS
// ---- hand-offs of a call-long role-S launch.  GEMM stream -> role S: flag_set_kernel runs behind a sub-batch's projection GEMM (whose
// end-of-kernel release has written its output back) and raises the count role S's loaders poll.  Role S -> UC stream:
// flag_wait_kernel holds the UC stream until every role-S workgroup has counted the sub-batch in (ChainBatchArgs.done); the
// kernels behind it start with the usual launch-time acquire.  One wave each; the wait is bounded and latches the error word.
__global__ void chain_call_desc_kernel(ChainCallDesc* dst, ChainCallDesc v) { if (threadIdx.x == 0) *dst = v; }
hipError_t launch_chain_call_desc(ChainCallDesc* dst, const ChainCallDesc& v, hipStream_t stream) {
    INFV_LAUNCH(chain_call_desc_kernel, dim3(1), dim3(64), 0, stream, dst, v);
    return hipGetLastError();
}
__global__ void flag_set_kernel(unsigned int* flag, unsigned int value) {
    if (threadIdx.x == 0) __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
hipError_t launch_flag_set(unsigned int* flag, unsigned int value, hipStream_t stream) {
    INFV_LAUNCH(flag_set_kernel, dim3(1), dim3(64), 0, stream, flag, value);
    return hipGetLastError();
}

#else
hipError_t launch_chain_call_desc(ChainCallDesc*, const ChainCallDesc&, hipStream_t) { return hipErrorNotSupported; }
hipError_t launch_flag_set(unsigned int*, unsigned int, hipStream_t) { return hipErrorNotSupported; }
#endif

hipError_t launch_chain_batch(const ChainBatchArgs& a_in, hipStream_t stream) {
    if (hipError_t e = chain_batch_attr()) return e;
    if (a_in.n_steps <= 0) return hipSuccess;
    ChainBatchArgs a = a_in;
    if (!chain_batch_supported(a.N, a.S, a.op.rows, a.op.tabw, a.H * a.QS * a.L)) return hipErrorInvalidValue;
    if (chain_batch2_applies(a)) {
        if ((long)a.op.rows * a.snew_ld * 4 >= (1l << 31)) return hipErrorInvalidValue;   // (32-bit byte offsets inside one step's rows of S'new)
        const int rpw = chain_batch3_rpw(a.op.rows, a.Q);
        a.QS = (a.Q + kBRows * rpw - 1) / (kBRows * rpw);             // tiles of 8 * rpw query rows
        const int G = a.H * a.QS;                                      // workgroups of a layer
        // XCD-aware launch: a layer's workgroups get block ids that are equal mod 8 (one XCD under round-robin placement, where
        // the exchange then stays in that XCD's L2); needs a layer to fit one XCD's CUs and a placement class per layer.  The
        // kernel verifies the placement itself (handshake) and is correct without it.
        static const bool linear = [] { const char* e = exp_env("INFV_CHAIN_LINEAR"); return e && atoi(e) != 0; }();   // experiments: linear grid (the layer spread over all XCDs)
        a.xcd_grid = (chain_batch3_mailboxes() && !linear && G <= 32 && a.L <= 8) ? 1 : 0;
        const int blocks = a.xcd_grid ? 8 * G : G * a.L;
        const bool dma = chain_dma_applies(a, rpw);
        const size_t lds = chain_batch3_launch_lds(a.N, a.S, a.op.rows, a.op.tabw, rpw, dma);
        a.wg_stamps = exp_stamps_reserve(WG_CHAIN, blocks);
#ifdef INFV_EXPERIMENTS
        if (dma) { INFV_LAUNCH((chain_batch3_kernel<2, false, true>), dim3(blocks), dim3(kBNT), lds, stream, a); return hipGetLastError(); }
#endif
        INFV_LAUNCH(chain3_fn(rpw), dim3(blocks), dim3(kBNT), lds, stream, a);
        return hipGetLastError();
    }
    a.wg_stamps = nullptr;
    INFV_LAUNCH(chain_batch_kernel, dim3(a.H * a.QS * a.L), dim3(kBNT), chain_batch_lds_bytes(a.N, a.S, a.op.rows, a.op.tabw),
                       stream, a);
    return hipGetLastError();
}

}  // namespace infv
