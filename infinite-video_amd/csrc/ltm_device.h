// Device-side building blocks shared by the per-stage kernels (ltm_kernels.hip) and the fused
// whole-video chain kernel (ltm_chain.hip).  All assume 256-thread workgroups (4 waves of 64).
#pragma once
#include "ltm_internal.h"
#include "wg_stamps.h"

namespace infv {

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int kBins = 128;               // histogram bins (129 edges); 127 of them are drawn from
constexpr int kVRows = 128;              // V' rows staged per read-out pass
constexpr int kVStride = 80;             // LDS row pitch of the V' stage: == 16 mod 32 -> conflict-free b32 column reads
constexpr int kDPitch = 132;             // LDS row pitch of the edge densities (129 used)
constexpr int kMPitch = 128;             // LDS row pitch of the per-row bin masses
constexpr double kMassScale = 1099511627776.0;   // 2^40: fixed-point scale of the sticky bin masses (integer atomics)
// The accumulator words also carry, above the mass, the number of workgroups that have added their share
// (persistent role S): one atomic both deposits the mass and announces the arrival, and a reader that sees the
// full count has the complete total in the same round trip.
constexpr int kArriveShift = 54;                                   // masses < 2^14, counts < 2^10
constexpr unsigned long long kMassMask = (1ull << kArriveShift) - 1;
__device__ inline double mass_of(unsigned long long word) { return (double)(word & kMassMask) * (1.0 / kMassScale); }

// Sum over the NT threads of the workgroup (NT/64 waves); every thread gets the result.
template <int NT>
__device__ inline double block_sum(double v, double* scratch /*LDS[NT/64]*/) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) t += scratch[i];
    return t;
}

// --------------------------------------------------------------------------------------
// Gibbs / sticky draw of one layer (reference long_term_attention_gibbs.py:202-208).
//   p_raw[j] = sum over (head, q-tile) partial masses (fixed order, f64 accumulate);
//   normalised twice (:203, then torch.distributions.Categorical);
//   then torch.multinomial's CPU algorithm: fp32 SEQUENTIAL running sum, divided by the total,
//   last bucket forced to 1, lower-bound search of each float64 uniform.
// The scan runs in one lane's registers (127 dependent fp32 adds, IEEE, same order as the CPU),
// so the draw is bit-exact given identical probabilities.
// Called by all NT threads; on return sidx[0..S) (LDS) holds the resampled box of every slot.
// All global reads (partials, uniforms) are issued up front so the draw costs one memory round trip;
// bin_box may point to an LDS copy.  Needs S <= SPT * NT.
// LDS: cdf[kBins] (16-B aligned), sidx[S], gsum[(NT/128)*kBins] doubles, scratch[NT/64] doubles, total[1].
// --------------------------------------------------------------------------------------
// ---- wave64 reductions on the DPP datapath (no LDS crossbar): xor-butterfly inside each row of 16
// lanes (quad_perm, quad_perm, row_half_mirror, row_mirror), then the 4 row sums via v_readlane.
// Every lane gets the result; the summation order is fixed.
template <int CTRL>
__device__ inline float dpp_f32(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ inline float readlane_f32(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ inline float wave_sum(float v) {
    v += dpp_f32<0xB1>(v);            // quad_perm [1,0,3,2]
    v += dpp_f32<0x4E>(v);            // quad_perm [2,3,0,1]
    v += dpp_f32<0x141>(v);           // row_half_mirror
    v += dpp_f32<0x140>(v);           // row_mirror
    return (readlane_f32(v, 0) + readlane_f32(v, 16)) + (readlane_f32(v, 32) + readlane_f32(v, 48));
}
__device__ inline float wave_max(float v) {
    v = fmaxf(v, dpp_f32<0xB1>(v));
    v = fmaxf(v, dpp_f32<0x4E>(v));
    v = fmaxf(v, dpp_f32<0x141>(v));
    v = fmaxf(v, dpp_f32<0x140>(v));
    return fmaxf(fmaxf(readlane_f32(v, 0), readlane_f32(v, 16)), fmaxf(readlane_f32(v, 32), readlane_f32(v, 48)));
}
template <int CTRL>
__device__ inline double dpp_f64(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffLL), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ inline double readlane_f64(double v, int lane) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), lane);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ inline double wave_sum_f64(double v) {
    v += dpp_f64<0xB1>(v);
    v += dpp_f64<0x4E>(v);
    v += dpp_f64<0x141>(v);
    v += dpp_f64<0x140>(v);
    return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}

// Registers of the draw: wave 0 of the workgroup owns the probabilities (lane j: bins j and j+64),
// every thread owns up to SPT uniforms.
template <int SPT>
struct DrawRegs {
    double us[SPT];      // this thread's uniforms (slots tid + i*NT)
    double acc;          // this thread's share (parts p = grp, grp+G, ..) of the mass of bin tid & 127
    float ovr[2];        // wave 0: teacher-forced probabilities of those bins (if overriding)
};

// Phase 1: issue every global read of the draw (uniforms, partial masses) -- no LDS, no barrier.
// `mass_acc` (fixed-point totals accumulated with integer atomics, exact and order-independent) replaces
// the per-workgroup float partials when non-null.
template <int NT, int SPT>
__device__ inline DrawRegs<SPT> draw_load(const float* __restrict__ part, int parts,
                                          const unsigned long long* __restrict__ mass_acc,
                                          const float* __restrict__ probs_override, bool use_override,
                                          const double* __restrict__ u, int S) {
    constexpr int nb = kBins - 1;
    const int tid = threadIdx.x;
    DrawRegs<SPT> r;
#pragma unroll
    for (int i = 0; i < SPT; ++i) r.us[i] = (tid + i * NT < S) ? u[tid + i * NT] : 2.0;
    r.acc = 0.0;
    r.ovr[0] = r.ovr[1] = 0.f;
    if (use_override) {
        if (tid < 64) {
            r.ovr[0] = probs_override[tid];
            if (tid + 64 < nb) r.ovr[1] = probs_override[tid + 64];
        }
    } else if (mass_acc != nullptr) {
        if (tid < kBins - 1) {                                                   // group 0 holds the totals
            unsigned long long tot = 0ull;                                       // integer sum over the replicas (kAccShards)
#pragma unroll
            for (int sh = 0; sh < kAccShards; ++sh) tot += mass_acc[sh * kBins + acc_word(tid)] & kMassMask;
            r.acc = mass_of(tot);
        }
    } else {
        constexpr int G = NT / kBins;                   // thread groups that split the partial rows
        const int j = tid & (kBins - 1), grp = tid / kBins;
#pragma unroll 4
        for (int p = grp; p < parts; p += G) r.acc += (double)part[(long)p * kBins + j];   // bin 127 is padding
    }
    return r;
}

// Phase 2: probabilities -> cdf (wave 0, registers only) -> one barrier -> draws (all threads).
// Ends with a barrier.  cdf[kBins] and sidx[S] live in LDS.
template <int NT, int SPT>
__device__ inline void draw_finish(const DrawRegs<SPT>& r, bool use_override, const int32_t* bin_box, int S,
                                   float* cdf, int32_t* sidx, double* gsum /*LDS[(NT/128)*128]*/, float* probs_out,
                                   int32_t* bins_out, int32_t* idx_out, long long* dbg = nullptr,
                                   const int32_t* __restrict__ forced = nullptr /*[S] bins to resample instead of the draw*/,
                                   float* __restrict__ probs_tr = nullptr, int32_t* __restrict__ bins_tr = nullptr /*draw trace*/) {
#define DSTAMP(i) do { if (dbg != nullptr && threadIdx.x == 0) dbg[i] = wall_clock64(); } while (0)
    constexpr int nb = kBins - 1;
    constexpr int G = NT / kBins;
    const int tid = threadIdx.x;
    if (!use_override) {
        gsum[tid] = r.acc;                               // [grp][bin]
        __syncthreads();
    }
    DSTAMP(12);
    if (tid < 64) {
        const int j0 = tid, j1 = tid + 64;
        float p0, p1;
        if (use_override) {
            p0 = r.ovr[0]; p1 = r.ovr[1];
        } else {
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int g2 = 0; g2 < G; ++g2) { a0 += gsum[g2 * kBins + j0]; a1 += gsum[g2 * kBins + j1]; }
            const float raw0 = (float)a0;
            const float raw1 = (j1 < nb) ? (float)a1 : 0.f;
            const float tot1 = (float)wave_sum_f64((double)raw0 + (double)raw1);
            const float q0 = raw0 / tot1, q1 = raw1 / tot1;                       // LTM.py:203
            const float tot2 = (float)wave_sum_f64((double)q0 + (double)q1);
            p0 = q0 / tot2; p1 = q1 / tot2;                                       // Categorical's own normalisation
        }
        if (j1 >= nb) p1 = 0.f;
        if (probs_out != nullptr) { probs_out[j0] = p0; if (j1 < nb) probs_out[j1] = p1; }
        if (probs_tr != nullptr) { probs_tr[j0] = p0; if (j1 < nb) probs_tr[j1] = p1; }
        DSTAMP(13);
        // Sequential fp32 running sum in bin order (torch.multinomial, CPU), as a systolic scan over the
        // lanes: c <- wave_shr:1(c) + p.  After t steps lane i holds the left-to-right sum of its last
        // min(t,i)+1 terms, so after 63 steps every lane holds ((p_0 + p_1) + ...) + p_i with exactly the
        // CPU's association and rounding -- one dependent VALU instruction per bin, no LDS.
        float c0 = p0;
#pragma unroll
        for (int t = 0; t < 63; ++t) c0 = dpp_f32<0x138>(c0) + p0;          // bins 0..63
        const float carry = readlane_f32(c0, 63);
        const float q1 = (j0 == 0) ? carry + p1 : p1;                       // bin 64 continues the chain
        float c1 = q1;
#pragma unroll
        for (int t = 0; t < 62; ++t) c1 = dpp_f32<0x138>(c1) + q1;          // bins 64..126 (lane 63: padding)
        const float run = readlane_f32(c1, 62);
        cdf[j0] = c0 / run;
        if (j1 < nb) cdf[j1] = (j1 == nb - 1) ? 1.f : c1 / run;
        if (tid == 0) cdf[nb] = 2.f;                                 // pad: never below a uniform
        DSTAMP(14);
    }
    __syncthreads();
    // lower bound of u in the non-decreasing cdf == number of entries below u (torch's binary search
    // returns the same index).  Two levels: 16 coarse entries cdf[8k+7], then the 8 entries of that group.
#pragma unroll
    for (int i = 0; i < SPT; ++i) {
        const int s = tid + i * NT;
        if (s < S) {
            const double us = r.us[i];
            int grp = 0;
#pragma unroll
            for (int k = 0; k < 16; ++k) grp += ((double)cdf[8 * k + 7] < us) ? 1 : 0;
            int lo = 8 * grp;
            if (grp < 16) {
                const floatx4 f0 = *reinterpret_cast<const floatx4*>(&cdf[8 * grp]);
                const floatx4 f1 = *reinterpret_cast<const floatx4*>(&cdf[8 * grp + 4]);
                lo += ((double)f0.x < us) + ((double)f0.y < us) + ((double)f0.z < us) + ((double)f0.w < us) +
                      ((double)f1.x < us) + ((double)f1.y < us) + ((double)f1.z < us);
            }
            lo = min(lo, nb - 1);
            const int box = bin_box[(forced != nullptr) ? min(max(forced[s], 0), nb - 1) : lo];
            sidx[s] = box;
            if (bins_out != nullptr) { bins_out[s] = lo; idx_out[s] = box; }
            if (bins_tr != nullptr) bins_tr[s] = lo;
        }
    }
    __syncthreads();
    DSTAMP(15);
#undef DSTAMP
}

// --------------------------------------------------------------------------------------
// Row-wise phase on a 16-row score tile in LDS (16 threads per row).
//   in : Ssm[row][n] = full scores S (reference :224-230)
//   out: Ssm[row][n] = alpha = w_n e^{S} / (sum_m w_m e^{S_m} + w_out)       (:247-248,269-282)
//        asum[row]   = sum_n alpha
//        part_out[j] = sum over valid rows of the trapezoid mass of histogram interval j+1 of the
//                      row's edge density (cum[j+1]-cum[j], :200-202), j = 0..126
// --------------------------------------------------------------------------------------
__device__ inline void row_phase(float* Ssm, int sstride, int N, int valid_rows, const float* __restrict__ w,
                                 float w_out, const StickyView& st, float* Dsm, float* Msm, float* asum,
                                 float* __restrict__ part_out) {
    const int tid = threadIdx.x;
    const int row = tid >> 4, sub = tid & 15;
    float m = -INFINITY;
    for (int n = sub; n < N; n += 16) m = fmaxf(m, Ssm[row * sstride + n]);
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    const float md = fmaxf(m, 0.f);                    // edges outside every box score 0
    for (int j = sub; j <= kBins; j += 16) {
        const int eb = st.edge_box[j];
        const float sc = (eb >= 0) ? Ssm[row * sstride + eb] : 0.f;
        Dsm[row * kDPitch + j] = expf(sc - md);
    }
    __syncthreads();                                   // raw-score reads done before alpha overwrites
    float esum = 0.f;
    for (int n = sub; n < N; n += 16) {
        const float e = w[n] * expf(Ssm[row * sstride + n] - m);
        Ssm[row * sstride + n] = e;
        esum += e;
    }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) esum += __shfl_xor(esum, off);
    const float inv = 1.0f / (esum + w_out * expf(-m));
    for (int n = sub; n < N; n += 16) Ssm[row * sstride + n] *= inv;
    if (sub == 0) asum[row] = esum * inv;
    float z = 0.f;
    for (int j = sub; j < kBins; j += 16) z += (Dsm[row * kDPitch + j] + Dsm[row * kDPitch + j + 1]) * st.edge_dx[j];
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) z += __shfl_xor(z, off);
    z *= 0.5f;
    for (int j = sub; j < kBins - 1; j += 16) {
        const float dl = Dsm[row * kDPitch + j + 1] / z, dr = Dsm[row * kDPitch + j + 2] / z;
        Msm[row * kMPitch + j] = (row < valid_rows) ? ((dl + dr) * st.edge_dx[j + 1]) * 0.5f : 0.f;
    }
    __syncthreads();
    if (tid < kBins - 1) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < kQTile; ++r) t += Msm[r * kMPitch + tid];
        part_out[tid] = t;
    }
}

// --------------------------------------------------------------------------------------
// Same row-wise phase with ONE WAVE PER QUERY ROW (1024-thread workgroups: wave w owns row w).
// Row reductions are wave reductions; the 16 rows proceed in parallel on the CU's 4 SIMDs.
// --------------------------------------------------------------------------------------
// ALPHA = false: only the sticky bin masses (the critical path of the persistent role S); the scores stay in Ssm and
// the softmax weights are computed later by alpha_rows_kernel, off the chain.
template <bool ALPHA = true>
__device__ inline void row_phase_wave(float* Ssm, int sstride, int N, int valid_rows, const float* w,
                                      float w_out, const int32_t* edge_box, const float* edge_dx, float* Dsm,
                                      float* Msm, float* asum, float* __restrict__ part_out,
                                      unsigned long long* __restrict__ mass_acc, int rows_in_tile,
                                      unsigned long long arrive_inc = 0ull) {
    constexpr int NI = 4;                              // N <= 256: at most 4 boxes per lane
    const int tid = threadIdx.x, lane = tid & 63, row = tid >> 6;
    float* Srow = Ssm + row * sstride;
    float* Drow = Dsm + row * kDPitch;
    float sv[NI];
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int n = lane + 64 * i;
        sv[i] = (n < N) ? Srow[n] : -INFINITY;
        m = fmaxf(m, sv[i]);
    }
    m = wave_max(m);
    const float md = fmaxf(m, 0.f);                    // edges outside every box score 0
    // edge densities feed the Gibbs draw: accurate expf.  129 edges: lanes take j, j+64 and lane 0 edge 128.
    {
        const int eb0 = edge_box[lane], eb1 = edge_box[lane + 64];
        const int eb2 = edge_box[kBins];
        const float s0 = (eb0 >= 0) ? Srow[eb0] : 0.f, s1 = (eb1 >= 0) ? Srow[eb1] : 0.f;
        const float s2 = (eb2 >= 0) ? Srow[eb2] : 0.f;
        Drow[lane] = expf(s0 - md);
        Drow[lane + 64] = expf(s1 - md);
        const float d2 = expf(s2 - md);                // same value in every lane
        if (lane == 0) Drow[kBins] = d2;
    }
    if (ALPHA) {
    __syncthreads();                                   // raw-score reads done before alpha overwrites
    // softmax weights feed only the read-out (1e-3 budget): hardware exp2
    float e[NI];
    float esum = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int n = lane + 64 * i;
        e[i] = (n < N) ? w[n] * __expf(sv[i] - m) : 0.f;
        esum += e[i];
    }
    esum = wave_sum(esum);
    const float inv = 1.0f / (esum + w_out * __expf(-m));
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int n = lane + 64 * i;
        if (n < N) Srow[n] = e[i] * inv;
    }
    if (lane == 0) asum[row] = esum * inv;
    } else {
        // Drow was written by this wave only (row == wave): its own LDS operations are ordered, no barrier needed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const float d0 = Drow[lane], d1 = Drow[lane + 1], d64 = Drow[lane + 64], d65 = Drow[lane + 65];
    const float dx0 = edge_dx[lane], dx1 = edge_dx[lane + 64];
    const float z = wave_sum((d0 + d1) * dx0 + (d64 + d65) * dx1) * 0.5f;
    const float inv_z = 1.0f / z;
    // mass of interval j+1 -> bin j (cum[j+1]-cum[j], LTM.py:201-202): lanes take j = lane and lane+64 (< 127)
    {
        const float d2 = Drow[lane + 2], d66 = (lane + 66 <= kBins) ? Drow[lane + 66] : 0.f;
        const float dxa = edge_dx[lane + 1], dxb = (lane + 65 < kBins) ? edge_dx[lane + 65] : 0.f;
        const bool ok = row < valid_rows;
        Msm[row * kMPitch + lane] = ok ? ((d1 * inv_z + d2 * inv_z) * dxa) * 0.5f : 0.f;
        if (lane + 64 < kBins - 1) Msm[row * kMPitch + lane + 64] = ok ? ((d65 * inv_z + d66 * inv_z) * dxb) * 0.5f : 0.f;
    }
    __syncthreads();
    if (tid < kBins - 1) {
        float t = 0.f;
        for (int r = 0; r < rows_in_tile; ++r) t += Msm[r * kMPitch + tid];
        if (part_out != nullptr) part_out[tid] = t;
        if (mass_acc != nullptr) atomicAdd(&mass_acc[acc_word(tid)], (unsigned long long)((double)t * kMassScale + 0.5) + arrive_inc);
    }
}

// --------------------------------------------------------------------------------------
// One query row by ONE wave, callable for several rows in turn (no workgroup barrier inside): the per-row part of
// row_phase_wave<true>.  In: Srow[n] = scores; out: Srow[n] = alpha, *asum_out = sum alpha, Mrow[j] = this row's
// trapezoid mass of histogram interval j+1 (zero if !valid).  N <= 256.
// --------------------------------------------------------------------------------------
__device__ inline void row_phase_row(float* Srow, int N, bool valid, const float* __restrict__ w, float w_out,
                                     const int32_t* __restrict__ edge_box, const float* __restrict__ edge_dx,
                                     float* Drow, float* Mrow, float* asum_out) {
    constexpr int NI = 4;
    const int lane = threadIdx.x & 63;
    float sv[NI];
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int n = lane + 64 * i;
        sv[i] = (n < N) ? Srow[n] : -INFINITY;
        m = fmaxf(m, sv[i]);
    }
    m = wave_max(m);
    const float md = fmaxf(m, 0.f);
    {
        const int eb0 = edge_box[lane], eb1 = edge_box[lane + 64];
        const int eb2 = edge_box[kBins];
        const float s0 = (eb0 >= 0) ? Srow[eb0] : 0.f, s1 = (eb1 >= 0) ? Srow[eb1] : 0.f;
        const float s2 = (eb2 >= 0) ? Srow[eb2] : 0.f;
        Drow[lane] = expf(s0 - md);
        Drow[lane + 64] = expf(s1 - md);
        const float d2 = expf(s2 - md);
        if (lane == 0) Drow[kBins] = d2;
    }
    // the row is private to this wave: its LDS operations are ordered, only the data must have landed
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float e[NI];
    float esum = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int n = lane + 64 * i;
        e[i] = (n < N) ? w[n] * __expf(sv[i] - m) : 0.f;
        esum += e[i];
    }
    esum = wave_sum(esum);
    const float inv = 1.0f / (esum + w_out * __expf(-m));
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int n = lane + 64 * i;
        if (n < N) Srow[n] = e[i] * inv;
    }
    if (lane == 0) *asum_out = esum * inv;
    const float d0 = Drow[lane], d1 = Drow[lane + 1], d64 = Drow[lane + 64], d65 = Drow[lane + 65];
    const float dx0 = edge_dx[lane], dx1 = edge_dx[lane + 64];
    const float z = wave_sum((d0 + d1) * dx0 + (d64 + d65) * dx1) * 0.5f;
    const float inv_z = 1.0f / z;
    const float d2 = Drow[lane + 2], d66 = (lane + 66 <= kBins) ? Drow[lane + 66] : 0.f;
    const float dxa = edge_dx[lane + 1], dxb = (lane + 65 < kBins) ? edge_dx[lane + 65] : 0.f;
    Mrow[lane] = valid ? ((d1 * inv_z + d2 * inv_z) * dxa) * 0.5f : 0.f;
    if (lane + 64 < kBins - 1) Mrow[lane + 64] = valid ? ((d65 * inv_z + d66 * inv_z) * dxb) * 0.5f : 0.f;
}

// --------------------------------------------------------------------------------------
// Read-out of one (head, 16-row tile): acc[r] = sum_n alpha[4g+r][n] * V'[n][16*wave + c]
//   Asm: alpha tile in LDS (pitch sstride == 2 mod 32); V' rows read from global at
//   Vhead + n * row_pitch (64 floats used), staged through Vsm in passes of kVRows rows.
//   MFMA 16x16x4 f32: A[row=lane&15][k=lane>>4], B[k=lane>>4][col=lane&15].
// --------------------------------------------------------------------------------------
__device__ inline floatx4 readout_tile(const float* Asm, int sstride, int N, const float* __restrict__ Vhead,
                                       long row_pitch, float* Vsm) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    floatx4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int base = 0; base < N; base += kVRows) {
        const int rows = min(kVRows, N - base);
        __syncthreads();                               // previous pass's reads of Vsm are done
        for (int i = tid; i < rows * 16; i += 256) {
            const int r = i >> 4, c4 = i & 15;
            const floatx4 v = *reinterpret_cast<const floatx4*>(Vhead + (long)(base + r) * row_pitch + c4 * 4);
            *reinterpret_cast<floatx4*>(&Vsm[r * kVStride + c4 * 4]) = v;
        }
        __syncthreads();
        for (int t = 0; t < rows / 4; ++t) {
            const float a = Asm[c * sstride + base + 4 * t + g];
            const float b = Vsm[(4 * t + g) * kVStride + 16 * wave + c];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
        }
    }
    return acc;
}

}  // namespace infv
