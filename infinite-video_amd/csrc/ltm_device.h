// Device-side building blocks shared by the per-stage kernels (ltm_kernels.hip) and the fused
// whole-video chain kernel (ltm_chain.hip).  All assume 256-thread workgroups (4 waves of 64).
#pragma once
#include "ltm_internal.h"

namespace infv {

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int kBins = 128;               // histogram bins (129 edges); 127 of them are drawn from
constexpr int kVRows = 128;              // V' rows staged per read-out pass
constexpr int kVStride = 80;             // LDS row pitch of the V' stage: == 16 mod 32 -> conflict-free b32 column reads
constexpr int kDPitch = 132;             // LDS row pitch of the edge densities (129 used)
constexpr int kMPitch = 128;             // LDS row pitch of the per-row bin masses

__device__ inline double block_sum_256(double v, double* scratch) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    return scratch[0] + scratch[1] + scratch[2] + scratch[3];
}

// --------------------------------------------------------------------------------------
// Gibbs / sticky draw of one layer (reference long_term_attention_gibbs.py:202-208).
//   p_raw[j] = sum over (head, q-tile) partial masses (fixed order, f64 accumulate);
//   normalised twice (:203, then torch.distributions.Categorical);
//   then torch.multinomial's CPU algorithm: fp32 SEQUENTIAL running sum, divided by the total,
//   last bucket forced to 1, lower-bound search of each float64 uniform.
// The scan runs in one lane's registers (127 dependent fp32 adds, IEEE, same order as the CPU),
// so the draw is bit-exact given identical probabilities.
// Called by all 256 threads; on return sidx[0..S) (LDS) holds the resampled box of every slot.
// --------------------------------------------------------------------------------------
__device__ inline void draw_core(const float* __restrict__ part, int parts,
                                 const float* __restrict__ probs_override, bool use_override,
                                 const StickyView& st, const double* __restrict__ u, int S,
                                 float* cdf /*LDS[kBins]*/, int32_t* sidx /*LDS[S]*/, double* scratch /*LDS[4]*/,
                                 float* total /*LDS[1]*/, float* probs_out, int32_t* bins_out, int32_t* idx_out) {
    constexpr int nb = kBins - 1;
    const int j = threadIdx.x;
    float prob = 0.f;
    if (use_override) {
        if (j < nb) prob = probs_override[j];
    } else {
        double acc = 0.0;
        if (j < nb) {
#pragma unroll 8
            for (int p = 0; p < parts; ++p) acc += (double)part[(long)p * kBins + j];
        }
        const float raw = (float)acc;
        const float tot1 = (float)block_sum_256((double)raw, scratch);
        const float p1 = raw / tot1;
        const float tot2 = (float)block_sum_256((j < nb) ? (double)p1 : 0.0, scratch);
        prob = p1 / tot2;
    }
    if (j < kBins) cdf[j] = (j < nb) ? prob : 0.f;
    if (probs_out != nullptr && j < nb) probs_out[j] = prob;
    __syncthreads();
    if (j == 0) {
        float v[kBins];
#pragma unroll
        for (int i = 0; i < kBins; i += 4) {
            const floatx4 t = *reinterpret_cast<const floatx4*>(&cdf[i]);
            v[i] = t.x; v[i + 1] = t.y; v[i + 2] = t.z; v[i + 3] = t.w;
        }
        float run = 0.f;
#pragma unroll
        for (int i = 0; i < nb; ++i) { run = run + v[i]; v[i] = run; }
#pragma unroll
        for (int i = 0; i < kBins; i += 4)
            *reinterpret_cast<floatx4*>(&cdf[i]) = floatx4{v[i], v[i + 1], v[i + 2], v[i + 3]};
        *total = run;
    }
    __syncthreads();
    if (j < nb) cdf[j] = (j == nb - 1) ? 1.f : cdf[j] / *total;
    __syncthreads();
    for (int s = j; s < S; s += 256) {
        const double us = u[s];
        int lo = 0, hi = nb;
        while (hi - lo > 0) {
            const int mid = lo + (hi - lo) / 2;
            if ((double)cdf[mid] < us) lo = mid + 1; else hi = mid;
        }
        const int box = st.bin_box[lo];
        sidx[s] = box;
        if (bins_out != nullptr) { bins_out[s] = lo; idx_out[s] = box; }
    }
    __syncthreads();
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
