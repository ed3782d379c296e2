// Dense-operator form of one LTM step (per-call path), for num_basis values whose fp32 boxes overlap.
//
// The rectangular basis of the reference evaluates psi_n(t) = 1[mu_n - w/2 <= t < mu_n + w/2] on fp32 numbers
// (basis_functions.py:248-250).  For most num_basis that are not a power of two (48, 80, 96, 192, ...) a few box
// bounds round so that neighbouring boxes overlap, or leave a gap, exactly at a sample position, a histogram edge or a
// resampling point: psi(t) then has TWO ones (or none).  F F^T is no longer diagonal, the ridge operator
//     G = F^T (F F^T + lambda I)^-1                                   long_term_attention_gibbs.py:68-84
// has two non-zeros in some rows (its inverse comes from LAPACK on the host exactly as in the reference), and the
// one-box-per-point tables of the sparse plan do not apply.  This file is the general form:
//     update   B[n][:] = sum_r G[r][n] x[r][:],   x = [ B_past^T psi(bins[b_s]) ; kbar ]        LTM.py:189,210-216
//              as an fp32 MFMA contraction over the S + T rows of x; a resampled row is the SUM of the (up to two)
//              memory rows whose boxes contain the bin's left edge, or zero when none does
//     masses   density at a histogram edge = exp(sum of the scores of the boxes containing it)      LTM.py:200-202,224-230
// Everything else of the step is shared with the sparse path: the draw (same probabilities -> same bins), the
// projection of the memory, scores, the count-weighted softmax and the read-out (the 1000-point grid of
// expected_value() never lands in two boxes for any supported num_basis; the host checks that).
#include "ltm_device.h"

namespace infv {

// ------------------------------------------------------------------------------------------------------
// B_next[l][n][c] = sum_{r < K} GT[n][r] * x_l[r][c]
//   r <  n_old : x_l[r] = sum_{j<2} B_prev[l][src(l, r)[j]]      (src = box pair of the resampled position, -1 = none)
//   r >= n_old : x_l[r] = kbar[r - n_old]
// One wave per 16 x 16 output tile, v_mfma_f32_16x16x4_f32 over k (an exact fp32 fma chain in row order).
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dense_update_kernel(const float* __restrict__ GT, int K, int ldg, int n_old,
                                                           const int32_t* __restrict__ bins, int bins_stride /*0: none (uniform)*/,
                                                           const int32_t* __restrict__ pos_box2,   /* sticky: [n_bins][2]; uniform: [S][2] */
                                                           const float* __restrict__ B_prev, const float* __restrict__ kbar,
                                                           float* __restrict__ B_next, int N, int d) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l = blockIdx.z;
    const int n0 = blockIdx.y * 16;
    const int c0 = (blockIdx.x * 4 + wave) * 16;
    if (c0 >= d) return;
    const int i = lane & 15, kq = lane >> 4;
    const float* Bp = B_prev + (long)l * N * d;
    const float* grow = GT + (long)(n0 + i) * ldg;
    floatx4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 4) {
        const int r = k0 + kq;
        float a = 0.f, b = 0.f;
        if (r < K) {
            a = grow[r];
            if (r < n_old) {
                const int p = bins_stride ? bins[(long)l * bins_stride + r] : r;
                const int b0 = pos_box2[2 * p], b1 = pos_box2[2 * p + 1];
                if (b0 >= 0) b = Bp[(long)b0 * d + c0 + i];
                if (b1 >= 0) b += Bp[(long)b1 * d + c0 + i];
            } else {
                b = kbar[(long)(r - n_old) * d + c0 + i];
            }
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
    // C/D map of the 16x16 MFMA: col = lane & 15, row = 4 * (lane >> 4) + reg
    float* out = B_next + (long)l * N * d;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) out[(long)(n0 + 4 * kq + rg) * d + c0 + i] = acc[rg];
}

hipError_t launch_dense_update(const float* GT, int K, int ldg, int n_old, const int32_t* bins, int bins_stride,
                               const int32_t* pos_box2, const float* B_prev, const float* kbar, float* B_next, int N, int d,
                               int n_layers, hipStream_t stream) {
    if (N % 16 || d % 16) return hipErrorInvalidValue;
    dim3 grid((d / 16 + 3) / 4, N / 16, n_layers);
    INFV_LAUNCH(dense_update_kernel, grid, dim3(256), 0, stream, GT, K, ldg, n_old, bins, bins_stride, pos_box2, B_prev,
                       kbar, B_next, N, d);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------
// Sticky bin masses from the scores of the step just taken (LTM.py:197-202), with edges that may lie in two boxes:
//   D[j] = exp(sum of scores[h][q][edge_box2[j][.]])   (no box: exp(0));   D /= trapz(D, edges);
//   mass[j] = (D[j+1] + D[j+2]) / 2 * dx[j+1],  j = 0..n_bins-2;   part[l][h][j] = sum_q mass[j]
// One workgroup per (layer, head); wave w takes query rows w, w + 4, ...; lane e takes edges e, e + 64 (and 128).
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dense_masses_kernel(const float* __restrict__ scores, int Q, int N, int H,
                                                           const int32_t* __restrict__ edge_box2, const float* __restrict__ edge_dx,
                                                           float* __restrict__ part) {
    __shared__ float Dsm[4][kBins + 4];
    __shared__ double acc_sm[4][kBins];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = blockIdx.x, l = blockIdx.y;
    const float* S = scores + ((long)l * H + h) * (long)Q * N;
    double m0 = 0.0, m1 = 0.0;
    for (int q = wave; q < Q; q += 4) {
        const float* row = S + (long)q * N;
        auto edge_score = [&](int j) {
            const int b0 = edge_box2[2 * j], b1 = edge_box2[2 * j + 1];
            float s = 0.f;
            if (b0 >= 0) s = row[b0];
            if (b1 >= 0) s += row[b1];
            return s;
        };
        const float s0 = edge_score(lane), s1 = edge_score(lane + 64), s2 = (lane == 0) ? edge_score(kBins) : 0.f;
        float mx = fmaxf(fmaxf(s0, s1), (lane == 0) ? s2 : -INFINITY);
        mx = wave_max(mx);                                   // exp(s - mx): the normalisation below cancels it
        Dsm[wave][lane] = expf(s0 - mx);
        Dsm[wave][lane + 64] = expf(s1 - mx);
        if (lane == 0) Dsm[wave][kBins] = expf(s2 - mx);
        __builtin_amdgcn_s_waitcnt(0xc07f);                   // lgkmcnt(0): this wave's LDS writes (wave-private row)
        __builtin_amdgcn_wave_barrier();
        const float* D = Dsm[wave];
        const float z = wave_sum((D[lane] + D[lane + 1]) * edge_dx[lane] + (D[lane + 64] + D[lane + 65]) * edge_dx[lane + 64]) * 0.5f;
        const float inv_z = 1.0f / z;
        m0 += (double)(((D[lane + 1] * inv_z + D[lane + 2] * inv_z) * edge_dx[lane + 1]) * 0.5f);
        if (lane + 64 < kBins - 1) m1 += (double)(((D[lane + 65] * inv_z + D[lane + 66] * inv_z) * edge_dx[lane + 65]) * 0.5f);
        __builtin_amdgcn_wave_barrier();
    }
    acc_sm[wave][lane] = m0;
    acc_sm[wave][lane + 64] = m1;
    __syncthreads();
    if (threadIdx.x < kBins) {
        const int j = threadIdx.x;
        const double t = (acc_sm[0][j] + acc_sm[1][j]) + (acc_sm[2][j] + acc_sm[3][j]);
        part[((long)l * H + h) * kBins + j] = (j < kBins - 1) ? (float)t : 0.f;
    }
}

hipError_t launch_dense_masses(const float* scores, int Q, int N, int H, int n_layers, const int32_t* edge_box2,
                               const float* edge_dx, float* part, hipStream_t stream) {
    INFV_LAUNCH(dense_masses_kernel, dim3(H, n_layers), dim3(256), 0, stream, scores, Q, N, H, edge_box2, edge_dx, part);
    return hipGetLastError();
}

}  // namespace infv
