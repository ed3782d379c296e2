// General-psi form of one LTM step (per-call path): basis families whose psi(t) is a DENSE row -- the reference's Gaussian
// family (GaussianBasisFunctions, basis_functions.py:135-164; built by add_gaussian_basis_functions,
// long_term_attention_gibbs.py:167-174).  With such a family nothing of the step is a gather any more:
//     resampled rows   x_old[s] = B_past^T psi(ts_s),  ts_s = bins[b_s] (sticky) or s / S (uniform)          LTM.py:207-212
//                      = row pos(s) of  Y = Psi_pos . B_past          (128 distinct positions when sticky: one small GEMM)
//     update           B = (x . G)^T,  x = [x_old ; kbar],  G dense                                          LTM.py:189,215-216
//     edge scores      z(t_j) = (q / sqrt(dh)) . (K^T psi(t_j)) = sum_n S[n] psi_n(t_j)  -> E = S . Psi_edge^T     LTM.py:224-230
//     sticky masses    density exp(z) / trapz(exp(z)), cumulative trapezoid, bin masses                      LTM.py:197-203
//     read-out         prob = exp(z(t_i)) / trapz(exp(z), t) on linspace(0,1,1000);  alpha_n = trapz_i(prob_i psi_n(t_i))
//                      = ((w o prob) . Psi_grid)_n;  ctx = alpha . (V' + bv)                                 LTM.py:251-286
// All contractions run as v_mfma_f32_16x16x4_f32 tiles (one wave per 16 x 16 output tile, operands straight from global
// memory / L2: a correctness path, no reference driver selects this family); the draw, the projection of the memory and
// the scores S themselves are the kernels of the other plans.
#include "ltm_device.h"

namespace infv {

// C[z][m][n] = sum_k A[z][m][k] * (TB ? B[z][n][k] : B[z][k][n]);  any M, Nc, K (tiles and k are masked)
template <bool TB>
__global__ __launch_bounds__(256) void psi_gemm_kernel(const float* __restrict__ A, int lda, long sA, const float* __restrict__ B, int ldb, long sB,
                                                       float* __restrict__ C, int ldc, long sC, int M, int Nc, int K) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m0 = blockIdx.y * 16, n0 = (blockIdx.x * 4 + wave) * 16;
    if (n0 >= Nc) return;
    const int i = lane & 15, kq = lane >> 4;
    const float* a_row = A + blockIdx.z * sA + (long)(m0 + i) * lda;
    const float* Bz = B + blockIdx.z * sB;
    const bool a_ok = m0 + i < M, b_ok = n0 + i < Nc;
    floatx4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 4) {
        const int k = k0 + kq;
        float a = 0.f, b = 0.f;
        if (k < K) {
            if (a_ok) a = a_row[k];
            if (b_ok) b = TB ? Bz[(long)(n0 + i) * ldb + k] : Bz[(long)k * ldb + n0 + i];
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
    float* Cz = C + blockIdx.z * sC;
    if (b_ok) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
            if (m0 + 4 * kq + rg < M) Cz[(long)(m0 + 4 * kq + rg) * ldc + n0 + i] = acc[rg];
    }
}

hipError_t launch_psi_gemm(bool transB, const float* A, int lda, long sA, const float* B, int ldb, long sB, float* C, int ldc, long sC,
                           int M, int Nc, int K, int batch, hipStream_t stream) {
    if (M <= 0 || Nc <= 0 || K <= 0 || batch <= 0) return hipSuccess;
    dim3 grid((Nc + 63) / 64, (M + 15) / 16, batch);
    if (transB) INFV_LAUNCH(psi_gemm_kernel<true>, grid, dim3(256), 0, stream, A, lda, sA, B, ldb, sB, C, ldc, sC, M, Nc, K);
    else INFV_LAUNCH(psi_gemm_kernel<false>, grid, dim3(256), 0, stream, A, lda, sA, B, ldb, sB, C, ldc, sC, M, Nc, K);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------
// B_next[l][n][c] = sum_{r < K} GT[n][r] * x_l[r][c];  r < n_old: x_l[r] = Y[l][pos(l, r)] (pos = bins[l][r] or r), else kbar[r - n_old]
// (dense_update_kernel with precomputed resampled rows)
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void psi_update_kernel(const float* __restrict__ GT, int K, int ldg, int n_old, const int32_t* __restrict__ bins,
                                                         int bins_stride, const float* __restrict__ Y, int n_pos, const float* __restrict__ kbar,
                                                         float* __restrict__ B_next, int N, int d) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l = blockIdx.z;
    const int n0 = blockIdx.y * 16;
    const int c0 = (blockIdx.x * 4 + wave) * 16;
    if (c0 >= d) return;
    const int i = lane & 15, kq = lane >> 4;
    const float* Yl = Y + (long)l * n_pos * d;
    const float* grow = GT + (long)(n0 + i) * ldg;
    floatx4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 4) {
        const int r = k0 + kq;
        float a = 0.f, b = 0.f;
        if (r < K) {
            a = grow[r];
            if (r < n_old) {
                const int p = bins_stride ? bins[(long)l * bins_stride + r] : r;
                b = Yl[(long)p * d + c0 + i];
            } else {
                b = kbar[(long)(r - n_old) * d + c0 + i];
            }
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
    float* out = B_next + (long)l * N * d;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) out[(long)(n0 + 4 * kq + rg) * d + c0 + i] = acc[rg];
}

hipError_t launch_psi_update(const float* GT, int K, int ldg, int n_old, const int32_t* bins, int bins_stride, const float* Y, int n_pos,
                             const float* kbar, float* B_next, int N, int d, int n_layers, hipStream_t stream) {
    if (N % 16 || d % 16) return hipErrorInvalidValue;
    dim3 grid((d / 16 + 3) / 4, N / 16, n_layers);
    INFV_LAUNCH(psi_update_kernel, grid, dim3(256), 0, stream, GT, K, ldg, n_old, bins, bins_stride, Y, n_pos, kbar, B_next, N, d);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------
// Sticky bin masses from the edge scores E[l][h][q][j] (row pitch ldE >= n_bins + 1):  D = exp(E) / trapz(exp(E), edges),
// mass[j] = (D[j+1] + D[j+2]) / 2 * dx[j+1];  part[l][h][j] = sum_q mass[j].   (dense_masses_kernel with given edge scores)
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void psi_masses_kernel(const float* __restrict__ E, int ldE, int Q, int H, const float* __restrict__ edge_dx,
                                                         float* __restrict__ part) {
    __shared__ float Dsm[4][kBins + 4];
    __shared__ double acc_sm[4][kBins];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = blockIdx.x, l = blockIdx.y;
    const float* Eh = E + ((long)l * H + h) * (long)Q * ldE;
    double m0 = 0.0, m1 = 0.0;
    for (int q = wave; q < Q; q += 4) {
        const float* row = Eh + (long)q * ldE;
        const float s0 = row[lane], s1 = row[lane + 64], s2 = row[kBins];
        const float mx = wave_max(fmaxf(fmaxf(s0, s1), s2));      // exp(s - mx): the normalisation below cancels it
        Dsm[wave][lane] = expf(s0 - mx);
        Dsm[wave][lane + 64] = expf(s1 - mx);
        if (lane == 0) Dsm[wave][kBins] = expf(s2 - mx);
        __builtin_amdgcn_s_waitcnt(0xc07f);                        // lgkmcnt(0): this wave's LDS writes (wave-private row)
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

hipError_t launch_psi_masses(const float* E, int ldE, int Q, int H, int n_layers, const float* edge_dx, float* part, hipStream_t stream) {
    INFV_LAUNCH(psi_masses_kernel, dim3(H, n_layers), dim3(256), 0, stream, E, ldE, Q, H, edge_dx, part);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------
// Rows of grid scores z(t_i) -> quadrature-weighted probabilities, in place:  P[i] = w_i exp(z_i) / sum_m w_m exp(z_m)
// (prob = exp(z) / trapz(exp(z), t) and the trapezoid weights of the second integral, LTM.py:247-248,282).  A wave per row.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void psi_grid_kernel(float* __restrict__ Eg, int ldg, int n_grid, long n_rows, const float* __restrict__ w) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_rows) return;
    float* row = Eg + r * ldg;
    float mx = -INFINITY;
    for (int i = lane; i < n_grid; i += 64) mx = fmaxf(mx, row[i]);
    mx = wave_max(mx);
    float z = 0.f;
    for (int i = lane; i < n_grid; i += 64) {
        const float e = w[i] * expf(row[i] - mx);
        row[i] = e;
        z += e;
    }
    z = wave_sum(z);
    const float inv = 1.0f / z;
    for (int i = lane; i < n_grid; i += 64) row[i] *= inv;
}

hipError_t launch_psi_grid(float* Eg, int ldg, int n_grid, long n_rows, const float* w, hipStream_t stream) {
    INFV_LAUNCH(psi_grid_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, stream, Eg, ldg, n_grid, n_rows, w);
    return hipGetLastError();
}

// ctx[l][q][h dh + c] = sum_n alpha[l][h][q][n] (V'[l][n][h dh + c] + bv[l][h dh + c])        (values = proj_value(B), LTM.py:284,313)
__global__ __launch_bounds__(256) void psi_ctx_kernel(const float* __restrict__ alpha, const float* __restrict__ KV, ProjPtrs proj, int Q, int N,
                                                      int H, int dh, float* __restrict__ ctx) {
    const int dm = H * dh;
    const int l = blockIdx.z, q = blockIdx.y, col = blockIdx.x * 256 + threadIdx.x;
    if (col >= dm) return;
    const int h = col / dh;
    const float* al = alpha + (((long)l * H + h) * Q + q) * N;
    const float* V = KV + (long)l * N * 2 * dm + dm + col;                 // V' half of row n: + n * 2 dm
    const float bv = proj.bv[l][col];
    float acc = 0.f;
    for (int n = 0; n < N; ++n) acc = fmaf(al[n], V[(long)n * 2 * dm] + bv, acc);
    ctx[((long)l * Q + q) * dm + col] = acc;
}

hipError_t launch_psi_ctx(const float* alpha, const float* KV, const ProjPtrs& proj, int Q, int N, int H, int dh, int n_layers, float* ctx,
                          hipStream_t stream) {
    INFV_LAUNCH(psi_ctx_kernel, dim3((H * dh + 255) / 256, Q, n_layers), dim3(256), 0, stream, alpha, KV, proj, Q, N, H, dh, ctx);
    return hipGetLastError();
}

}  // namespace infv
