// Internal interface between the video Q-former C ABI (vqf_capi.hip) and its gfx950 kernels (vqf_kernels.hip).
#pragma once
#include "knobs.h"
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace infv {

constexpr int kQfMaxSeg = 4;

// C[z][m][o] (+)= sum_k A[b][m][k] * Bop[b][k][o]   for z = b * splitk + s, k in split s.
// The batch index b = bo * inner + bi addresses operands as  X + bi * strideX + bo * strideX2  (inner = 0: one level).
//   NT: Bop[k][o] = B[o][k]  (B rows are output columns; up to kQfMaxSeg row segments of seg_rows rows each)
//   NN: Bop[k][o] = B[k][o]
struct QfGemm {
    const float* A;  long lda;  long strideA;      // per-batch stride (0 = shared)
    const float* B[kQfMaxSeg];  long ldb;  long strideB;  int seg_rows;
    float* C;  long ldc;  long strideC;  long split_stride;
    int M, N, k_per_split, splitk, nbatch;
    int K;                                           // total depth; 0 = splitk * k_per_split (the last split may be shorter)
    int inner;  long strideA2, strideB2, strideC2;   // optional second batch level (zero-initialised = unused)
};
hipError_t launch_qf_gemm(const QfGemm& g, bool nn, hipStream_t stream);
int qf_pick_splitk(int M, int N, int K, int nbatch);
// split-K that fills whole rounds of the 256 CUs: returns the split count and the (32-aligned) depth per split
int qf_pick_splitk_fill(int M, int N, int K, int nbatch, int* k_per_split);

enum QfAct { QF_ACT_NONE = 0, QF_ACT_GELU = 1 };
// out[m][:] = LN?( scale * act( sum_s parts[s][m][:] + bias ) + res_scale * residual[m % res_rows] )    one workgroup per row
struct QfEpilogue {
    const float* parts;  int nsplit;  long split_stride;  long ld_in;
    const float* bias[kQfMaxSeg];  int seg_cols;               // bias segment s covers columns [s*seg_cols, (s+1)*seg_cols); nullptr = none
    int act;
    float scale, res_scale;                                    // set both to 1 for a plain residual add
    const float* residual;  long ld_res;  int res_rows;        // residual row = m % res_rows (broadcast of a shared [res_rows] block)
    const float* gamma;  const float* beta;  float eps;        // LayerNorm if gamma != nullptr
    float* out;  long ld_out;
    int M, width;
};
hipError_t launch_qf_epilogue(const QfEpilogue& e, hipStream_t stream);

// qkv [nb*Q][3*hidden] (bias applied) -> ctx [nb*Q][hidden]; softmax(q k^T / sqrt(dh)) v per head, Q <= 32, dh = 64
hipError_t launch_qf_self_attention(const float* qkv, int nb, int Q, int H, float* ctx, hipStream_t stream);

// qt[b][h*Q + q][j] = sum_e xq[b][q][h*64 + e] / sqrt(64) * Wk[h*64 + e][j]
hipError_t launch_qf_qtilde(const float* xq, int nb, int Q, int H, int d, const float* wk, float* qt, hipStream_t stream);

// in-place softmax of rows [n_rows][len] (leading dimension ld)
hipError_t launch_qf_softmax_rows(float* S, long n_rows, int len, long ld, hipStream_t stream);

// parts[0][i] = sum_s parts[s][i]  (slabs `stride` floats apart, n a multiple of 4)
hipError_t launch_qf_sum_slabs(float* parts, int nsplit, long stride, long n, hipStream_t stream);

// ---- split-bf16 contractions (split_gemm.hip) ----
// C[z][m][o] = sum_k (A_hi + A_lo)[b][m][k] * (B_hi + B_lo)[b][o][k], z = b * splitk + s; all operands K-contiguous bf16
struct SplitGemm {
    const __bf16* A_hi; const __bf16* A_lo; long lda, strideA;
    const __bf16* B_hi; const __bf16* B_lo; long ldb, strideB;
    float* C; long ldc, strideC, split_stride;
    int M, N, K, k_per_split, splitk, nbatch;
    long long* wg_stamps;            // residency / loop-cycle experiment (wg_stamps.h), or nullptr
};
hipError_t launch_split_gemm(const SplitGemm& g, hipStream_t stream, int lds_pad = 0);
// fp32-accurate contraction from three bf16 planes per operand (six partial products, fp32 accumulation): C = A . B^T
struct SplitGemm6 {
    const __bf16* A[3]; long lda;    // planes of the [M][K] operand, k-tile-major (launch_split3_rows; lda unused)
    const __bf16* B[3]; long ldb;    // planes of the [N][K] operand, k-tile-major
    float* C; long ldc;
    int M, N, K;
    long long* wg_stamps;            // residency experiment (wg_stamps.h), or nullptr
    int narrow;                      // 1: the 128 x 128 kernel even where the 384 x 256 one applies (tests: the two agree bit for bit)
};
hipError_t launch_gemm_x6(const SplitGemm6& g, hipStream_t stream);
// the projection GEMM of a whole consolidate call as one resident launch (gemm_x6_call_kernel): sub-batch b multiplies rows
// [b * sub_rows, ...) of the k-tile-major planes A_all (per sub-batch: split3's layout with rows_total = the sub-batch's rows) by the
// [N][K] planes B into output set b % n_sets
struct GemmCallDesc {
    const __bf16* A_all[3];
    const __bf16* B[3];
    float* C_set[8]; int n_sets;
    long ldc; int N, K;
    int sub_rows;                    // rows of a full sub-batch
    long total_rows;                 // rows of all sub-batches of the launch
    int n_batches;
    int s_col_tile0;                 // first 256-column tile of the S' block (the tiles from there on are a sub-batch's first)
    const unsigned int* pool_done;   // [b] rows of sub-batch b the pooling launch has written
    const unsigned int* uc_done;     // sub-batches whose UC kernel has finished
    unsigned int* tile_ctr;          // next tile of the queue
    unsigned int* done_s;            // [b] S' tiles of sub-batch b complete (role S's loaders poll it)
    unsigned int* done_v;            // [b] V' tiles complete (the UC stream's wait kernel)
    unsigned int* error; int spin_limit;
};
hipError_t launch_gemm_x6_call(const GemmCallDesc& d, GemmCallDesc* d_dev, int n_wgs, hipStream_t stream);   // d_dev: device memory for the descriptor
// x [rows][cols] fp32 = rows row0.. of an operand with rows_total rows -> p0 + p1 + p2 = x, bf16 planes in k-tile-major order
// (element (r, c) at ((c / 16) * rows_total + r) * 16 + c % 16); cols % 16 == 0
hipError_t launch_split3_rows(const float* x, long ld_in, long rows, int cols, void* p0, void* p1, void* p2, long row0, long rows_total,
                              hipStream_t stream);
int shared_worker_stream(hipStream_t* out);   // ltm_capi.hip: the process-wide worker streams
int split_gemm_pick_splitk(int M, int N, int K, int nbatch, int* k_per_split);
long split_gemm_wide_tile_count(int M, int N);   // tiles of the 384 x 256 kernel per batch entry, 0 if the shape has no whole ones   // fills the chip with the tile shape that will run
// x [rows][cols] fp32 -> hi = bf16(x), lo = bf16(x - hi)
hipError_t launch_split_rows(const float* x, long ld_in, long rows, int cols, void* hi, void* lo, long ld_out, hipStream_t stream);
// F [nb][n][d] fp32 -> Fh/Fl [nb][n][d] and Th/Tl [nb][d][n] (bf16 hi/lo)
hipError_t launch_split_transpose(const float* F, int nb, int n, int d, void* Fh, void* Fl, void* Th, void* Tl, hipStream_t stream,
                                  float* kbar = nullptr, int P = 0);   // kbar: also the frame means [nb][n / P][d]
// softmax of fp32 score rows, written as bf16 hi/lo
hipError_t launch_softmax_rows_split(const float* S, long n_rows, int len, long ld, void* Ph, void* Pl, long ld_out, hipStream_t stream);

// out[m][:] = mean over nb of in[b][m][:]
hipError_t launch_qf_mean(const float* in, int nb, long n, float* out, hipStream_t stream);

}  // namespace infv
