// gfx950 kernels of the video Q-former path around the LTM (rows a11/a12, next-row f1):
// the short-term cross-attention over a chunk's frame tokens, re-associated so that the frame tokens are only
// ever multiplied by [H*Q x d] operands (never projected to K/V), plus the small query-token blocks of the
// BERT layer (linear + bias + GELU/residual/LayerNorm, 32-token self-attention).
//
// Reference arithmetic restated (infty-Video-LLaMA/InfVideoLLaMA/models/Qformer.py):
//   scores  = (q W_q)_h (k W_k,h^T + b_k,h)^T / sqrt(dh)      :232,244,278
//           = qt_h k^T + const(row)        with qt_h = (q_h / sqrt(dh)) W_k,h ; the row constant cancels in the softmax
//   ctx_h   = softmax(scores) (k W_v,h^T + b_v,h)              :284,298
//           = (softmax(scores) k) W_v,h^T + b_v,h             (rows of the softmax sum to one)
// so one chunk costs two [H*Q x d x T*P] contractions (4.8 GFLOP each at the headline shape) instead of the
// 2 x 9.7 GFLOP K/V projections plus the attention itself.
#include "vqf_internal.h"
#include <cstdint>
#include "ltm_device.h"

namespace infv {

namespace {
constexpr int kBK = 32;
constexpr int kStride = kBK + 4;
}

// ------------------------------------------------------------------------------------------------------
// fp32 MFMA GEMM, 32x32x2 tiles, 4 waves (2x2), LDS-staged operands with register prefetch of the next k-tile.
// ------------------------------------------------------------------------------------------------------
template <int BM, int BN, bool NN>
__global__ __launch_bounds__(256) void qf_gemm_kernel(QfGemm g) {
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int AR = BM / 32, BR = BN / 32;
    constexpr int kBPitch = BN + 4;                  // NN: B tile kept [k][n] as it lies in memory (b128 stores, b32 fragment reads)
    __shared__ float As[BM * kStride];
    __shared__ float Bs[NN ? kBK * kBPitch : BN * kStride];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int b = blockIdx.z / g.splitk, s = blockIdx.z - b * g.splitk;
    const int kbeg = s * g.k_per_split;
    const int kend = (g.K > 0 && kbeg + g.k_per_split > g.K) ? g.K : kbeg + g.k_per_split;   // the last split may be shorter
    const int ntiles = kend > kbeg ? (kend - kbeg) / kBK : 0;
    const int bo = g.inner > 0 ? b / g.inner : 0, bi = g.inner > 0 ? b - bo * g.inner : b;
    const float* A = g.A + (long)bi * g.strideA + (long)bo * g.strideA2;
    float* C = g.C + (long)bi * g.strideC + (long)bo * g.strideC2 + (long)s * g.split_stride;
    const long boff = (long)bi * g.strideB + (long)bo * g.strideB2;

    const int c4 = tid & 7, row0 = tid >> 3;
    const float* a_src[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        const int m = m0 + row0 + 32 * i;
        a_src[i] = (m < g.M) ? A + (long)m * g.lda + kbeg + c4 * 4 : nullptr;
    }
    // B staging: NT -> same map as A (rows of B are output columns); NN -> thread = (one k row, 4 consecutive output
    // columns), consecutive lanes along the row
    constexpr int KT = 256 / (BN / 4);               // NN: k rows covered per pass
    const int kr = tid / (BN / 4), cn = tid - kr * (BN / 4);
    const float* b_src[BR];
    long b_step;
    if (!NN) {
#pragma unroll
        for (int i = 0; i < BR; ++i) {
            const int o = n0 + row0 + 32 * i;
            if (o < g.N) {
                const int seg = o / g.seg_rows;
                b_src[i] = g.B[seg] + boff + (long)(o - seg * g.seg_rows) * g.ldb + kbeg + c4 * 4;
            } else {
                b_src[i] = nullptr;
            }
        }
        b_step = kBK;
    } else {
#pragma unroll
        for (int i = 0; i < BR; ++i)
            b_src[i] = (n0 + 4 * cn < g.N)
                           ? g.B[0] + boff + (long)(kbeg + kr + KT * i) * g.ldb + n0 + 4 * cn
                           : nullptr;
        b_step = (long)kBK * g.ldb;
    }

    floatx4 a_reg[AR], b_reg[BR];
    auto load_tile = [&](int t) {
#pragma unroll
        for (int i = 0; i < AR; ++i)
            a_reg[i] = a_src[i] ? *reinterpret_cast<const floatx4*>(a_src[i] + (long)t * kBK) : floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < BR; ++i)
            b_reg[i] = b_src[i] ? *reinterpret_cast<const floatx4*>(b_src[i] + t * b_step) : floatx4{0.f, 0.f, 0.f, 0.f};
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < AR; ++i)
            *reinterpret_cast<floatx4*>(&As[(row0 + 32 * i) * kStride + c4 * 4]) = a_reg[i];
        if (!NN) {
#pragma unroll
            for (int i = 0; i < BR; ++i)
                *reinterpret_cast<floatx4*>(&Bs[(row0 + 32 * i) * kStride + c4 * 4]) = b_reg[i];
        } else {
#pragma unroll
            for (int i = 0; i < BR; ++i)
                *reinterpret_cast<floatx4*>(&Bs[(kr + KT * i) * kBPitch + 4 * cn]) = b_reg[i];
        }
    };

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int li = lane & 31, kk = lane >> 5;
    load_tile(0);
    for (int t = 0; t < ntiles; ++t) {
        store_tile();
        __syncthreads();
        if (t + 1 < ntiles) load_tile(t + 1);
        floatx4 af[TM][4];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int v = 0; v < 4; ++v)
                af[i][v] = *reinterpret_cast<const floatx4*>(&As[(wm * (BM / 2) + i * 32 + li) * kStride + 16 * kk + 4 * v]);
        if (!NN) {
            floatx4 bf[TN][4];
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int v = 0; v < 4; ++v)
                    bf[j][v] = *reinterpret_cast<const floatx4*>(&Bs[(wn * (BN / 2) + j * 32 + li) * kStride + 16 * kk + 4 * v]);
#pragma unroll
            for (int st = 0; st < 16; ++st)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][st >> 2][st & 3], bf[j][st >> 2][st & 3],
                                                                         acc[i][j], 0, 0, 0);
        } else {
            // lane (li, kk) of MFMA step st needs B[k = 16 kk + st][column li]: 32 consecutive floats per half wave
            float bs[TN][16];
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int st = 0; st < 16; ++st)
                    bs[j][st] = Bs[(16 * kk + st) * kBPitch + wn * (BN / 2) + j * 32 + li];
#pragma unroll
            for (int st = 0; st < 16; ++st)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][st >> 2][st & 3], bs[j][st], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
                const int o = n0 + wn * (BN / 2) + j * 32 + li;
                if (m < g.M && o < g.N) C[(long)m * g.ldc + o] = acc[i][j][r];
            }
}

static bool qf_big_tiles(int M, int N) { return M >= 128 && N >= 128; }

int qf_pick_splitk(int M, int N, int K, int nbatch) {
    const int bm = qf_big_tiles(M, N) ? 128 : 64;
    const long tiles = (long)((M + bm - 1) / bm) * ((N + bm - 1) / bm) * nbatch;
    int sk = 1;
    while (tiles * sk < 192 && sk < 16 && K % (kBK * sk * 2) == 0) sk *= 2;
    return sk;
}

int qf_pick_splitk_fill(int M, int N, int K, int nbatch, int* k_per_split) {
    const int bm = qf_big_tiles(M, N) ? 128 : 64;
    const long tiles = (long)((M + bm - 1) / bm) * ((N + bm - 1) / bm) * nbatch;
    const int ktiles = K / kBK;
    int best_sk = 1;
    double best = -1.0;
    for (int sk = 1; sk <= 16 && sk <= ktiles; ++sk) {
        const int per = (ktiles + sk - 1) / sk;
        if ((long)per * (sk - 1) >= ktiles) continue;                 // an empty last split
        const long wgs = tiles * sk;
        // rounds of 256 CUs, each as long as its longest workgroup (per k-tiles)
        const double work = (double)tiles * ktiles;
        const double cost = (double)((wgs + 255) / 256) * 256.0 * per;
        const double eff = work / cost - 0.002 * sk;                  // prefer fewer slabs at equal fill
        if (eff > best) { best = eff; best_sk = sk; }
    }
    *k_per_split = ((ktiles + best_sk - 1) / best_sk) * kBK;
    return best_sk;
}

hipError_t launch_qf_gemm(const QfGemm& g, bool nn, hipStream_t stream) {
    if (g.M <= 0 || g.N <= 0 || g.nbatch <= 0) return hipSuccess;
    if (g.k_per_split % kBK != 0 || g.k_per_split <= 0) return hipErrorInvalidValue;
    if (nn && (g.N % 4 != 0)) return hipErrorInvalidValue;
    const bool big = qf_big_tiles(g.M, g.N);
    const int bm = big ? 128 : 64;
    dim3 grid((g.M + bm - 1) / bm, (g.N + bm - 1) / bm, g.nbatch * g.splitk);
    if (big) {
        if (nn) INFV_LAUNCH((qf_gemm_kernel<128, 128, true>), grid, dim3(256), 0, stream, g);
        else    INFV_LAUNCH((qf_gemm_kernel<128, 128, false>), grid, dim3(256), 0, stream, g);
    } else {
        if (nn) INFV_LAUNCH((qf_gemm_kernel<64, 64, true>), grid, dim3(256), 0, stream, g);
        else    INFV_LAUNCH((qf_gemm_kernel<64, 64, false>), grid, dim3(256), 0, stream, g);
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------
// Row epilogue: split-K reduction + bias + GELU + residual + LayerNorm (Qformer.py:322-326, 401-403, 415-418).
// ------------------------------------------------------------------------------------------------------
constexpr int kEpiMaxPerThread = 16;      // width <= 4096

__global__ __launch_bounds__(256) void qf_epilogue_kernel(QfEpilogue e) {
    __shared__ double scratch[4];
    const int m = blockIdx.x, tid = threadIdx.x;
    float x[kEpiMaxPerThread];
    const float* in = e.parts + (long)m * e.ld_in;
    const float* res = e.residual ? e.residual + (long)(m % e.res_rows) * e.ld_res : nullptr;
    // Every global load of the row is issued before anything waits: the slabs (eight at a time for all of the
    // thread's columns), bias, residual, gamma and beta.  Chains of dependent round trips made this kernel 2-3x slower.
    float bias_v[kEpiMaxPerThread], res_v[kEpiMaxPerThread], gam_v[kEpiMaxPerThread], bet_v[kEpiMaxPerThread];
#pragma unroll
    for (int i = 0; i < kEpiMaxPerThread; ++i) {
        x[i] = 0.f; bias_v[i] = 0.f; res_v[i] = 0.f; gam_v[i] = 1.f; bet_v[i] = 0.f;
        const int c = tid + 256 * i;
        if (256 * i < e.width && c < e.width) {
            const int seg = c / e.seg_cols;
            if (e.bias[seg]) bias_v[i] = e.bias[seg][c - seg * e.seg_cols];
            if (res) res_v[i] = res[c];
            if (e.gamma) { gam_v[i] = e.gamma[c]; bet_v[i] = e.beta[c]; }
        }
    }
    for (int s = 0; s < e.nsplit; s += 8) {
        float t[kEpiMaxPerThread][8];
#pragma unroll
        for (int i = 0; i < kEpiMaxPerThread; ++i) {
            const int c = tid + 256 * i;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                t[i][k] = (256 * i < e.width && c < e.width && s + k < e.nsplit) ? in[(long)(s + k) * e.split_stride + c] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < kEpiMaxPerThread; ++i)
#pragma unroll
            for (int k = 0; k < 8; ++k) x[i] += t[i][k];
    }
#pragma unroll
    for (int i = 0; i < kEpiMaxPerThread; ++i) {
        const int c = tid + 256 * i;
        if (c < e.width) {
            float v = x[i] + bias_v[i];
            if (e.act == QF_ACT_GELU) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
            v *= e.scale;
            if (res) v += e.res_scale * res_v[i];
            x[i] = v;
        }
    }
    if (e.gamma) {
        double sum = 0.0;
#pragma unroll
        for (int i = 0; i < kEpiMaxPerThread; ++i) sum += (tid + 256 * i < e.width) ? (double)x[i] : 0.0;
        const float mean = (float)(block_sum<256>(sum, scratch) / e.width);
        double sq = 0.0;
#pragma unroll
        for (int i = 0; i < kEpiMaxPerThread; ++i) {
            const float dlt = x[i] - mean;
            sq += (tid + 256 * i < e.width) ? (double)dlt * dlt : 0.0;
        }
        const float var = (float)(block_sum<256>(sq, scratch) / e.width);
        const float rstd = 1.0f / sqrtf(var + e.eps);
#pragma unroll
        for (int i = 0; i < kEpiMaxPerThread; ++i) x[i] = (x[i] - mean) * rstd * gam_v[i] + bet_v[i];
    }
    float* out = e.out + (long)m * e.ld_out;
#pragma unroll
    for (int i = 0; i < kEpiMaxPerThread; ++i) {
        const int c = tid + 256 * i;
        if (c < e.width) out[c] = x[i];
    }
}

// The same epilogue with 16-byte accesses: a thread owns NV float4 of the row (columns 4 * (tid + 256 v)), so a 768-wide
// row is ONE float4 per thread and slab instead of up to 16 predicated scalar slots (9 us -> 4 us per launch on the
// 32-row calls of the per-chunk path, which issues twelve of them per chunk).  Same slab summation order.
template <int NV>
__global__ __launch_bounds__(256) void qf_epilogue_vec_kernel(QfEpilogue e) {
    __shared__ double scratch[4];
    const int m = blockIdx.x, tid = threadIdx.x;
    const float* in = e.parts + (long)m * e.ld_in;
    const float* res = e.residual ? e.residual + (long)(m % e.res_rows) * e.ld_res : nullptr;
    floatx4 x[NV], bias_v[NV], res_v[NV], gam_v[NV], bet_v[NV];
    bool on[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int c = 4 * (tid + 256 * v);
        on[v] = c < e.width;
        x[v] = floatx4{0.f, 0.f, 0.f, 0.f}; bias_v[v] = x[v]; res_v[v] = x[v]; bet_v[v] = x[v];
        gam_v[v] = floatx4{1.f, 1.f, 1.f, 1.f};
        if (on[v]) {
            const int seg = c / e.seg_cols;
            if (e.bias[seg]) bias_v[v] = *reinterpret_cast<const floatx4*>(e.bias[seg] + (c - seg * e.seg_cols));
            if (res) res_v[v] = *reinterpret_cast<const floatx4*>(res + c);
            if (e.gamma) { gam_v[v] = *reinterpret_cast<const floatx4*>(e.gamma + c); bet_v[v] = *reinterpret_cast<const floatx4*>(e.beta + c); }
        }
    }
    for (int s = 0; s < e.nsplit; s += 8) {             // eight slabs in flight
        floatx4 t[NV][8];
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int k = 0; k < 8; ++k)
                t[v][k] = (on[v] && s + k < e.nsplit) ? *reinterpret_cast<const floatx4*>(in + (long)(s + k) * e.split_stride + 4 * (tid + 256 * v))
                                                      : floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int k = 0; k < 8; ++k) x[v] += t[v][k];
    }
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float y = x[v][j] + bias_v[v][j];
            if (e.act == QF_ACT_GELU) y = 0.5f * y * (1.0f + erff(y * 0.70710678118654752440f));
            y *= e.scale;
            if (res) y += e.res_scale * res_v[v][j];
            x[v][j] = y;
        }
    if (e.gamma) {
        double sum = 0.0;
#pragma unroll
        for (int v = 0; v < NV; ++v)
            if (on[v]) sum += ((double)x[v][0] + (double)x[v][1]) + ((double)x[v][2] + (double)x[v][3]);
        const float mean = (float)(block_sum<256>(sum, scratch) / e.width);
        double sq = 0.0;
#pragma unroll
        for (int v = 0; v < NV; ++v)
            if (on[v]) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { const float dlt = x[v][j] - mean; sq += (double)dlt * dlt; }
            }
        const float var = (float)(block_sum<256>(sq, scratch) / e.width);
        const float rstd = 1.0f / sqrtf(var + e.eps);
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int j = 0; j < 4; ++j) x[v][j] = (x[v][j] - mean) * rstd * gam_v[v][j] + bet_v[v][j];
    }
    float* out = e.out + (long)m * e.ld_out;
#pragma unroll
    for (int v = 0; v < NV; ++v)
        if (on[v]) *reinterpret_cast<floatx4*>(out + 4 * (tid + 256 * v)) = x[v];
}

hipError_t launch_qf_epilogue(const QfEpilogue& e, hipStream_t stream) {
    if (e.M <= 0) return hipSuccess;
    if (e.width > 256 * kEpiMaxPerThread || e.seg_cols <= 0 || (e.width + e.seg_cols - 1) / e.seg_cols > kQfMaxSeg)
        return hipErrorInvalidValue;
    {
        auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
        bool vec = e.width % 4 == 0 && e.seg_cols % 4 == 0 && e.ld_in % 4 == 0 && e.split_stride % 4 == 0 && e.ld_out % 4 == 0 &&
                   al(e.parts) && al(e.out) && (!e.residual || (al(e.residual) && e.ld_res % 4 == 0)) &&
                   (!e.gamma || (al(e.gamma) && al(e.beta)));
        for (int sg = 0; sg < kQfMaxSeg; ++sg) vec = vec && (!e.bias[sg] || al(e.bias[sg]));
        if (vec && e.width <= 1024) {
            INFV_LAUNCH(qf_epilogue_vec_kernel<1>, dim3(e.M), dim3(256), 0, stream, e);
            return hipGetLastError();
        }
        if (vec && e.width <= 4096) {
            INFV_LAUNCH(qf_epilogue_vec_kernel<4>, dim3(e.M), dim3(256), 0, stream, e);
            return hipGetLastError();
        }
    }
    INFV_LAUNCH(qf_epilogue_kernel, dim3(e.M), dim3(256), 0, stream, e);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------
// Self-attention over the (<= 32) query tokens of one chunk, one workgroup per (head, chunk)   (Qformer.py:238-301)
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void qf_self_attention_kernel(const float* __restrict__ qkv, int Q, int H,
                                                                float* __restrict__ ctx) {
    __shared__ float qs[32][65], ks[32][65], vs[32][65], ps[32][33];
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int hidden = H * 64;
    const float* base = qkv + (long)b * Q * 3 * hidden;
    for (int e = tid; e < 32 * 64; e += 256) {
        const int r = e >> 6, c = e & 63;
        const bool ok = r < Q;
        const float* row = base + (long)r * 3 * hidden + h * 64 + c;
        qs[r][c] = ok ? row[0] * 0.125f : 0.f;
        ks[r][c] = ok ? row[hidden] : 0.f;
        vs[r][c] = ok ? row[2 * hidden] : 0.f;
    }
    __syncthreads();
    const int r = tid >> 3, g8 = tid & 7;
    float sc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int kc = g8 * 4 + j;
        float a = 0.f;
        for (int c = 0; c < 64; ++c) a = fmaf(qs[r][c], ks[kc][c], a);
        sc[j] = (kc < Q) ? a : -INFINITY;
    }
    float mx = fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3]));
    for (int off = 1; off < 8; off <<= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { sc[j] = expf(sc[j] - mx); sum += sc[j]; }
    for (int off = 1; off < 8; off <<= 1) sum += __shfl_xor(sum, off);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int j = 0; j < 4; ++j) ps[r][g8 * 4 + j] = sc[j] * inv;
    __syncthreads();
    if (r < Q) {
        float* out = ctx + ((long)b * Q + r) * hidden + h * 64 + g8 * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float a = 0.f;
            for (int j = 0; j < 32; ++j) a = fmaf(ps[r][j], vs[j][g8 * 8 + e], a);
            out[e] = a;
        }
    }
}

hipError_t launch_qf_self_attention(const float* qkv, int nb, int Q, int H, float* ctx, hipStream_t stream) {
    if (Q > 32) return hipErrorInvalidValue;
    INFV_LAUNCH(qf_self_attention_kernel, dim3(H, nb), dim3(256), 0, stream, qkv, Q, H, ctx);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------
// Pre-multiplied cross-attention queries: qt[b][h*Q+q][j] = sum_e xq[b][q][h*64+e]/8 * Wk[h*64+e][j]
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void qf_qtilde_kernel(const float* __restrict__ xq, int Q, int H, int d,
                                                        const float* __restrict__ wk, float* __restrict__ qt) {
    __shared__ __attribute__((aligned(16))) float qs[32 * 64];
    const int QT = (Q + 31) / 32;                                // query rows in tiles of 32 (VideoChat2: Q = 96)
    const int h = blockIdx.x, b = blockIdx.z / QT, q0 = (blockIdx.z - b * QT) * 32, tid = threadIdx.x;
    const int hidden = H * 64;
    const int qn = min(32, Q - q0);
    for (int e = tid; e < 32 * 64; e += 256) {
        const int r = e >> 6, c = e & 63;
        qs[e] = (r < qn) ? xq[((long)b * Q + q0 + r) * hidden + h * 64 + c] * 0.125f : 0.f;
    }
    __syncthreads();
    const int j = blockIdx.y * 256 + tid;
    if (j >= d) return;
    float w[64];
#pragma unroll
    for (int e = 0; e < 64; ++e) w[e] = wk[(long)(h * 64 + e) * d + j];
    float* out = qt + ((long)b * H * Q + (long)h * Q + q0) * d + j;
    // four rows at a time: each row is a chain of 64 dependent FMAs, four independent chains keep the VALU busy
    // (rows past qn are zero in LDS and not stored); same per-row order, so the same bits
    for (int r = 0; r < 32; r += 4) {
        if (r >= qn) break;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int e4 = 0; e4 < 16; ++e4) {
            const floatx4 q0v = *reinterpret_cast<const floatx4*>(&qs[(r + 0) * 64 + e4 * 4]);
            const floatx4 q1v = *reinterpret_cast<const floatx4*>(&qs[(r + 1) * 64 + e4 * 4]);
            const floatx4 q2v = *reinterpret_cast<const floatx4*>(&qs[(r + 2) * 64 + e4 * 4]);
            const floatx4 q3v = *reinterpret_cast<const floatx4*>(&qs[(r + 3) * 64 + e4 * 4]);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float wv = w[e4 * 4 + t];
                a0 = fmaf(q0v[t], wv, a0); a1 = fmaf(q1v[t], wv, a1);
                a2 = fmaf(q2v[t], wv, a2); a3 = fmaf(q3v[t], wv, a3);
            }
        }
        out[(long)r * d] = a0;
        if (r + 1 < qn) out[(long)(r + 1) * d] = a1;
        if (r + 2 < qn) out[(long)(r + 2) * d] = a2;
        if (r + 3 < qn) out[(long)(r + 3) * d] = a3;
    }
}

hipError_t launch_qf_qtilde(const float* xq, int nb, int Q, int H, int d, const float* wk, float* qt,
                            hipStream_t stream) {
    INFV_LAUNCH(qf_qtilde_kernel, dim3(H, (d + 255) / 256, nb * ((Q + 31) / 32)), dim3(256), 0, stream, xq, Q, H, d, wk, qt);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------
// Row softmax in place (Qformer.py:284); rows are L2-resident, so three passes cost little.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void qf_softmax_rows_kernel(float* __restrict__ S, int len, long ld) {
    __shared__ float red[4];
    float* row = S + (long)blockIdx.x * ld;
    const int tid = threadIdx.x;
    float mx = -INFINITY;
    for (int c = tid * 4; c < len; c += 1024) {
        const floatx4 v = *reinterpret_cast<const floatx4*>(row + c);
        mx = fmaxf(mx, fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
    }
    mx = wave_max(mx);
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int c = tid * 4; c < len; c += 1024) {
        floatx4 v = *reinterpret_cast<const floatx4*>(row + c);
        v[0] = expf(v[0] - mx); v[1] = expf(v[1] - mx); v[2] = expf(v[2] - mx); v[3] = expf(v[3] - mx);
        *reinterpret_cast<floatx4*>(row + c) = v;
        sum += (v[0] + v[1]) + (v[2] + v[3]);
    }
    sum = wave_sum(sum);
    if ((tid & 63) == 0) red[tid >> 6] = sum;
    __syncthreads();
    const float inv = 1.0f / ((red[0] + red[1]) + (red[2] + red[3]));
    for (int c = tid * 4; c < len; c += 1024) {
        floatx4 v = *reinterpret_cast<const floatx4*>(row + c);
        v[0] *= inv; v[1] *= inv; v[2] *= inv; v[3] *= inv;
        *reinterpret_cast<floatx4*>(row + c) = v;
    }
}

hipError_t launch_qf_softmax_rows(float* S, long n_rows, int len, long ld, hipStream_t stream) {
    if (len % 4 != 0 || ld % 4 != 0) return hipErrorInvalidValue;
    INFV_LAUNCH(qf_softmax_rows_kernel, dim3((unsigned)n_rows), dim3(256), 0, stream, S, len, ld);
    return hipGetLastError();
}

// out[i] = sum_s parts[s][i]   (n multiple of 4)
__global__ __launch_bounds__(256) void qf_sum_slabs_kernel(const float* __restrict__ parts, int nsplit, long stride,
                                                           long n4, float* __restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    floatx4 a = {0.f, 0.f, 0.f, 0.f};
    int s = 0;
    for (; s + 4 <= nsplit; s += 4) {
        const floatx4 v0 = *reinterpret_cast<const floatx4*>(parts + (long)(s + 0) * stride + 4 * i);
        const floatx4 v1 = *reinterpret_cast<const floatx4*>(parts + (long)(s + 1) * stride + 4 * i);
        const floatx4 v2 = *reinterpret_cast<const floatx4*>(parts + (long)(s + 2) * stride + 4 * i);
        const floatx4 v3 = *reinterpret_cast<const floatx4*>(parts + (long)(s + 3) * stride + 4 * i);
        a += v0; a += v1; a += v2; a += v3;
    }
    for (; s < nsplit; ++s) a += *reinterpret_cast<const floatx4*>(parts + (long)s * stride + 4 * i);
    *reinterpret_cast<floatx4*>(out + 4 * i) = a;
}

// in-place reduction of split-K slabs into slab 0
hipError_t launch_qf_sum_slabs(float* parts, int nsplit, long stride, long n, hipStream_t stream) {
    if (nsplit <= 1) return hipSuccess;
    if (n % 4) return hipErrorInvalidValue;
    INFV_LAUNCH(qf_sum_slabs_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, stream, parts, nsplit,
                       stride, n / 4, parts);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void qf_mean_kernel(const float* __restrict__ in, int nb, long n, float* __restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float a = 0.f;
    for (int b = 0; b < nb; ++b) a += in[(long)b * n + i];      // sequential, like torch.mean over dim 0 of a stack
    out[i] = a / (float)nb;
}

hipError_t launch_qf_mean(const float* in, int nb, long n, float* out, hipStream_t stream) {
    INFV_LAUNCH(qf_mean_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, in, nb, n, out);
    return hipGetLastError();
}

}  // namespace infv
