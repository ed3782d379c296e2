// Split-bf16 ("bf16x3") contractions for gfx950: an fp32 operand x is carried as hi = bf16(x), lo = bf16(x - hi)
// (x = hi + lo up to 2^-17 relative), and a product of two such operands as hi*hi + hi*lo + lo*hi on the bf16 MFMA
// pipe with fp32 accumulation (the dropped lo*lo term is below 2^-16 relative).  One bf16 32x32x16 MFMA does 16x
// the work of the fp32 32x32x2 one per cycle, so three of them are ~5x cheaper than an exact fp32 MFMA contraction.
// Used where the result only feeds the 1e-3 read-out budget (short-term attention of the video Q-former, V' half of
// the LTM projection) -- never for the LTM scores, whose rounding feeds the bit-exact Gibbs draw.
#include "vqf_internal.h"
#include "ltm_device.h"

namespace infv {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {
constexpr int kSBK = 64;                          // k per tile (four 32x32x16 steps)
constexpr int kSPitch = 2 * kSBK + 16;            // bytes per LDS row: 64 bf16 + 16 B (b128 reads spread over all banks)
constexpr int kSArr = 128 * kSPitch;              // bytes of one operand array tile (128 rows)
constexpr int kSLds = 4 * kSArr;                  // A_hi, A_lo, B_hi, B_lo tiles (73.7 KB: dynamic LDS)
constexpr int kSVec = kSBK / 16;                  // 16-byte vectors per thread per array tile (a thread owns half a row)

__device__ inline void split2(float x, __bf16& hi, __bf16& lo) {
    hi = (__bf16)x;
    lo = (__bf16)(x - (float)hi);
}
__device__ inline unsigned pack2(__bf16 a, __bf16 b) {
    return (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
}
}  // namespace

// ------------------------------------------------------------------------------------------------------
// C[z][m][o] = sum_k (A_hi + A_lo)[b][m][k] * (B_hi + B_lo)[b][o][k]   (three bf16 MFMA products, fp32 accumulate)
// 128 x 128 x 64 tiles, 4 waves as 2 x 2, register prefetch of the next k-tile.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void split_gemm_kernel(SplitGemm g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * 128, n0 = blockIdx.y * 128;
    const int b = blockIdx.z / g.splitk, s = blockIdx.z - b * g.splitk;
    const int kbeg = s * g.k_per_split;
    const int kend = (kbeg + g.k_per_split > g.K) ? g.K : kbeg + g.k_per_split;
    const int ntiles = kend > kbeg ? (kend - kbeg) / kSBK : 0;
    float* C = g.C + (long)b * g.strideC + (long)s * g.split_stride;

    // staging: thread -> (row = tid >> 1, one half of the row's k-tile segment)
    const int row = tid >> 1, half = tid & 1;
    const bool a_ok = m0 + row < g.M, b_ok = n0 + row < g.N;
    const long a_off = (long)b * g.strideA + (long)(m0 + row) * g.lda + kbeg + half * (kSBK / 2);
    const long b_off = (long)b * g.strideB + (long)(n0 + row) * g.ldb + kbeg + half * (kSBK / 2);
    const uint4* src[4] = {reinterpret_cast<const uint4*>(g.A_hi + a_off), reinterpret_cast<const uint4*>(g.A_lo + a_off),
                           reinterpret_cast<const uint4*>(g.B_hi + b_off), reinterpret_cast<const uint4*>(g.B_lo + b_off)};
    uint4 reg[4][kSVec];
    auto load_tile = [&](int t) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const bool ok = a < 2 ? a_ok : b_ok;
#pragma unroll
            for (int v = 0; v < kSVec; ++v)
                reg[a][v] = ok ? src[a][t * (kSBK / 8) + v] : make_uint4(0u, 0u, 0u, 0u);      // 8 bf16 per uint4
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int v = 0; v < kSVec; ++v)
                *reinterpret_cast<uint4*>(smem + a * kSArr + row * kSPitch + half * kSBK + v * 16) = reg[a][v];
    };

    floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int li = lane & 31, kh = lane >> 5;
    if (ntiles > 0) load_tile(0);
    for (int t = 0; t < ntiles; ++t) {
        store_tile();
        __syncthreads();
        if (t + 1 < ntiles) load_tile(t + 1);
#pragma unroll
        for (int ks = 0; ks < kSBK / 16; ++ks) {
            bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int off = (wm * 64 + i * 32 + li) * kSPitch + ks * 32 + kh * 16;
                ah[i] = *reinterpret_cast<const bf16x8*>(smem + 0 * kSArr + off);
                al[i] = *reinterpret_cast<const bf16x8*>(smem + 1 * kSArr + off);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int off = (wn * 64 + j * 32 + li) * kSPitch + ks * 32 + kh * 16;
                bh[j] = *reinterpret_cast<const bf16x8*>(smem + 2 * kSArr + off);
                bl[j] = *reinterpret_cast<const bf16x8*>(smem + 3 * kSArr + off);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                const int o = n0 + wn * 64 + j * 32 + li;
                if (m < g.M && o < g.N) C[(long)m * g.ldc + o] = acc[i][j][r];
            }
}

hipError_t launch_split_gemm(const SplitGemm& g, hipStream_t stream, int lds_pad) {
    if (g.M <= 0 || g.N <= 0 || g.nbatch <= 0) return hipSuccess;
    {                                                   // tiles live in dynamic LDS; `lds_pad` more (unused) bytes cap the kernel at one workgroup per CU
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(split_gemm_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            attr_set = true;
        }
    }
    if (lds_pad > 160 * 1024 - kSLds) lds_pad = 160 * 1024 - kSLds;
    if (g.K % kSBK || g.k_per_split % kSBK || g.k_per_split <= 0 || g.lda % 8 || g.ldb % 8 || g.strideA % 8 || g.strideB % 8)
        return hipErrorInvalidValue;
    dim3 grid((g.M + 127) / 128, (g.N + 127) / 128, g.nbatch * g.splitk);
    hipLaunchKernelGGL(split_gemm_kernel, grid, dim3(256), kSLds + (lds_pad > 0 ? lds_pad : 0), stream, g);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------
// x [rows][cols] fp32 (leading dimension ld_in) -> hi, lo [rows][cols] bf16 (leading dimension ld_out); cols % 4 == 0
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ x, long ld_in, long rows, int cols4,
                                                         __bf16* __restrict__ hi, __bf16* __restrict__ lo, long ld_out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * cols4) return;
    const long r = i / cols4;
    const int c = (int)(i - r * cols4) * 4;
    const floatx4 v = *reinterpret_cast<const floatx4*>(x + r * ld_in + c);
    __bf16 h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) split2(v[e], h[e], l[e]);
    *reinterpret_cast<uint2*>(hi + r * ld_out + c) = make_uint2(pack2(h[0], h[1]), pack2(h[2], h[3]));
    *reinterpret_cast<uint2*>(lo + r * ld_out + c) = make_uint2(pack2(l[0], l[1]), pack2(l[2], l[3]));
}

hipError_t launch_split_rows(const float* x, long ld_in, long rows, int cols, void* hi, void* lo, long ld_out,
                             hipStream_t stream) {
    if (rows <= 0) return hipSuccess;
    if (cols % 4 || ld_in % 4 || ld_out % 4) return hipErrorInvalidValue;
    const long n = rows * (cols / 4);
    hipLaunchKernelGGL(split_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, ld_in, rows, cols / 4,
                       static_cast<__bf16*>(hi), static_cast<__bf16*>(lo), ld_out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------
// Frame tokens of a chunk, both ways round:  F [n][d] fp32 ->  F_hi/F_lo [n][d]   (B operand of the score contraction)
//                                                             FT_hi/FT_lo [d][n]  (B operand of the token-mean contraction)
// 64 x 64 tiles through LDS; grid (n / 64, d / 64, chunks).
// ------------------------------------------------------------------------------------------------------
// With kbar != nullptr the same pass also writes the frame means kbar[chunk][n / P][d] (the long-term memory's pooled
// frames, Qformer.py:236): P divides 64, so a tile holds 64 / P whole frames; their P tokens are summed in token order and
// divided by P exactly as pool_frames_kernel does, so the two agree bit for bit and the frame tokens are read once.
__global__ __launch_bounds__(256) void split_transpose_kernel(const float* __restrict__ F, int n, int d,
                                                              __bf16* __restrict__ Fh, __bf16* __restrict__ Fl,
                                                              __bf16* __restrict__ Th, __bf16* __restrict__ Tl,
                                                              float* __restrict__ kbar, int P) {
    __shared__ float tile[64][65];
    const int tid = threadIdx.x;
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const long cb = (long)blockIdx.z * n * d;
    const float* src = F + cb;
    // load 64 x 64 fp32 (float4 along d), write the untransposed hi/lo
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int e = tid + 256 * p;                 // 1024 float4 of the tile
        const int r = e >> 4, c4 = (e & 15) * 4;
        floatx4 v = {0.f, 0.f, 0.f, 0.f};
        if (r0 + r < n) v = *reinterpret_cast<const floatx4*>(src + (long)(r0 + r) * d + c0 + c4);
        __bf16 h[4], l[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { split2(v[k], h[k], l[k]); tile[r][c4 + k] = v[k]; }
        if (r0 + r < n) {
            const long o = cb + (long)(r0 + r) * d + c0 + c4;
            *reinterpret_cast<uint2*>(Fh + o) = make_uint2(pack2(h[0], h[1]), pack2(h[2], h[3]));
            *reinterpret_cast<uint2*>(Fl + o) = make_uint2(pack2(l[0], l[1]), pack2(l[2], l[3]));
        }
    }
    __syncthreads();
    // transposed: row = column c of the tile, 4 consecutive tokens per thread
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int e = tid + 256 * p;
        const int c = e >> 4, r4 = (e & 15) * 4;
        if (r0 + r4 < n) {                           // n % 4 == 0 (multiple of 32)
            __bf16 h[4], l[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) split2(tile[r4 + k][c], h[k], l[k]);
            const long o = cb + (long)(c0 + c) * n + r0 + r4;
            *reinterpret_cast<uint2*>(Th + o) = make_uint2(pack2(h[0], h[1]), pack2(h[2], h[3]));
            *reinterpret_cast<uint2*>(Tl + o) = make_uint2(pack2(l[0], l[1]), pack2(l[2], l[3]));
        }
    }
    if (kbar) {
        const int fpt = 64 / P;                      // frames per tile
        for (int e = tid; e < fpt * 64; e += 256) {
            const int f = e >> 6, c = e & 63;
            if (r0 + (f + 1) * P <= n) {
                float acc = 0.f;
                for (int p = 0; p < P; ++p) acc += tile[f * P + p][c];
                kbar[((long)blockIdx.z * (n / P) + r0 / P + f) * d + c0 + c] = acc / (float)P;
            }
        }
    }
}

hipError_t launch_split_transpose(const float* F, int nb, int n, int d, void* Fh, void* Fl, void* Th, void* Tl,
                                  hipStream_t stream, float* kbar, int P) {
    if (n % 32 || d % 64) return hipErrorInvalidValue;
    if (kbar && (P < 1 || 64 % P || n % P)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(split_transpose_kernel, dim3((n + 63) / 64, d / 64, nb), dim3(256), 0, stream, F, n, d,
                       static_cast<__bf16*>(Fh), static_cast<__bf16*>(Fl), static_cast<__bf16*>(Th), static_cast<__bf16*>(Tl),
                       kbar, P);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------
// Row softmax of fp32 scores [n_rows][len] (leading dimension ld), written as split bf16 P_hi / P_lo (ld_out)
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_rows_split_kernel(const float* __restrict__ S, int len, long ld,
                                                                 __bf16* __restrict__ Ph, __bf16* __restrict__ Pl, long ld_out) {
    __shared__ float red[4];
    const float* row = S + (long)blockIdx.x * ld;
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
        const floatx4 v = *reinterpret_cast<const floatx4*>(row + c);
        sum += (expf(v[0] - mx) + expf(v[1] - mx)) + (expf(v[2] - mx) + expf(v[3] - mx));
    }
    sum = wave_sum(sum);
    if ((tid & 63) == 0) red[tid >> 6] = sum;
    __syncthreads();
    const float inv = 1.0f / ((red[0] + red[1]) + (red[2] + red[3]));
    __bf16* ph = Ph + (long)blockIdx.x * ld_out;
    __bf16* pl = Pl + (long)blockIdx.x * ld_out;
    for (int c = tid * 4; c < len; c += 1024) {
        const floatx4 v = *reinterpret_cast<const floatx4*>(row + c);
        __bf16 h[4], l[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) split2(expf(v[k] - mx) * inv, h[k], l[k]);
        *reinterpret_cast<uint2*>(ph + c) = make_uint2(pack2(h[0], h[1]), pack2(h[2], h[3]));
        *reinterpret_cast<uint2*>(pl + c) = make_uint2(pack2(l[0], l[1]), pack2(l[2], l[3]));
    }
}

hipError_t launch_softmax_rows_split(const float* S, long n_rows, int len, long ld, void* Ph, void* Pl, long ld_out,
                                     hipStream_t stream) {
    if (len % 4 || ld % 4 || ld_out % 4) return hipErrorInvalidValue;
    hipLaunchKernelGGL(softmax_rows_split_kernel, dim3((unsigned)n_rows), dim3(256), 0, stream, S, len, ld,
                       static_cast<__bf16*>(Ph), static_cast<__bf16*>(Pl), ld_out);
    return hipGetLastError();
}

}  // namespace infv
