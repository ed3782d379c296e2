// gfx950 (MI355X, CDNA4) kernels of the infinity-Video LTM consolidation path.
//
// Reference semantics: infty-Video-LLaMA/InfVideoLLaMA/models/long_term_attention_gibbs.py
// ("LTM.py" below).  With box basis functions the ridge operator G has one non-zero per row,
// so the reference's dense algebra becomes:
//     pool      kbar[t]   = mean_p k[t,p]                                    LTM.py:304
//     rows      R[r]      = val_box(r) * sum_{frames of row r} kbar[f]        LTM.py:189,216-218 (new part of x @ G)
//     project   Pnew[r]   = R[r] . [Wk;Wv]^T                                  LTM.py:312-313 restricted to new rows
//     draw      p -> cdf -> bins -> rows idx                                  LTM.py:200-208
//     update    B'[n]     = val_n * sum_{slots s in box n} B[idx_s] + R[row(n)]      LTM.py:210-216
//               K'[n], V'[n] likewise from their own previous rows + Pnew (projection is linear,
//               so projecting the gathered rows == gathering the projected rows)
//     attend    S = q.(K'+bk)^T/sqrt(dh); alpha = w e^S/(sum w e^S + w_out); ctx = alpha.(V'+bv)
//               and the next step's sticky bin masses                          LTM.py:224-230,247-248,269-284,200-202
//
// Wavefront = 64 lanes everywhere; MFMA shapes are the f32-input ones (exact fp32 fma chains).
#include "ltm_device.h"

#include <cstdlib>
#include <cstring>

namespace infv {

// ======================================================================================
// 1. frame mean-pool:  k [n_frames][P][d] -> kbar [n_frames][d]            (LTM.py:304)
//    One wave per (frame, 256-float column slice): every load instruction of the wave reads
//    1 KiB contiguous; P loads per lane are independent, so the whole 32 KiB of a unit is in
//    flight at once.  This is the only HBM-heavy stage of the path (25.2 MB per chunk).
// ======================================================================================
// Tokens as the producer stores them: fp32 (the reference's layout) or bf16 (half the HBM bytes of the only heavy
// stream of the path; every bf16 value is exact in fp32, the sum runs in fp32 in the same order).
typedef unsigned int uintx4_t __attribute__((ext_vector_type(4)));
struct TokF32 {
    typedef floatx4 vec;                               // 4 columns per lane
    static __device__ inline floatx4 widen(floatx4 v) { return v; }
    // streaming (nt) buffer load: scalar resource + scalar row offset + one lane offset, no 64-bit address per load in flight
    static __device__ inline floatx4 load_nt(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
        const uintx4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 2 /* nt */);
        floatx4 r;
        r.x = __uint_as_float(v.x); r.y = __uint_as_float(v.y); r.z = __uint_as_float(v.z); r.w = __uint_as_float(v.w);
        return r;
    }
};
typedef unsigned int uintx2 __attribute__((ext_vector_type(2)));
struct TokBF16 {
    typedef uintx2 vec;                                // 4 bf16 columns per lane (8 bytes)
    static __device__ inline floatx4 widen(uintx2 v) {
        floatx4 r;
        r.x = __uint_as_float(v.x << 16); r.y = __uint_as_float(v.x & 0xffff0000u);
        r.z = __uint_as_float(v.y << 16); r.w = __uint_as_float(v.y & 0xffff0000u);
        return r;
    }
    static __device__ inline uintx2 load_nt(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
        return __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 2 /* nt */);
    }
};

template <int UNROLL, int NT, class Tok = TokF32>
__global__ __launch_bounds__(NT) void pool_frames_kernel(const void* __restrict__ k_,
                                                          float* __restrict__ kbar,
                                                          long n_units, int P, int d4, int slices) {
    typedef typename Tok::vec tvec;
    const int lane = threadIdx.x & 63;
    // grid-stride over units: with a full grid every wave takes exactly one unit; a smaller grid throttles the kernel
    for (long unit = (long)blockIdx.x * (NT / 64) + (threadIdx.x >> 6); unit < n_units; unit += (long)gridDim.x * (NT / 64)) {
        const long frame = unit / slices;
        const int c4 = (int)(unit - frame * slices) * 64 + lane;
        if (c4 >= d4) continue;
        const tvec* src = reinterpret_cast<const tvec*>(k_) + frame * (long)P * d4 + c4;
        floatx4 acc = {0.f, 0.f, 0.f, 0.f};
        int p = 0;
        for (; p + UNROLL <= P; p += UNROLL) {
            tvec v[UNROLL];
#pragma unroll
            for (int i = 0; i < UNROLL; ++i) v[i] = __builtin_nontemporal_load(src + (long)(p + i) * d4);
#pragma unroll
            for (int i = 0; i < UNROLL; ++i) acc += Tok::widen(v[i]);
        }
        for (; p < P; ++p) acc += Tok::widen(__builtin_nontemporal_load(src + (long)p * d4));
        const float fp = (float)P;
        acc.x /= fp; acc.y /= fp; acc.z /= fp; acc.w /= fp;      // mean = sum / P, as torch does
        // streaming store: leave no dirty lines in L2 (every kernel boundary of the concurrent chain writes L2 back)
        __builtin_nontemporal_store(acc, reinterpret_cast<floatx4*>(kbar) + frame * d4 + c4);
    }
}

// `lds_pad` bytes of (unused) dynamic LDS per workgroup cap how many of them a CU hosts, so that a
// latency-critical kernel on another stream always finds wave slots and LDS (see consolidate()).
template <class Tok>
static hipError_t launch_pool_t(const void* k, float* kbar, int64_t n_frames, int P, int d, hipStream_t stream, int lds_pad);

hipError_t launch_pool(const void* k, int k_bf16, float* kbar, int64_t n_frames, int P, int d, hipStream_t stream, int lds_pad) {
    return k_bf16 ? launch_pool_t<TokBF16>(k, kbar, n_frames, P, d, stream, lds_pad)
                  : launch_pool_t<TokF32>(k, kbar, n_frames, P, d, stream, lds_pad);
}

template <class Tok>
static hipError_t launch_pool_t(const void* k, float* kbar, int64_t n_frames, int P, int d, hipStream_t stream, int lds_pad) {
    const int d4 = d / 4;
    const int slices = (d4 + 63) / 64;
    const long n_units = (long)n_frames * slices;
    if (n_units == 0) return hipSuccess;
    if (lds_pad > 0) {
        // overlapped mode: 512-thread workgroups whose padding LDS lets only ONE of them live on a CU
        // (8 waves x 4 KiB in flight still cover the HBM latency), so the chain kernel always finds room.
        // Loads in flight per wave: the pool's outstanding bytes set the queueing delay that every other kernel's memory
        // operation sees while it runs; 4 KiB per wave (8 MiB chip-wide) keeps ~95 % of the in-situ pooling rate and gives
        // role S back ~0.5 ms per video; 16 is fastest for the pool alone.  (Rounds 1-3 also measured 256- and 1024-thread
        // workgroups, 2 / 8 / 16 KiB in flight, a bounded grid and a rolling double buffer: all within noise or slower in
        // the pipeline -- DESIGN notebook; those instantiations are gone.)
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(pool_frames_kernel<4, 512, Tok>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            attr_set = true;
        }
        const unsigned grid = (unsigned)((n_units + 7) / 8);
        INFV_LAUNCH((pool_frames_kernel<4, 512, Tok>), dim3(grid), dim3(512), lds_pad, stream, k, kbar, n_units, P, d4, slices);
    } else {
        INFV_LAUNCH((pool_frames_kernel<16, 256, Tok>), dim3((unsigned)((n_units + 3) / 4)), dim3(256), 0, stream, k, kbar,
                           n_units, P, d4, slices);
    }
    return hipGetLastError();
}

// ======================================================================================
// 2. new coefficient rows:  R[c][r][:] = val * sum_{f in [begin_r,end_r)} kbar[c][f][:]
// ======================================================================================
__global__ __launch_bounds__(256) void build_rows_kernel(const float* __restrict__ kbar, int T, int d4,
                                                         OperatorView op, float* __restrict__ R) {
    const int r = blockIdx.x, c = blockIdx.y;
    const int b = op.row_begin[r], e = op.row_end[r];
    const float val = op.box_val[op.row_box[r]];
    const floatx4* src = reinterpret_cast<const floatx4*>(kbar) + ((long)c * T) * d4;
    floatx4* dst = reinterpret_cast<floatx4*>(R) + ((long)c * op.rows + r) * d4;
    for (int c4 = threadIdx.x; c4 < d4; c4 += blockDim.x) {
        floatx4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int f = b; f < e; ++f) {
            const floatx4 v = src[(long)f * d4 + c4];
            acc.x = fmaf(val, v.x, acc.x); acc.y = fmaf(val, v.y, acc.y);
            acc.z = fmaf(val, v.z, acc.z); acc.w = fmaf(val, v.w, acc.w);
        }
        __builtin_nontemporal_store(acc, dst + c4);
    }
}

// ------------------------------------------------------------------------------------------------------
// 2c. pool + rows with SHORT-LIVED workgroups: one workgroup per (chunk, row); wave (fi, slice) pools frame fb + fi of the row for
//     one 256-float slice exactly as pool_frames_kernel does (32 loads of 1 KiB, U per burst), parks the frame mean in LDS, and the
//     fi = 0 waves run build_rows_kernel's fma chain over the row's frames in order -- same bits as the two kernels, no kbar round
//     trip through HBM, no rows kernel, and a workgroup lives as long as one of pool_frames_kernel's (a frame's worth of loads per
//     wave), which is what lets role S find empty CUs at its launches (pool_rows_kernel's workgroups live four times longer).
// ------------------------------------------------------------------------------------------------------
// (bf16 three-piece split of an fp32 value, as split_gemm.hip's split3_rows_kernel makes them: x = p0 + p1 + p2 exactly)
__device__ inline void pool_split3(float x, __bf16& p0, __bf16& p1, __bf16& p2) {
    p0 = (__bf16)x;
    const float r1 = x - (float)p0;
    p1 = (__bf16)r1;
    p2 = (__bf16)(r1 - (float)p1);
}
__device__ inline unsigned pool_pack2(__bf16 a, __bf16 b) {
    return (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
}

template <int U, class Tok, bool CALL>
__global__ __launch_bounds__(1024) void pool_rows2_kernel(const void* __restrict__ k_, long chunk_stride, int P, int d4, int slices,
                                                          OperatorView op, long n_rows_total, float* __restrict__ R,
                                                          long long* __restrict__ stamps, int prio, int tid_addr, PoolCallDesc pc) {
    typedef typename Tok::vec tvec;
    extern __shared__ __attribute__((aligned(16))) float pr2_lds[];          // [4 frames][d4] float4
    floatx4* park = reinterpret_cast<floatx4*>(pr2_lds);
    wg_stamp_begin(stamps);
    // (experiment INFV_POOL_PRIO=1: issue priority over co-resident waves.  Beside a projection GEMM workgroup the matrix waves'
    //  back-to-back MFMAs win the VALU port and this kernel's few adds per load wait behind them -- a pooling workgroup then
    //  lives 40 us instead of 14, tools/sweep_r03y.sh; with the priority its workgroups live 14 us everywhere, but the UC and
    //  chain kernels pay for it and the call gets slower, 122 k against 135 k chunks/s, tools/sweep_r03z.sh.  Off.)
    if (prio == 3) __builtin_amdgcn_s_setprio(3);
    else if (prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (prio == 1) __builtin_amdgcn_s_setprio(1);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fi = wave / slices, sl = wave - fi * slices;
    const int c4 = sl * 64 + lane;
    const bool col_ok = c4 < d4;
    const float fp = (float)P;
    // (rows + planes: one row per workgroup.  Round 5 also ran this form as 176 RESIDENT workgroups walking the rows grid-stride, with
    //  role S and the GEMM resident too -- every CU with exactly one resident kernel: the pooling finished its 51.5 GB in 11.2 ms
    //  (4.6 TB/s) and the call took 19-22 ms: beside a workgroup that streams without a pause the UC / alpha workgroups lived up to
    //  four times longer and role S's exchange slowed.  Short-lived workgroups that leave gaps are part of why the pipeline works.)
    // (round 6 tried 2 / 4 / 8 consecutive rows per workgroup in the one-launch form -- set-up, store drain and dispatcher turn-around
    //  paid once per n rows: 14.46 / 15.80 / 16.45 against 13.65 ms per video.  Shorter-lived pooling workgroups win.)
    for (long rr = blockIdx.x; rr < n_rows_total; rr += CALL ? n_rows_total : (long)gridDim.x) {
        const long c = rr / op.rows;
        const int r = (int)(rr - c * op.rows);
        const int fb = op.row_begin[r], fe = op.row_end[r];
        const float val = op.box_val[op.row_box[r]];
        floatx4 racc = {0.f, 0.f, 0.f, 0.f};
        for (int f0 = fb; f0 < fe; f0 += 4) {
            const int f = f0 + fi;
            if (f < fe && col_ok) {
                const tvec* frame = reinterpret_cast<const tvec*>(k_) + c * chunk_stride + (long)f * P * d4;       // wave-uniform
                const int row_bytes = d4 * (int)sizeof(tvec);
                floatx4 acc = {0.f, 0.f, 0.f, 0.f};
                int p = 0;
                if (tid_addr) {
                    // loads with NO vector address operand: the resource has ADD_TID_ENABLE (index = lane id, stride = one lane's
                    // bytes) and starts at this wave's 64-lane column slice, the row offset is scalar -- nothing of the
                    // instruction is read from the vector register file, whose read ports a co-resident GEMM's MFMAs keep busy
                    __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc(const_cast<tvec*>(frame + sl * 64), (short)sizeof(tvec), 64, 1 << 23);
                    for (; p + U <= P; p += U) {
                        tvec v[U];
#pragma unroll
                        for (int i = 0; i < U; ++i) v[i] = Tok::load_nt(rt, 0, (p + i) * row_bytes);
#pragma unroll
                        for (int i = 0; i < U; ++i) acc += Tok::widen(v[i]);
                    }
                    for (; p < P; ++p) acc += Tok::widen(Tok::load_nt(rt, 0, p * row_bytes));
                } else {
                __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<tvec*>(frame), 0, P * d4 * (int)sizeof(tvec), 0x00020000);
                const int voff = c4 * (int)sizeof(tvec);
                for (; p + U <= P; p += U) {
                    tvec v[U];
#pragma unroll
                    for (int i = 0; i < U; ++i) v[i] = Tok::load_nt(rs, voff, (p + i) * row_bytes);
#pragma unroll
                    for (int i = 0; i < U; ++i) acc += Tok::widen(v[i]);
                }
                for (; p < P; ++p) acc += Tok::widen(Tok::load_nt(rs, voff, p * row_bytes));
                }
                acc.x /= fp; acc.y /= fp; acc.z /= fp; acc.w /= fp;              // mean = sum / P, as torch does (LTM.py:304)
                park[fi * d4 + c4] = acc;
            }
            __syncthreads();
            if (fi == 0 && col_ok) {
                const int nf = min(4, fe - f0);
                for (int j = 0; j < nf; ++j) {
                    const floatx4 v = park[j * d4 + c4];
                    racc.x = fmaf(val, v.x, racc.x); racc.y = fmaf(val, v.y, racc.y);
                    racc.z = fmaf(val, v.z, racc.z); racc.w = fmaf(val, v.w, racc.w);
                }
            }
            __syncthreads();
        }
        if constexpr (!CALL) {
            if (fi == 0 && col_ok) __builtin_nontemporal_store(racc, reinterpret_cast<floatx4*>(R) + rr * (long)d4 + c4);
        } else {
            // call-long launch: consumers are kernels launched behind a flag wait while this launch is still running -- every store is
            // write-through (sc1), drained, then the workgroup counts itself into its sub-batch's word
            const long cb = c / pc.sub;
            if (fi == 0 && col_ok) {
                __amdgpu_buffer_rsrc_t rr_ = __builtin_amdgcn_make_buffer_rsrc(pc.R_all + rr * (long)d4 * 4, 0, d4 * 16, 0x00020000);
                uintx4_t w;
                w.x = __float_as_uint(racc.x); w.y = __float_as_uint(racc.y); w.z = __float_as_uint(racc.z); w.w = __float_as_uint(racc.w);
                // (sc1 = write-through: needed when consumers run while this launch is resident (done != nullptr); behind a kernel boundary
                //  any policy is correct: store_mode, an experiment of round 6)
                const int smode = pc.done != nullptr ? 0 : pc.store_mode;
                if (smode == 1) __builtin_amdgcn_raw_buffer_store_b128(w, rr_, c4 * 16, 0, 0);
                else if (smode == 2) __builtin_amdgcn_raw_buffer_store_b128(w, rr_, c4 * 16, 0, 2 /* nt */);
                else __builtin_amdgcn_raw_buffer_store_b128(w, rr_, c4 * 16, 0, 16 /* sc1 */);
                if (pc.plane[0] != nullptr) {
                    const long nb = (pc.n_chunks - cb * pc.sub) < pc.sub ? (pc.n_chunks - cb * pc.sub) : pc.sub;   // chunks of this sub-batch
                    const long Mb = nb * op.rows, m = (c - cb * pc.sub) * op.rows + r;
                    const int col = 4 * c4;
                    const long base = cb * pc.sub * op.rows * (long)(4 * d4);                      // (uniform) first element of the sub-batch's planes
                    const int voff = (int)((((long)(col >> 4) * Mb + m) * 16 + (col & 15)) * 2);   // (per lane) byte offset inside them
                    __bf16 a[4], bb[4], dd[4];
                    pool_split3(racc.x, a[0], bb[0], dd[0]); pool_split3(racc.y, a[1], bb[1], dd[1]);
                    pool_split3(racc.z, a[2], bb[2], dd[2]); pool_split3(racc.w, a[3], bb[3], dd[3]);
                    uintx2 v0, v1, v2;
                    v0.x = pool_pack2(a[0], a[1]); v0.y = pool_pack2(a[2], a[3]);
                    v1.x = pool_pack2(bb[0], bb[1]); v1.y = pool_pack2(bb[2], bb[3]);
                    v2.x = pool_pack2(dd[0], dd[1]); v2.y = pool_pack2(dd[2], dd[3]);
                    const int span = (int)(Mb * (long)(4 * d4) * 2);
                    __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(static_cast<__bf16*>(pc.plane[0]) + base, 0, span, 0x00020000);
                    __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(static_cast<__bf16*>(pc.plane[1]) + base, 0, span, 0x00020000);
                    __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(static_cast<__bf16*>(pc.plane[2]) + base, 0, span, 0x00020000);
                    if (smode == 1) {
                        __builtin_amdgcn_raw_buffer_store_b64(v0, r0, voff, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b64(v1, r1, voff, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b64(v2, r2, voff, 0, 0);
                    } else if (smode == 2) {
                        __builtin_amdgcn_raw_buffer_store_b64(v0, r0, voff, 0, 2 /* nt */);
                        __builtin_amdgcn_raw_buffer_store_b64(v1, r1, voff, 0, 2 /* nt */);
                        __builtin_amdgcn_raw_buffer_store_b64(v2, r2, voff, 0, 2 /* nt */);
                    } else {
                    __builtin_amdgcn_raw_buffer_store_b64(v0, r0, voff, 0, 16 /* sc1 */);
                    __builtin_amdgcn_raw_buffer_store_b64(v1, r1, voff, 0, 16 /* sc1 */);
                    __builtin_amdgcn_raw_buffer_store_b64(v2, r2, voff, 0, 16 /* sc1 */);
                    }
                }
            }
            if (pc.done != nullptr) {                                   // (nullptr: one launch per sub-batch, the kernel boundary is the hand-off)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (threadIdx.x == 0) __hip_atomic_fetch_add(pc.done + cb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
#ifdef INFV_EXPERIMENTS
    if (stamps != nullptr) { __syncthreads(); wg_stamp_end(stamps); }
#endif
}

#ifdef INFV_EXPERIMENTS
static long long* g_stamps = nullptr;            // [kStampCap][4], launches appended
static long g_stamp_fill = 0;
constexpr long kStampCap = 1l << 19;
static int g_log_kind[1 << 12];                   // launch log: kind and record range of every stamped launch
static long g_log_first[1 << 12], g_log_n[1 << 12];
static int g_log_fill = 0;
long long* exp_stamps_reserve(int kind, long n_wgs) {
    static const bool want = exp_env("INFV_WG_STAMPS") != nullptr;
    if (!want) return nullptr;
    if (!g_stamps) {
        if (hipMalloc(&g_stamps, kStampCap * 4 * sizeof(long long)) != hipSuccess) return nullptr;
        (void)hipMemset(g_stamps, 0, kStampCap * 4 * sizeof(long long));
    }
    if (g_stamp_fill + n_wgs > kStampCap || g_log_fill >= (1 << 12)) return nullptr;
    long long* p = g_stamps + 4 * g_stamp_fill;
    g_log_kind[g_log_fill] = kind; g_log_first[g_log_fill] = g_stamp_fill; g_log_n[g_log_fill] = n_wgs; ++g_log_fill;
    g_stamp_fill += n_wgs;
    return p;
}
// copies the records out (kind filled in from the launch log) and rewinds
extern "C" long infv_exp_wg_stamps(long long* host, long cap) {
    if (!g_stamps) return 0;
    (void)hipDeviceSynchronize();
    const long n = g_stamp_fill < cap ? g_stamp_fill : cap;
    (void)hipMemcpy(host, g_stamps, (size_t)n * 4 * sizeof(long long), hipMemcpyDeviceToHost);
    for (int i = 0; i < g_log_fill; ++i)
        for (long j = g_log_first[i]; j < g_log_first[i] + g_log_n[i] && j < n; ++j) host[4 * j + 3] = g_log_kind[i];
    (void)hipMemset(g_stamps, 0, (size_t)n * 4 * sizeof(long long));
    g_stamp_fill = 0; g_log_fill = 0;
    return n;
}
#endif

// (2d. A variant of pool_rows2_kernel that streamed the token rows global -> LDS directly (`buffer_load ... lds`, round 3's
//  INFV_POOL_DMA=1) is gone: each wave read a piece right behind its own counted vmcnt wait, with no barrier in between -- LDS-DMA data
//  is ordered for a ds_read only by that wait FOLLOWED by a barrier (cdna_hip_programming.md, "read a staged buffer one phase after
//  the wait that retires it").  Round 5's variant test caught it: 1 run in 3 differed by 1.6e-4 in B, with round 4's library too.
//  It was never faster than the register-load kernel either.)

constexpr int kPoolTidAddr = 0;           // pool_rows2_kernel: lane-id addressed buffer loads (no vector address operand); INFV_POOL_TID in the experiments build

template <class Tok>
static hipError_t launch_pool_rows2_t(const void* k, int n_chunks, int T, int P, int d, const OperatorView& op, float* R,
                                      hipStream_t stream, int u, int lds_pad, int max_wgs, const PoolCallDesc* call = nullptr,
                                      void* const* planes = nullptr, bool* planes_done = nullptr) {
    PoolCallDesc pc;
    memset(&pc, 0, sizeof(pc));
    if (planes_done != nullptr) *planes_done = false;
    if (call != nullptr) { pc = *call; max_wgs = 0; }
    const int d4 = d / 4, slices = (d4 + 63) / 64;
    if (4 * slices * 64 > 1024) return hipErrorInvalidValue;
    const long n_rows_total = (long)n_chunks * op.rows;
    size_t lds = (size_t)4 * d4 * sizeof(floatx4);
    if ((size_t)lds_pad > lds) lds = lds_pad;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(pool_rows2_kernel<4, Tok, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(pool_rows2_kernel<8, Tok, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(pool_rows2_kernel<4, Tok, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(pool_rows2_kernel<8, Tok, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#ifdef INFV_EXPERIMENTS
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(pool_rows2_kernel<16, Tok, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(pool_rows2_kernel<32, Tok, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#endif
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    unsigned grid = (unsigned)n_rows_total;
    if (max_wgs > 0 && grid > (unsigned)max_wgs) grid = (unsigned)max_wgs;
    const dim3 block(4 * slices * 64);
    static const int prio = [] { const char* e = exp_env("INFV_POOL_PRIO"); return e ? atoi(e) : 0; }();
    PoolCallDesc one;                                                         // rows + planes of ONE sub-batch: the call-long kernel without a completion count
    if (call == nullptr && planes != nullptr && (long)grid == n_rows_total && n_rows_total * (long)d * 2 < (1l << 31)) {   // (a grid-stride grid writes rows only)
        memset(&one, 0, sizeof(one));
        one.sub = n_chunks; one.n_chunks = n_chunks; one.R_all = R;
        for (int i = 0; i < 3; ++i) one.plane[i] = planes[i];
        { static const int sm = [] { const char* e = exp_env("INFV_POOL_STORE"); return e ? atoi(e) : 0; }(); one.store_mode = sm; }
        call = &one; pc = one;
        if (planes_done != nullptr) *planes_done = true;
    }
    long long* stamps = exp_stamps_reserve(WG_POOL, grid);
    static const int want_tid = [] { const char* e = exp_env("INFV_POOL_TID"); return e ? atoi(e) : kPoolTidAddr; }();
    const int tid_addr = (want_tid && d4 % 64 == 0) ? 1 : 0;               // every lane of every slice holds a column
#ifdef INFV_EXPERIMENTS
    if (call == nullptr && u >= 32 && P % 32 == 0) { INFV_LAUNCH((pool_rows2_kernel<32, Tok, false>), dim3(grid), block, lds, stream, k, (long)T * P * d4, P, d4, slices, op, n_rows_total, R, stamps, prio, tid_addr, pc); return hipGetLastError(); }
    if (call == nullptr && u >= 16 && P % 16 == 0) { INFV_LAUNCH((pool_rows2_kernel<16, Tok, false>), dim3(grid), block, lds, stream, k, (long)T * P * d4, P, d4, slices, op, n_rows_total, R, stamps, prio, tid_addr, pc); return hipGetLastError(); }
#endif
    if (call != nullptr) {
        if (u >= 8 && P % 8 == 0) INFV_LAUNCH((pool_rows2_kernel<8, Tok, true>), dim3(grid), block, lds, stream, k, (long)T * P * d4, P, d4, slices, op, n_rows_total, R, stamps, prio, tid_addr, pc);
        else INFV_LAUNCH((pool_rows2_kernel<4, Tok, true>), dim3(grid), block, lds, stream, k, (long)T * P * d4, P, d4, slices, op, n_rows_total, R, stamps, prio, tid_addr, pc);
        return hipGetLastError();
    }
    if (u >= 8 && P % 8 == 0) INFV_LAUNCH((pool_rows2_kernel<8, Tok, false>), dim3(grid), block, lds, stream, k, (long)T * P * d4, P, d4, slices, op, n_rows_total, R, stamps, prio, tid_addr, pc);
    else INFV_LAUNCH((pool_rows2_kernel<4, Tok, false>), dim3(grid), block, lds, stream, k, (long)T * P * d4, P, d4, slices, op, n_rows_total, R, stamps, prio, tid_addr, pc);
    return hipGetLastError();
}

hipError_t launch_pool_rows2(const void* k, int k_bf16, int n_chunks, int T, int P, int d, const OperatorView& op, float* R,
                             hipStream_t stream, int u, int lds_pad, int max_wgs, void* const* planes, bool* planes_done) {
    if (planes_done != nullptr) *planes_done = false;
    if (op.rows == 0 || n_chunks == 0) return hipSuccess;
    if (planes != nullptr && d % 16 != 0) return hipErrorInvalidValue;
    return k_bf16 ? launch_pool_rows2_t<TokBF16>(k, n_chunks, T, P, d, op, R, stream, u, lds_pad, max_wgs, nullptr, planes, planes_done)
                  : launch_pool_rows2_t<TokF32>(k, n_chunks, T, P, d, op, R, stream, u, lds_pad, max_wgs, nullptr, planes, planes_done);
}

hipError_t launch_pool_rows2_call(const void* k, int k_bf16, int T, int P, int d, const OperatorView& op, const PoolCallDesc& pc,
                                  hipStream_t stream, int u, int lds_pad) {
    if (op.rows == 0 || pc.n_chunks == 0) return hipSuccess;
    if (pc.R_all == nullptr || pc.sub <= 0) return hipErrorInvalidValue;
    return k_bf16 ? launch_pool_rows2_t<TokBF16>(k, pc.n_chunks, T, P, d, op, nullptr, stream, u, lds_pad, 0, &pc)
                  : launch_pool_rows2_t<TokF32>(k, pc.n_chunks, T, P, d, op, nullptr, stream, u, lds_pad, 0, &pc);
}

bool pool_rows2_supported(int d) { return d % 4 == 0 && ((d / 4 + 63) / 64) * 4 * 64 <= 1024; }

// ======================================================================================
// 3. projection GEMM (NT):  C[sk][m][o] = sum_{k in split sk} A[m][k] * Wrow(o)[k]
//    A [M][K] row-major; Wrow(o) = row (o % rows_per_seg) of segment o / rows_per_seg (a list of row-major
//    [rows_per_seg][K] matrices: the layers' Wk / Wv, or the pre-multiplied queries of the fast path).
//    fp32 MFMA 32x32x2, block tile BM x BN x 32, 4 waves as 2x2, register-staged prefetch.
// ======================================================================================
constexpr int kBK = 32;
constexpr int kLdsStride = kBK + 4;   // +1 access width (16 B) against ds_read_b128 conflicts

// A operand of the per-call step (ROWS): row m of A is not read but built on the fly from the chunk's pooled frames,
//   A[m][:] = sum_{f in [row_begin[m], row_end[m])} val_m * kbar[f][:]      (build_rows_kernel's fma chain, same bits)
struct RowsA {
    const float* kbar; const int32_t* row_begin; const int32_t* row_end; const int32_t* row_box; const float* box_val;
};

template <int BM, int BN, bool ROWS>
__device__ inline void gemm_nt_tile(const float* __restrict__ A, const RowsA& ra, int M, int K, const WSegs& segs,
                                    float* __restrict__ C, int ldc, long split_stride, int k_per_split, int y_off,
                                    float* As, float* Bs) {
    constexpr int TM = BM / 64, TN = BN / 64;          // 32x32 tiles per wave in each dim
    constexpr int AR = BM / 32, BR = BN / 32;          // rows staged per thread

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * BM, n0 = (blockIdx.y + y_off) * BN;
    const int kbeg = blockIdx.z * k_per_split;
    const int ntiles = k_per_split / kBK;
    C += (long)blockIdx.z * split_stride;

    // staging map: thread -> (row0 + 32*i, float4 column c4)
    const int c4 = tid & 7, row0 = tid >> 3;
    const float* a_src[AR];
    const float* b_src[BR];
    int fb[AR], fe[AR]; float aval[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        const int m = m0 + row0 + 32 * i;
        fb[i] = fe[i] = 0; aval[i] = 0.f;
        if (ROWS) {
            a_src[i] = ra.kbar + kbeg + c4 * 4;
            if (m < M) { fb[i] = ra.row_begin[m]; fe[i] = ra.row_end[m]; aval[i] = ra.box_val[ra.row_box[m]]; }
        } else {
            a_src[i] = (m < M) ? A + (long)m * K + kbeg + c4 * 4 : nullptr;
        }
    }
#pragma unroll
    for (int i = 0; i < BR; ++i) {
        const int o = n0 + row0 + 32 * i;
        int seg = 0;
        while (o >= segs.start[seg + 1]) ++seg;
        b_src[i] = segs.base[seg] + (long)(o - segs.start[seg]) * K + kbeg + c4 * 4;
    }

    // Two tiles in flight (the tile staged into LDS at step t was requested at step t-2).  It buys no time by itself --
    // the kernel is MFMA-paced -- but the 92 extra VGPRs make a workgroup of this kernel too large to share a SIMD with
    // the two resident waves of role S (chain_batch2_kernel): the GEMMs co-reside with the pooling kernel only, never with
    // the latency-critical chain (measured: 112 k chunks/s against 102 k with the one-tile version).
    floatx4 a_reg[2][AR], b_reg[2][BR];
    auto load_tile = [&](int t, floatx4 (&ar)[AR], floatx4 (&br)[BR]) {
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            if (ROWS) {
                // four frames per trip, loaded unconditionally (index clamped into the row's range) so that the loads are in
                // flight together; a frame past the end contributes fma(0, v, acc) = acc: build_rows_kernel's chain, same bits
                floatx4 acc = {0.f, 0.f, 0.f, 0.f};
                for (int f0 = fb[i]; f0 < fe[i]; f0 += 4) {
                    floatx4 v[4]; float w[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int f = min(f0 + j, fe[i] - 1);
                        v[j] = *reinterpret_cast<const floatx4*>(a_src[i] + (long)f * K + t * kBK);
                        w[j] = (f0 + j < fe[i]) ? aval[i] : 0.f;
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc.x = fmaf(w[j], v[j].x, acc.x); acc.y = fmaf(w[j], v[j].y, acc.y);
                        acc.z = fmaf(w[j], v[j].z, acc.z); acc.w = fmaf(w[j], v[j].w, acc.w);
                    }
                }
                ar[i] = acc;
            } else {
                ar[i] = a_src[i] ? *reinterpret_cast<const floatx4*>(a_src[i] + t * kBK)
                                 : floatx4{0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int i = 0; i < BR; ++i) br[i] = *reinterpret_cast<const floatx4*>(b_src[i] + t * kBK);
    };
    auto store_tile = [&](const floatx4 (&ar)[AR], const floatx4 (&br)[BR]) {
#pragma unroll
        for (int i = 0; i < AR; ++i)
            *reinterpret_cast<floatx4*>(&As[(row0 + 32 * i) * kLdsStride + c4 * 4]) = ar[i];
#pragma unroll
        for (int i = 0; i < BR; ++i)
            *reinterpret_cast<floatx4*>(&Bs[(row0 + 32 * i) * kLdsStride + c4 * 4]) = br[i];
    };

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int li = lane & 31, kk = lane >> 5;     // MFMA 32x32x2: row/col = lane&31, k = lane>>5
    auto compute_tile = [&]() {
        // lane half kk consumes k = 16*kk + s at MFMA step s (any bijection of k works as long
        // as A and B agree): 4 x ds_read_b128 per operand tile.
        floatx4 af[TM][4], bf[TN][4];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int v = 0; v < 4; ++v)
                af[i][v] = *reinterpret_cast<const floatx4*>(
                    &As[(wm * (BM / 2) + i * 32 + li) * kLdsStride + 16 * kk + 4 * v]);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int v = 0; v < 4; ++v)
                bf[j][v] = *reinterpret_cast<const floatx4*>(
                    &Bs[(wn * (BN / 2) + j * 32 + li) * kLdsStride + 16 * kk + 4 * v]);
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s >> 2][s & 3], bf[j][s >> 2][s & 3],
                                                                     acc[i][j], 0, 0, 0);
    };
    load_tile(0, a_reg[0], b_reg[0]);
    if (ntiles > 1) load_tile(1, a_reg[1], b_reg[1]);
    for (int t = 0; t < ntiles; t += 2) {
        store_tile(a_reg[0], b_reg[0]);
        __syncthreads();
        if (t + 2 < ntiles) load_tile(t + 2, a_reg[0], b_reg[0]);
        compute_tile();
        __syncthreads();
        if (t + 1 < ntiles) {
            store_tile(a_reg[1], b_reg[1]);
            __syncthreads();
            if (t + 3 < ntiles) load_tile(t + 3, a_reg[1], b_reg[1]);
            compute_tile();
            __syncthreads();
        }
    }
    // C/D map of 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
                const int o = n0 + wn * (BN / 2) + j * 32 + li;
                if (m < M) __builtin_nontemporal_store(acc[i][j][r], &C[(long)m * ldc + o]);
            }
}

template <int BM, int BN>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const float* __restrict__ A, int M, int K,
                                                      WSegs segs,
                                                      float* __restrict__ C, int ldc, long split_stride,
                                                      int k_per_split, int y_off) {
    __shared__ float As[BM * kLdsStride];
    __shared__ float Bs[BN * kLdsStride];
    gemm_nt_tile<BM, BN, false>(A, RowsA{}, M, K, segs, C, ldc, split_stride, k_per_split, y_off, As, Bs);
}

// Per-call step, first launch: the new-row projection with the rows built on the fly (no rows kernel, no R buffer) on the
// planes z < splitk, and -- independent of it -- the Gibbs draw of every layer on the extra plane z == splitk (one workgroup
// per layer; the other workgroups of that plane exit at once).  Two launches and two kernel boundaries fewer per forward.
__global__ __launch_bounds__(256) void step_project_kernel(RowsA ra, int M, int K, WSegs segs, float* __restrict__ C, int ldc,
                                                           long split_stride, int k_per_split, int splitk, StepDraw dr) {
    __shared__ __attribute__((aligned(16))) float smem[2 * 64 * kLdsStride];
    if ((int)blockIdx.z == splitk) {
        const int l = blockIdx.y;
        if (blockIdx.x != 0 || l >= dr.n_layers) return;
        static_assert(2 * 64 * kLdsStride * sizeof(float) >= kBins * sizeof(float) + 256 * sizeof(double) + 1024 * sizeof(int32_t),
                      "draw scratch fits in the GEMM's tiles");
        double* gsum = reinterpret_cast<double*>(smem);                  // [256]
        int32_t* sidx = reinterpret_cast<int32_t*>(smem + 512);          // [1024]
        float* cdf = smem + 512 + 1024;                                  // [kBins]
        const bool ovr = (dr.override_mask >> l) & 1u;
        const DrawRegs<4> r = draw_load<256, 4>(dr.bin_part + (long)l * dr.parts * kBins, dr.parts, nullptr,
                                                dr.probs_override + l * kBins, ovr, dr.u + (long)l * dr.S, dr.S);
        draw_finish<256, 4>(r, ovr, dr.sticky.bin_box, dr.S, cdf, sidx, gsum, dr.probs_out + l * kBins, dr.bins_out + (long)l * dr.S,
                            dr.idx_out + (long)l * dr.S, nullptr, ((dr.forced_mask >> l) & 1u) ? dr.bins_forced + (long)l * dr.S : nullptr);
        // resolved gather table for the update kernel (draw_finish ends with a barrier: sidx holds every slot's source box)
        if (dr.tab_out != nullptr)
            for (int e = threadIdx.x; e < dr.tab_entries; e += 256) {
                const int sl = dr.slot_tab[e];
                dr.tab_out[(long)l * dr.tab_entries + e] = sl >= 0 ? sidx[sl] : -1;
            }
        return;
    }
    gemm_nt_tile<64, 64, true>(nullptr, ra, M, K, segs, C, ldc, split_stride, k_per_split, 0, smem, smem + 64 * kLdsStride);
}

// ======================================================================================
// 3b. The same GEMM with dedicated LOADER waves (sub-batch projections of the fast path; INFV_GEMM_LW=0 disables).  While the pooling kernel streams on the same CU every
//     vector-memory instruction waits ~0.2 us at issue; in gemm_nt_kernel the wave that issues a tile's 8 loads is the
//     wave that should be issuing its 64 MFMAs, so the kernel runs at 50-55 TFLOP/s in situ against 88 alone.  Here waves
//     0-3 only read LDS and issue MFMAs (same fragment reads, same MFMA order: same bits) and waves 4-7 only move tiles:
//     global -> registers three tiles ahead (three register sets) -> LDS one tile ahead (two buffers), ONE barrier per
//     k-tile, which orders LDS traffic only (no vmcnt drain).  Measured in situ (2048-chunk video): the two projection GEMMs
//     10.3 -> 7.1 ms busy, 111 k -> 115 k chunks/s (with the pooling kernel's padding LDS at 84 KB, so that a 74 KB
//     workgroup of this kernel still fits beside a pooling workgroup).
// ======================================================================================
#ifndef INFV_LW_WAVES
#define INFV_LW_WAVES 2
#endif
__device__ inline void lds_only_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int BM, int BN, int NT>
__global__ __launch_bounds__(512, INFV_LW_WAVES) void gemm_nt_lw_kernel(const float* __restrict__ A, int M, int K, WSegs segs,
                                                         float* __restrict__ C, int ldc, long split_stride,
                                                         long long* __restrict__ stamps, int y0) {
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int AR = BM / 32, BR = BN / 32;
    constexpr int kBuf = (BM + BN) * kLdsStride;       // floats per LDS buffer
    wg_stamp_begin(stamps);
    extern __shared__ __attribute__((aligned(16))) float lw_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * BM, n0 = (blockIdx.y + y0) * BN;    // (y0: first column tile of this launch, when the launch is one slice of the GEMM)
    constexpr int ntiles = NT;                         // k-tiles per split: a compile-time constant, so that the loader's
    const int kbeg = blockIdx.z * (NT * kBK);          // loop unrolls completely and every vmcnt wait is an exact count
    C += (long)blockIdx.z * split_stride;

    if (wave >= 4) {
        // ---------------- loader waves ----------------
        const int lt = tid - 256;
        const int c4 = lt & 7, row0 = lt >> 3;
        const float* a_src[AR];
        const float* b_src[BR];
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const int m = m0 + row0 + 32 * i;
            a_src[i] = A + (long)(m < M ? m : M - 1) * K + kbeg + c4 * 4;     // rows past M: clamped (their outputs are dropped)
        }
#pragma unroll
        for (int i = 0; i < BR; ++i) {
            const int o = n0 + row0 + 32 * i;
            int seg = 0;
            while (o >= segs.start[seg + 1]) ++seg;
            b_src[i] = segs.base[seg] + (long)(o - segs.start[seg]) * K + kbeg + c4 * 4;
        }
        floatx4 a0[AR], b0[BR], a1[AR], b1[BR], a2[AR], b2[BR];
#define INFV_LW_LOAD(t, ar, br)                                                                              \
        {                                                                                                    \
            _Pragma("unroll") for (int i = 0; i < AR; ++i) ar[i] = *reinterpret_cast<const floatx4*>(a_src[i] + (t) * kBK); \
            _Pragma("unroll") for (int i = 0; i < BR; ++i) br[i] = *reinterpret_cast<const floatx4*>(b_src[i] + (t) * kBK); \
        }
#define INFV_LW_STORE(t, ar, br)                                                                             \
        {                                                                                                    \
            float* As_ = lw_smem + ((t) & 1) * kBuf;                                                         \
            float* Bs_ = As_ + BM * kLdsStride;                                                              \
            _Pragma("unroll") for (int i = 0; i < AR; ++i)                                                   \
                *reinterpret_cast<floatx4*>(&As_[(row0 + 32 * i) * kLdsStride + c4 * 4]) = ar[i];            \
            _Pragma("unroll") for (int i = 0; i < BR; ++i)                                                   \
                *reinterpret_cast<floatx4*>(&Bs_[(row0 + 32 * i) * kLdsStride + c4 * 4]) = br[i];            \
        }
        // NT % 3 == 0.  Straight-line code (full unroll): in a rolled loop, or with loads behind conditions, the compiler's
        // wait-count analysis falls back to vmcnt(0) at the first store of an iteration, which drains the two tiles that
        // should stay in flight.
        static_assert(NT % 3 == 0 && NT >= 3, "three register sets");
        constexpr int last = NT - 1;
        INFV_LW_LOAD(0, a0, b0);
        INFV_LW_LOAD(1, a1, b1);
        INFV_LW_LOAD(2, a2, b2);
#pragma unroll
        for (int t = 0; t < NT; t += 3) {
            INFV_LW_STORE(t, a0, b0);
            INFV_LW_LOAD(t + 3 < last ? t + 3 : last, a0, b0);
            lds_only_barrier();
            INFV_LW_STORE(t + 1, a1, b1);
            INFV_LW_LOAD(t + 4 < last ? t + 4 : last, a1, b1);
            lds_only_barrier();
            INFV_LW_STORE(t + 2, a2, b2);
            INFV_LW_LOAD(t + 5 < last ? t + 5 : last, a2, b2);
            lds_only_barrier();
        }
#undef INFV_LW_LOAD
#undef INFV_LW_STORE
        return;
    }
    // ---------------- MFMA waves ----------------
    const int wm = wave >> 1, wn = wave & 1;
    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int li = lane & 31, kk = lane >> 5;
    for (int t = 0; t < ntiles; ++t) {
        lds_only_barrier();                              // tile t is in buffer t & 1
        const float* As = lw_smem + (t & 1) * kBuf;
        const float* Bs = As + BM * kLdsStride;
        floatx4 af[TM][4], bf[TN][4];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int v = 0; v < 4; ++v)
                af[i][v] = *reinterpret_cast<const floatx4*>(&As[(wm * (BM / 2) + i * 32 + li) * kLdsStride + 16 * kk + 4 * v]);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int v = 0; v < 4; ++v)
                bf[j][v] = *reinterpret_cast<const floatx4*>(&Bs[(wn * (BN / 2) + j * 32 + li) * kLdsStride + 16 * kk + 4 * v]);
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s >> 2][s & 3], bf[j][s >> 2][s & 3], acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
                const int o = n0 + wn * (BN / 2) + j * 32 + li;
                if (m < M) __builtin_nontemporal_store(acc[i][j][r], &C[(long)m * ldc + o]);
            }
    wg_stamp_end(stamps);                              // (wave 0's last store issued: the other matrix waves end within a tile of it)
}

static void segs_clear(WSegs& s) {
    for (int i = 0; i < kMaxSegs; ++i) s.base[i] = nullptr;
    for (int i = 0; i <= kMaxSegs; ++i) s.start[i] = 0x7fffffff;
    s.start[0] = 0;
}
static void segs_push(WSegs& s, int& n, const float* base, int rows) {
    s.base[n] = base;
    s.start[n + 1] = s.start[n] + rows;
    ++n;
}
static WSegs kv_segs(const ProjPtrs& proj, int layer_base, int n_layers, int dm) {
    WSegs s;
    segs_clear(s);
    int n = 0;
    for (int l = 0; l < n_layers; ++l) { segs_push(s, n, proj.wk[layer_base + l], dm); segs_push(s, n, proj.wv[layer_base + l], dm); }
    return s;
}

static hipError_t launch_gemm(const float* A, int M, int K, const WSegs& segs,
                              int n_cols, float* C, int ldc, int splitk, long split_stride,
                              hipStream_t stream, int lds_pad = 0) {
    if (M <= 0) return hipSuccess;
    const int k_per_split = K / splitk;
    static bool attr_set = false;
    if (lds_pad > 0 && !attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_kernel<128, 128>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_kernel<64, 64>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    static const bool want_lw = [] { const char* e = exp_env("INFV_GEMM_LW"); return !e || atoi(e) != 0; }();   // default on
    if (want_lw && M >= 1024 && lds_pad <= 0 && k_per_split == 24 * kBK) {
        static bool attr_lw = false;
        if (!attr_lw) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_lw_kernel<128, 128, 24>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_lw_kernel<64, 128, 24>),
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            if (e != hipSuccess) return e;
            attr_lw = true;
        }
        if (((M + 127) / 128) * (n_cols / 128) <= 160 && ((M + 63) / 64) * (n_cols / 128) <= 256) {
            dim3 grid((M + 63) / 64, n_cols / 128, splitk);
            INFV_LAUNCH((gemm_nt_lw_kernel<64, 128, 24>), grid, dim3(512), 2 * (64 + 128) * kLdsStride * sizeof(float), stream,
                               A, M, K, segs, C, ldc, split_stride, exp_stamps_reserve(WG_GEMM, (long)grid.x * grid.y * grid.z), 0);
        } else {
            // Column slices launched one after the other on the stream (experiment INFV_GEMM_SLICES): a launch whose workgroups do not
            // all find a seat leaves a backlog at the dispatcher, and while it stands the pooling stream's workgroups are not
            // dispatched either (tools/residency.py timelines); slices of <= ~200 workgroups never queue.
            static const int slices = [] { const char* e = exp_env("INFV_GEMM_SLICES"); return e ? atoi(e) : 1; }();
            // (experiment INFV_GEMM_LW_LDS: total dynamic LDS of the launch; 90 KB keeps pooling workgroups off the GEMM's CUs)
            static const size_t lw_lds = [] { const char* e = exp_env("INFV_GEMM_LW_LDS"); return e ? (size_t)atol(e) : (size_t)0; }();
            size_t lds_bytes = 2 * (128 + 128) * kLdsStride * sizeof(float);
            if (lw_lds > lds_bytes && lw_lds <= 96 * 1024) lds_bytes = lw_lds;
            const int gy = n_cols / 128;
            for (int sl = 0; sl < slices; ++sl) {
                const int y0 = gy * sl / slices, y1 = gy * (sl + 1) / slices;
                if (y1 <= y0) continue;
                dim3 grid((M + 127) / 128, y1 - y0, splitk);
                INFV_LAUNCH((gemm_nt_lw_kernel<128, 128, 24>), grid, dim3(512), lds_bytes, stream,
                                   A, M, K, segs, C, ldc, split_stride, exp_stamps_reserve(WG_GEMM, (long)grid.x * grid.y * grid.z), y0);
            }
        }
        return hipGetLastError();
    }
    if (M >= 1024 && ((M + 127) / 128) * (n_cols / 128) <= 160 && ((M + 63) / 64) * (n_cols / 128) <= 256) {
        // few column tiles (the score half of the fast path: 126 tiles of 128 x 128 would leave half the CUs idle):
        // 64-row tiles double the workgroups of the one round
        const int gx = (M + 63) / 64, gy = n_cols / 128;
        static bool attr64 = false;
        if (lds_pad > 0 && !attr64) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_kernel<64, 128>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
            if (e != hipSuccess) return e;
            attr64 = true;
        }
        dim3 grid(gx, gy, splitk);
        INFV_LAUNCH((gemm_nt_kernel<64, 128>), grid, dim3(256), lds_pad, stream, A, M, K, segs,
                           C, ldc, split_stride, k_per_split, 0);
    } else if (M >= 1024) {
        const int gx = (M + 127) / 128, gy = n_cols / 128;
        if (lds_pad > 0) {
            // overlapped mode: one workgroup per CU (padding LDS) and never more workgroups than CUs in a
            // launch, so no workgroup of this kernel ever waits for a slot: a kernel with pending workgroups
            // was measured to stall the dispatch of the concurrent chain kernel for tens of microseconds.
            int ny = 256 / gx;                                   // column tiles per launch
            if (ny < 1) ny = 1;
            for (int y0 = 0; y0 < gy; y0 += ny) {
                dim3 grid(gx, (gy - y0 < ny) ? gy - y0 : ny, splitk);
                INFV_LAUNCH((gemm_nt_kernel<128, 128>), grid, dim3(256), lds_pad, stream, A, M, K, segs,
                                   C, ldc, split_stride, k_per_split, y0);
            }
        } else {
            dim3 grid(gx, gy, splitk);
            INFV_LAUNCH((gemm_nt_kernel<128, 128>), grid, dim3(256), 0, stream, A, M, K, segs,
                               C, ldc, split_stride, k_per_split, 0);
        }
    } else {
        dim3 grid((M + 63) / 64, n_cols / 64, splitk);
        INFV_LAUNCH((gemm_nt_kernel<64, 64>), grid, dim3(256), 0, stream, A, M, K, segs,
                           C, ldc, split_stride, k_per_split, 0);
    }
    return hipGetLastError();
}

int project_splitk(int M, int K) {
    // small-M calls (one chunk) cannot fill 256 CUs with output tiles alone: split K.
    if (M >= 1024) return 1;
    int sk = 8;
    while (sk > 1 && (K % (sk * kBK)) != 0) sk >>= 1;
    return sk;
}

hipError_t launch_rows(const float* kbar, int n_chunks, int T, int d, const OperatorView& op, float* R,
                       hipStream_t stream) {
    if (op.rows == 0 || n_chunks == 0) return hipSuccess;
    INFV_LAUNCH(build_rows_kernel, dim3(op.rows, n_chunks), dim3(256), 0, stream, kbar, T, d / 4, op, R);
    return hipGetLastError();
}

hipError_t launch_project(int n_chunks, int d, int dm, int n_layers, const OperatorView& op, const ProjPtrs& proj,
                          const float* R, float* Pnew, hipStream_t stream, int lds_pad) {
    if (op.rows == 0 || n_chunks == 0) return hipSuccess;
    const int M = n_chunks * op.rows;
    const int n_cols = n_layers * 2 * dm;
    const int sk = project_splitk(M, d);
    return launch_gemm(R, M, d, kv_segs(proj, 0, n_layers, dm), n_cols, Pnew, n_cols, sk, (long)M * n_cols, stream, lds_pad);
}

// Per-call step: Pnew[sk][r][l][kv][dm] for the rows of ONE chunk straight from its pooled frames (+ the draw, see
// step_project_kernel).  Same split-K slabs as launch_project.
hipError_t launch_step_project(const float* kbar, int d, int dm, int n_layers, const OperatorView& op, const ProjPtrs& proj,
                               float* Pnew, int* splitk, const StepDraw& draw, hipStream_t stream) {
    const int M = op.rows, n_cols = n_layers * 2 * dm;
    const int sk = project_splitk(M > 0 ? M : 1, d);
    *splitk = sk;
    if (M == 0 && draw.n_layers == 0) return hipSuccess;
    if (n_cols % 64 || (d / sk) % kBK) return hipErrorInvalidValue;
    RowsA ra{kbar, op.row_begin, op.row_end, op.row_box, op.box_val};
    int gy = n_cols / 64;
    if (gy < draw.n_layers) gy = draw.n_layers;
    dim3 grid((M + 63) / 64 > 0 ? (M + 63) / 64 : 1, gy, sk + 1);
    INFV_LAUNCH(step_project_kernel, grid, dim3(256), 0, stream, ra, M, d, kv_segs(proj, 0, n_layers, dm), Pnew, n_cols,
                       (long)M * n_cols, d / sk, sk, draw);
    return hipGetLastError();
}

// Fast path: ONE GEMM for the V' half of the new rows and their scores under the pre-multiplied queries qt:
//   C[sk][m][0 : L*dm)           = R[m] . Wv[l]^T                       (projection of the new rows, V' half only)
//   C[sk][m][L*dm : L*dm + n_out) = R[m] . qt[o],  o = (l*H + h)*Q + q   (new-row scores; the K half is never formed)
hipError_t launch_project_fast(int M, int d, int dm, int n_layers, int n_out, const ProjPtrs& proj, const float* qt,
                               const float* R, float* C, int* splitk, hipStream_t stream, int lds_pad) {
    const int n_cols = n_layers * dm + n_out;
    *splitk = project_splitk(M, d);
    if (M == 0) return hipSuccess;
    WSegs s;
    segs_clear(s);
    int n = 0;
    for (int l = 0; l < n_layers; ++l) segs_push(s, n, proj.wv[l], dm);
    segs_push(s, n, qt, n_out);
    return launch_gemm(R, M, d, s, n_cols, C, n_cols, *splitk, (long)M * n_cols, stream, lds_pad);
}

// Score columns only (exact fp32: their rounding feeds the bit-exact draw): C[m][o] = R[m] . qt[o], o < n_out
hipError_t launch_project_scores(int M, int d, int n_out, const float* qt, const float* R, float* C, int ldc,
                                 hipStream_t stream, int lds_pad) {
    if (M == 0) return hipSuccess;
    WSegs s;
    segs_clear(s);
    int n = 0;
    segs_push(s, n, qt, n_out);
    return launch_gemm(R, M, d, s, n_out, C, ldc, 1, 0, stream, lds_pad);
}

// Value columns only: C[m][l*dm + o] = R[m] . Wv[l][o]   (the V' half of the new rows, consumed by the UC kernel)
hipError_t launch_project_values(int M, int d, int dm, int n_layers, const ProjPtrs& proj, const float* R, float* C,
                                 int ldc, hipStream_t stream, int lds_pad) {
    if (M == 0) return hipSuccess;
    WSegs s;
    segs_clear(s);
    int n = 0;
    for (int l = 0; l < n_layers; ++l) segs_push(s, n, proj.wv[l], dm);
    return launch_gemm(R, M, d, s, n_layers * dm, C, ldc, 1, 0, stream, lds_pad);
}

// qt[(l*H + h)*Q + q][:] = sum_e (q[l][q][h*64+e] / sqrt(dh)) * Wk[l][h*64+e][:]   and   cq[(l*H+h)*Q + q] = q_h . bk_h / sqrt(dh)
// so that  S'new = (q/sqrt(dh)) . (R Wk^T)_h^T = R . qt^T  without projecting the K half of the new rows.
__global__ __launch_bounds__(256) void qtilde_kernel(const float* __restrict__ q, int Q, int H, int d, ProjPtrs proj,
                                                     float* __restrict__ qt, float* __restrict__ cq) {
    extern __shared__ float qs[];                      // [Q][64] scaled query slice of this head
    const int h = blockIdx.x, l = blockIdx.y;
    const int dm = H * kHeadSize;
    const float scale = 1.0f / sqrtf((float)kHeadSize);
    for (int e = threadIdx.x; e < Q * kHeadSize; e += 256) {
        const int r = e / kHeadSize, c = e - r * kHeadSize;
        qs[e] = q[((long)l * Q + r) * dm + h * kHeadSize + c] * scale;
    }
    __syncthreads();
    const float* wk = proj.wk[l] + (long)h * kHeadSize * d;
    const float* bk = proj.bk[l] + h * kHeadSize;
    const long row0 = ((long)l * H + h) * Q;
    if (blockIdx.z == 0 && (int)threadIdx.x < Q) {
        float c = 0.f;
        for (int e = 0; e < kHeadSize; ++e) c = fmaf(qs[threadIdx.x * kHeadSize + e], bk[e], c);
        cq[row0 + threadIdx.x] = c;
    }
    // one 256-column block per workgroup; the 64 weights of a column are fetched up front (independent loads)
    const int k = blockIdx.z * 256 + threadIdx.x;
    if (k >= d) return;
    float w[kHeadSize];
#pragma unroll
    for (int e = 0; e < kHeadSize; ++e) w[e] = wk[(long)e * d + k];
    for (int r = 0; r < Q; ++r) {
        float acc = 0.f;
#pragma unroll
        for (int e = 0; e < kHeadSize; ++e) acc = fmaf(qs[r * kHeadSize + e], w[e], acc);
        qt[(row0 + r) * d + k] = acc;
    }
}

hipError_t launch_qtilde(const float* q, int Q, int H, int d, int n_layers, const ProjPtrs& proj, float* qt, float* cq,
                         hipStream_t stream) {
    INFV_LAUNCH(qtilde_kernel, dim3(H, n_layers, (d + 255) / 256), dim3(256), (size_t)Q * kHeadSize * sizeof(float), stream, q, Q, H,
                       d, proj, qt, cq);
    return hipGetLastError();
}

// ======================================================================================
// 4. Gibbs / sticky draw, one workgroup per layer (draw_core in ltm_device.h; LTM.py:202-208)
// ======================================================================================
__global__ __launch_bounds__(256) void draw_kernel(const float* __restrict__ bin_part, int parts,
                                                   const float* __restrict__ probs_override,
                                                   unsigned override_mask, StickyView sticky,
                                                   const double* __restrict__ u, int S,
                                                   float* __restrict__ probs_out, int32_t* __restrict__ bins_out,
                                                   int32_t* __restrict__ idx_out,
                                                   const int32_t* __restrict__ bins_forced, unsigned forced_mask) {
    __shared__ float cdf[kBins];
    __shared__ double gsum[256];
    __shared__ int32_t sidx[1024];
    const int l = blockIdx.x;
    const bool ovr = (override_mask >> l) & 1u;
    const DrawRegs<4> r = draw_load<256, 4>(bin_part + (long)l * parts * kBins, parts, nullptr, probs_override + l * kBins,
                                            ovr, u + (long)l * S, S);
    draw_finish<256, 4>(r, ovr, sticky.bin_box, S, cdf, sidx, gsum, probs_out + l * kBins, bins_out + (long)l * S,
                        idx_out + (long)l * S, nullptr, ((forced_mask >> l) & 1u) ? bins_forced + (long)l * S : nullptr);
}

hipError_t launch_draw(const float* bin_part, int parts, const float* probs_override, unsigned override_mask,
                       const StickyView& sticky, const double* u, int S, int n_layers, float* probs,
                       int32_t* bins, int32_t* idx, hipStream_t stream, const int32_t* bins_forced, unsigned forced_mask) {
    if (S > 1024) return hipErrorInvalidValue;
    INFV_LAUNCH(draw_kernel, dim3(n_layers), dim3(256), 0, stream, bin_part, parts, probs_override,
                       override_mask, sticky, u, S, probs, bins, idx, bins_forced, forced_mask);
    return hipGetLastError();
}

// ======================================================================================
// 5. memory update: one workgroup per (box n, layer), three row families at once.
//    next[n] = val_n * sum_{s in slots(n)} prev[idx[s]]  +  new row of box n
// ======================================================================================
__global__ __launch_bounds__(1024) void update_kernel(OperatorView op, int N, int d4, int dm4, int n_layers,
                                                     int S, const int32_t* __restrict__ idx,
                                                     int idx_layer_stride, const float* __restrict__ R,
                                                     const float* __restrict__ Pnew, int splitk,
                                                     long split_stride4,
                                                     const float* __restrict__ B_prev,
                                                     const float* __restrict__ KV_prev,
                                                     float* __restrict__ B_next, float* __restrict__ KV_next,
                                                     const float* __restrict__ kbar, const int32_t* __restrict__ tab) {
    const int n = blockIdx.x, l = blockIdx.y;
    const float val = op.box_val[n];
    const int row = op.box_row[n];
    int fb = 0, fe = 0;                                         // R == nullptr: the new row of B is built here from the pooled frames
    if (R == nullptr && row >= 0) { fb = op.row_begin[row]; fe = op.row_end[row]; }
    const floatx4* kb4 = reinterpret_cast<const floatx4*>(kbar);
    int sb = 0, se = 0;
    // with a resolved table the box's sources are one row of `tab` (same slots in the same order as the CSR walk below)
    const int32_t* my_tab = tab ? tab + ((long)l * N + n) * op.tabw : nullptr;
    if (my_tab != nullptr) { se = op.tabw; }
    else if (op.old_ptr != nullptr && idx != nullptr) { sb = op.old_ptr[n]; se = op.old_ptr[n + 1]; }
    const int32_t* my_idx = idx ? idx + (long)l * idx_layer_stride : nullptr;
    const int kv4 = 2 * dm4;                                   // floats4 of a [K'|V'] row
    const int total4 = d4 + kv4;
    const floatx4* Bp = reinterpret_cast<const floatx4*>(B_prev) + (long)l * N * d4;
    const floatx4* KVp = reinterpret_cast<const floatx4*>(KV_prev) + (long)l * N * kv4;
    floatx4* Bn = reinterpret_cast<floatx4*>(B_next) + ((long)l * N + n) * d4;
    floatx4* KVn = reinterpret_cast<floatx4*>(KV_next) + ((long)l * N + n) * kv4;
    const floatx4* R4 = reinterpret_cast<const floatx4*>(R);
    const floatx4* P4 = reinterpret_cast<const floatx4*>(Pnew);
    for (int c = threadIdx.x; c < total4; c += blockDim.x) {
        const bool isB = c < d4;
        const int cc = isB ? c : c - d4;
        const int pitch = isB ? d4 : kv4;
        const floatx4* prev = isB ? Bp : KVp;
        floatx4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int s0 = sb; s0 < se; s0 += 4) {           // 4 gathered rows in flight at a time
            floatx4 v[4];
            int src[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                src[k] = (s0 + k < se) ? (my_tab != nullptr ? my_tab[s0 + k] : my_idx[op.old_slot[s0 + k]]) : -1;
                v[k] = prev[(long)max(src[k], 0) * pitch + cc];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (src[k] >= 0) {
                    acc.x = fmaf(val, v[k].x, acc.x); acc.y = fmaf(val, v[k].y, acc.y);
                    acc.z = fmaf(val, v[k].z, acc.z); acc.w = fmaf(val, v[k].w, acc.w);
                }
        }
        if (row >= 0) {
            if (isB) {
                if (R != nullptr) {
                    acc += R4[(long)row * d4 + cc];
                } else {                                        // build_rows_kernel's fma chain, same bits
                    floatx4 rr = {0.f, 0.f, 0.f, 0.f};
                    for (int f0 = fb; f0 < fe; f0 += 4) {       // four frames in flight (clamped index, zero weight past the end)
                        floatx4 v[4]; float w[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            v[j] = kb4[(long)min(f0 + j, fe - 1) * d4 + cc];
                            w[j] = (f0 + j < fe) ? val : 0.f;
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            rr.x = fmaf(w[j], v[j].x, rr.x); rr.y = fmaf(w[j], v[j].y, rr.y);
                            rr.z = fmaf(w[j], v[j].z, rr.z); rr.w = fmaf(w[j], v[j].w, rr.w);
                        }
                    }
                    acc += rr;
                }
            } else {
                const long off = ((long)row * n_layers + l) * kv4 + cc;
                for (int k = 0; k < splitk; ++k) acc += P4[off + k * split_stride4];
            }
        }
        (isB ? Bn : KVn)[cc] = acc;
    }
}

hipError_t launch_update(const OperatorView& op, int N, int d, int dm, int n_layers, int S, const int32_t* idx,
                         int idx_layer_stride, const float* R, const float* Pnew, int splitk,
                         long split_stride, const float* B_prev, const float* KV_prev, float* B_next,
                         float* KV_next, hipStream_t stream, const float* kbar, const int32_t* tab) {
    // a thread per float4 of the [B | K' | V'] row (576 at the headline shape: one pass of table read -> gathers -> new-row adds
    // instead of three in sequence; the kernel is nothing but those dependent round trips: 13.3 -> 7 us per step)
    const int total4 = d / 4 + 2 * (dm / 4);
    const int nt = total4 >= 1024 ? 1024 : ((total4 + 63) / 64) * 64;
    INFV_LAUNCH(update_kernel, dim3(N, n_layers), dim3(nt), 0, stream, op, N, d / 4, dm / 4, n_layers, S,
                       idx, idx_layer_stride, R, Pnew, splitk, split_stride / 4, B_prev, KV_prev, B_next,
                       KV_next, kbar, (op.slot_tab != nullptr && op.tabw > 0) ? tab : nullptr);
    return hipGetLastError();
}

hipError_t launch_reproject(const float* B, int N, int d, int dm, int n_layers, const ProjPtrs& proj, float* KV,
                            hipStream_t stream) {
    for (int l = 0; l < n_layers; ++l) {
        hipError_t e = launch_gemm(B + (long)l * N * d, N, d, kv_segs(proj, l, 1, dm), 2 * dm, KV + (long)l * N * 2 * dm,
                                   2 * dm, 1, 0, stream);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// ======================================================================================
// 6. attend: one workgroup per (head h, 16-row query tile, layer).
//      S[q][n]  = (q_h[q] . (K'_h[n] + bk_h)) / sqrt(dh)            MFMA 16x16x4 f32, K' from L2
//      alpha, next sticky bin masses                                 row_phase   (ltm_device.h)
//      ctx[q]   = sum_n alpha[q][n] (V'_h[n] + bv_h)                readout_tile (ltm_device.h)
// ======================================================================================
__global__ __launch_bounds__(256) void attend_kernel(const float* __restrict__ q, int Q, int N, int H,
                                                     const float* __restrict__ KV, ProjPtrs proj,
                                                     const float* __restrict__ readout_w, float w_out,
                                                     StickyView sticky, float* __restrict__ ctx,
                                                     float* __restrict__ bin_part, float* __restrict__ scores) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int sstride = N + 2;                          // == 2 mod 32 for N % 32 == 0
    float* Ssm = lds;                                   // [16][N+2]   scores, then alpha
    float* Vsm = lds + ((kQTile * sstride + 3) & ~3);   // [128][80]   V' rows of the current pass (16-B aligned)
    float* Dsm = Vsm + kVRows * kVStride;               // [16][132]   edge densities
    float* Msm = Dsm + kQTile * kDPitch;                // [16][128]   per-row bin masses
    float* cq = Msm + kQTile * kMPitch;                 // [16] q . bk
    float* asum = cq + 16;                              // [16] sum_n alpha

    const int h = blockIdx.x, qt = blockIdx.y, l = blockIdx.z;
    const int QT = gridDim.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int dm = H * kHeadSize;
    const int c = lane & 15, g = lane >> 4;
    const float scale = 1.0f / sqrtf((float)kHeadSize);
    const float* ql = q + (long)l * Q * dm;
    const float* KVl = KV + (long)l * N * 2 * dm;
    const float* bk = proj.bk[l] + h * kHeadSize;
    const float* bv = proj.bv[l] + h * kHeadSize;

    // ---- q fragment (A operand): lane (row c, group g) holds q[row][16g + j], j = 0..15 ----
    float qa[16];
    {
        const int row = qt * kQTile + c;
        if (row < Q) {
            const floatx4* src = reinterpret_cast<const floatx4*>(ql + (long)row * dm + h * kHeadSize + 16 * g);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const floatx4 t = src[v];
                qa[4 * v + 0] = t.x * scale; qa[4 * v + 1] = t.y * scale;
                qa[4 * v + 2] = t.z * scale; qa[4 * v + 3] = t.w * scale;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) qa[j] = 0.f;
        }
    }
    // q . bk per row: lanes of one row are (c, g=0..3); wave 0 computes and shares it.
    if (wave == 0) {
        float part = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) part = fmaf(qa[j], bk[16 * g + j], part);
        part += __shfl_xor(part, 16);
        part += __shfl_xor(part, 32);
        if (g == 0) cq[c] = part;
    }
    __syncthreads();

    // ---- scores: n-tiles of 16 boxes round-robin over the 4 waves; the K' rows of up to four tiles of a wave are
    //      requested together (one round trip instead of four dependent ones) ----
    for (int nt0 = wave; nt0 < N / 16; nt0 += 16) {
        floatx4 kb[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int nt = nt0 + 4 * u;
            if (nt < N / 16) {
                const floatx4* src = reinterpret_cast<const floatx4*>(KVl + (long)(nt * 16 + c) * 2 * dm + h * kHeadSize + 16 * g);
#pragma unroll
                for (int v = 0; v < 4; ++v) kb[u][v] = src[v];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int nt = nt0 + 4 * u;
            if (nt < N / 16) {
                floatx4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[j], kb[u][j >> 2][j & 3], acc, 0, 0, 0);
                // C/D map 16x16: col = lane&15 (box), row = 4*(lane>>4) + reg (query row)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 4 * g + r;
                    const float sv = acc[r] + cq[row];
                    Ssm[row * sstride + nt * 16 + c] = sv;
                    if (scores != nullptr && qt * kQTile + row < Q)
                        scores[(((long)l * H + h) * Q + qt * kQTile + row) * N + nt * 16 + c] = sv;
                }
            }
        }
    }
    __syncthreads();

    row_phase(Ssm, sstride, N, min(kQTile, Q - qt * kQTile), readout_w, w_out, sticky, Dsm, Msm, asum,
              bin_part + (((long)l * H + h) * QT + qt) * kBins);

    const floatx4 acc = readout_tile(Ssm, sstride, N, KVl + dm + h * kHeadSize, 2L * dm, Vsm);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int rr = 4 * g + r;
        const int qrow = qt * kQTile + rr;
        if (qrow < Q) {
            const int col = 16 * wave + c;
            ctx[((long)l * Q + qrow) * dm + h * kHeadSize + col] = acc[r] + asum[rr] * bv[col];
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// attend, N <= 256: same result as attend_kernel, restructured around its three latencies (measured 1.0 + 4.0 + 6.8 +
// 7.6 us for q / scores / row phase / read-out):
//   * the head's whole V' tile (N x 64) is requested at kernel start and parked in LDS after the scores, so its
//     round trip hides behind the score MFMAs and the read-out never waits for memory;
//   * the row phase runs one wave per query row (4 rows per wave in turn, wave reductions, no workgroup barriers).
// ------------------------------------------------------------------------------------------------------
constexpr int kAtPitch = 80;                        // LDS pitch of the V' tile: == 16 mod 32 -> conflict-free column reads

// RT = query rows per workgroup: 16 (one MFMA row tile, four rows per wave in turn: 48 workgroups at the headline shape) or 4
// (one row per wave: 192 workgroups, the row phase -- the longest part, a dependent chain of ~250 instructions per row -- runs
// once instead of four times per wave; the score and read-out MFMAs then carry 12 idle rows, which costs nothing here).
template <int RT>
__global__ __launch_bounds__(256) void attend_small_kernel(const float* __restrict__ q, int Q, int N, int H,
                                                           const float* __restrict__ KV, ProjPtrs proj,
                                                           const float* __restrict__ readout_w, float w_out,
                                                           StickyView sticky, float* __restrict__ ctx,
                                                           float* __restrict__ bin_part, float* __restrict__ scores) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int sstride = N + 2;
    float* Ssm = lds;                                   // [16][N+2]
    float* Vall = lds + ((kQTile * sstride + 3) & ~3);  // [N][80]
    float* Dsm = Vall + N * kAtPitch;                   // [16][132]
    float* Msm = Dsm + kQTile * kDPitch;                // [16][128]
    float* cq = Msm + kQTile * kMPitch;                 // [16]
    float* asum = cq + 16;                              // [16]

    const int h = blockIdx.x, qt = blockIdx.y, l = blockIdx.z;
    const int QT = gridDim.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int dm = H * kHeadSize;
    const int c = lane & 15, g = lane >> 4;
    const float scale = 1.0f / sqrtf((float)kHeadSize);
    const float* ql = q + (long)l * Q * dm;
    const float* KVl = KV + (long)l * N * 2 * dm;
    const float* bk = proj.bk[l] + h * kHeadSize;
    const float* bv = proj.bv[l] + h * kHeadSize;

    // ---- V' tile of this head: N rows x 16 float4, 16 per thread (N <= 256), in flight until after the scores ----
    floatx4 vreg[16];
    {
        const float* Vh = KVl + dm + h * kHeadSize;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int e = tid + 256 * u;
            const int r = e >> 4, c4 = e & 15;
            vreg[u] = (r < N) ? *reinterpret_cast<const floatx4*>(Vh + (long)r * 2 * dm + c4 * 4) : floatx4{0.f, 0.f, 0.f, 0.f};
        }
    }
    float qa[16];
    {
        const int row = qt * RT + c;
        if (c < RT && row < Q) {
            const floatx4* src = reinterpret_cast<const floatx4*>(ql + (long)row * dm + h * kHeadSize + 16 * g);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const floatx4 t = src[v];
                qa[4 * v + 0] = t.x * scale; qa[4 * v + 1] = t.y * scale;
                qa[4 * v + 2] = t.z * scale; qa[4 * v + 3] = t.w * scale;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) qa[j] = 0.f;
        }
    }
    // K' rows of this wave's score tiles (n-tiles wave, wave+4, ...): requested together
    floatx4 kb[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int nt = wave + 4 * u;
        if (nt < N / 16) {
            const floatx4* src = reinterpret_cast<const floatx4*>(KVl + (long)(nt * 16 + c) * 2 * dm + h * kHeadSize + 16 * g);
#pragma unroll
            for (int v = 0; v < 4; ++v) kb[u][v] = src[v];
        }
    }
    if (wave == 0) {
        float part = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) part = fmaf(qa[j], bk[16 * g + j], part);
        part += __shfl_xor(part, 16);
        part += __shfl_xor(part, 32);
        if (g == 0) cq[c] = part;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int nt = wave + 4 * u;
        if (nt < N / 16) {
            floatx4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 16; ++j)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[j], kb[u][j >> 2][j & 3], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * g + r;
                const float sv = acc[r] + cq[row];
                Ssm[row * sstride + nt * 16 + c] = sv;
                if (scores != nullptr && row < RT && qt * RT + row < Q)
                    scores[(((long)l * H + h) * Q + qt * RT + row) * N + nt * 16 + c] = sv;
            }
        }
    }
    // park the V' tile
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int e = tid + 256 * u;
        const int r = e >> 4, c4 = e & 15;
        if (r < N) *reinterpret_cast<floatx4*>(&Vall[r * kAtPitch + c4 * 4]) = vreg[u];
    }
    __syncthreads();
    // ---- row phase: wave w takes query rows (RT / 4) w .. in turn ----
    const int valid = min(RT, Q - qt * RT);
#pragma unroll 1
    for (int r = 0; r < RT / 4; ++r) {
        const int row = (RT / 4) * wave + r;
        row_phase_row(Ssm + row * sstride, N, row < valid, readout_w, w_out, sticky.edge_box, sticky.edge_dx,
                      Dsm + row * kDPitch, Msm + row * kMPitch, asum + row);
    }
    __syncthreads();
    if (tid < kBins - 1) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < RT; ++r) t += Msm[r * kMPitch + tid];
        bin_part[(((long)l * H + h) * QT + qt) * kBins + tid] = t;
    }
    // ---- read-out: acc[r] = sum_n alpha[4g+r][n] * V'[n][16*wave + c], the whole tile is in LDS ----
    floatx4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < N / 4; t += 2) {
        const float a0 = Ssm[c * sstride + 4 * t + g];
        const float b0 = Vall[(4 * t + g) * kAtPitch + 16 * wave + c];
        const float a1 = Ssm[c * sstride + 4 * (t + 1) + g];
        const float b1 = Vall[(4 * (t + 1) + g) * kAtPitch + 16 * wave + c];
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc1, 0, 0, 0);
    }
    const floatx4 acc = acc0 + acc1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int rr = 4 * g + r;
        const int qrow = qt * RT + rr;
        if (rr < RT && qrow < Q) {
            const int col = 16 * wave + c;
            ctx[((long)l * Q + qrow) * dm + h * kHeadSize + col] = acc[r] + asum[rr] * bv[col];
        }
    }
}

size_t attend_small_lds_bytes(int N) {
    const int sstride = N + 2;
    size_t floats = ((kQTile * sstride + 3) & ~3) + (size_t)N * kAtPitch + kQTile * kDPitch + kQTile * kMPitch + 64;
    return floats * sizeof(float);
}

// query rows per attend workgroup: 4 (one row per wave) where the small kernel applies, else one 16-row MFMA tile
static bool attend_small_wanted() {
    static const bool want_small = [] { const char* e = exp_env("INFV_ATTEND_SMALL"); return !e || atoi(e) != 0; }();
    return want_small;
}
static int attend_tile_rows(int N) {
    static const int rt_env = [] { const char* e = exp_env("INFV_ATTEND_RT"); return e ? atoi(e) : 4; }();
    return (attend_small_wanted() && N <= 256 && N % 16 == 0 && rt_env == 4) ? 4 : kQTile;
}
int attend_parts(int Q, int H, int N) { const int rt = attend_tile_rows(N); return H * ((Q + rt - 1) / rt); }

size_t attend_lds_bytes(int N) {
    const int sstride = N + 2;
    size_t floats = ((kQTile * sstride + 3) & ~3) + kVRows * kVStride + kQTile * kDPitch + kQTile * kMPitch + 64;
    return floats * sizeof(float);
}

hipError_t launch_attend(const float* q, int Q, int N, int H, int n_layers, const float* KV, const ProjPtrs& proj,
                         const float* readout_w, float readout_w_out, const StickyView& sticky, float* ctx,
                         float* bin_part, float* scores, hipStream_t stream) {
    const int rt = attend_tile_rows(N);
    const int QT = (Q + rt - 1) / rt;
    const size_t lds = attend_lds_bytes(N);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attend_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    if (attend_small_wanted() && N <= 256 && N % 16 == 0) {
        static bool attr2 = false;
        if (!attr2) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attend_small_kernel<4>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(attend_small_kernel<16>),
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            attr2 = true;
        }
        if (rt == 4)
            INFV_LAUNCH(attend_small_kernel<4>, dim3(H, QT, n_layers), dim3(256), attend_small_lds_bytes(N), stream, q, Q, N, H,
                               KV, proj, readout_w, readout_w_out, sticky, ctx, bin_part, scores);
        else
            INFV_LAUNCH(attend_small_kernel<16>, dim3(H, QT, n_layers), dim3(256), attend_small_lds_bytes(N), stream, q, Q, N, H,
                               KV, proj, readout_w, readout_w_out, sticky, ctx, bin_part, scores);
        return hipGetLastError();
    }
    INFV_LAUNCH(attend_kernel, dim3(H, QT, n_layers), dim3(256), lds, stream, q, Q, N, H, KV, proj,
                       readout_w, readout_w_out, sticky, ctx, bin_part, scores);
    return hipGetLastError();
}

}  // namespace infv

namespace infv {
__global__ void sum_parts_kernel(const float* __restrict__ part, int parts, int pitch, float* __restrict__ out) {
    const int j = threadIdx.x;
    if (j >= pitch) return;
    double acc = 0.0;
    for (int p = 0; p < parts; ++p) acc += (double)part[(long)p * pitch + j];
    out[j] = (float)acc;
}
hipError_t launch_sum_parts(const float* bin_part_layer, int parts, int pitch, float* bin_mass, hipStream_t stream) {
    INFV_LAUNCH(sum_parts_kernel, dim3(1), dim3(256), 0, stream, bin_part_layer, parts, pitch, bin_mass);
    return hipGetLastError();
}
}  // namespace infv
