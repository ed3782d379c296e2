// State update + read-out of a whole sub-batch of chunks in ONE launch ("UC kernel").
//
// Role S of the chain publishes, per chunk c, the resolved gather table tab_c[n][k] (source box of the
// k-th resampled slot of box n) and the softmax weights alpha_c.  Given those, the memory update
//     X_c[n][:] = val_n * sum_k X_{c-1}[tab_c[n][k]][:] + Xnew_c[row(n)][:]        X in {B, V'}
// acts on every COLUMN of B / V' independently (reference long_term_attention_gibbs.py:210-216:
// B_past^T . samples^T and x @ G are column-wise linear maps).  So a workgroup that owns a 32-column
// slice of one matrix keeps that slice in LDS and walks through all chunks of the sub-batch with no
// inter-workgroup dependency at all; V' slices also produce their 32 columns of the read-out
//     ctx_c[q][cols] = sum_n alpha_c[h][q][n] * V'_c[n][cols] + (sum_n alpha) * bv[cols]        (:284)
// on the MFMA pipe.  The sequential dimension (chunks) costs only LDS latency here, not kernel launches.
#include "ltm_device.h"

#include <cstdlib>

namespace infv {

constexpr int kUcNT = 512;
constexpr int kUcCols = 32;               // columns per slice
constexpr int kUcPitch = kUcCols + 1;     // LDS row pitch of a slice (conflict-free row gathers)
constexpr int kUcMaxN = 256;
constexpr int kUcQ = 32;                  // query rows per read-out pass

// Workgroup barrier that orders LDS traffic only: __syncthreads() also drains the vector-memory counter,
// which would expose the latency of the next chunk's prefetch loads at every barrier.
__device__ inline void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct UcSmem { int cur, nxt, tab, val, brow, newr, alpha, asum, red, total; };

__host__ __device__ inline UcSmem uc_smem(int N, int tabw, int rows_max) {
    UcSmem m;
    int o = 0;
    auto take = [&](int n) { int r = o; o += (n + 3) & ~3; return r; };
    m.cur = take(N * kUcPitch);
    m.nxt = take(N * kUcPitch);
    m.tab = take(N * tabw);
    m.val = take(N);
    m.brow = take(N);
    m.newr = take(rows_max * kUcCols);
    m.alpha = take(kUcQ * (N + 2));
    m.asum = take(kUcQ);
    m.red = take(8 * 64 * 4);
    m.total = o;
    return m;
}

__global__ __launch_bounds__(kUcNT) void uc_kernel(UcArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = a.N, tabw = a.tabw, rows = a.op.rows;
    const UcSmem m = uc_smem(N, tabw, rows);
    float* cur = lds + m.cur;
    float* nxt = lds + m.nxt;
    int32_t* tab = reinterpret_cast<int32_t*>(lds + m.tab);
    float* val = lds + m.val;
    int32_t* brow = reinterpret_cast<int32_t*>(lds + m.brow);
    float* newr = lds + m.newr;
    float* Asm = lds + m.alpha;
    float* asum = lds + m.asum;
    float* red = lds + m.red;
    const int sstride = N + 2;

    // which slice: per layer first the B slices (d / 32), then the V' slices (dm / 32)
    const int sb = a.d / kUcCols, sv = a.dm / kUcCols;
    const int per_layer = sb + sv;
    const int l = blockIdx.x / per_layer;
    const int sl = blockIdx.x - l * per_layer;
    const bool isV = sl >= sb;
    const int col0 = (isV ? sl - sb : sl) * kUcCols;
    const int dm = a.dm, H = a.H, Q = a.Q;
    const int h = col0 / kHeadSize;                      // head of a V' slice
    // global views of this slice
    const int pitch = isV ? 2 * dm : a.d;
    const float* src = isV ? a.KV_prev + (long)l * N * 2 * dm + dm + col0 : a.B_prev + (long)l * N * a.d + col0;
    float* dst = isV ? a.KV_next + (long)l * N * 2 * dm + dm + col0 : a.B_next + (long)l * N * a.d + col0;

    // ---- load the slice and the static operator tables ----
    for (int e = tid; e < N * kUcCols; e += kUcNT) {
        const int n = e / kUcCols, j = e - n * kUcCols;
        cur[n * kUcPitch + j] = a.have_state ? src[(long)n * pitch + j] : 0.f;
    }
    for (int n = tid; n < N; n += kUcNT) { val[n] = a.op.box_val[n]; brow[n] = a.op.box_row[n]; }
    __syncthreads();

    // Per-chunk inputs are prefetched one chunk ahead into registers (all are written by earlier launches,
    // none depends on this kernel), so the sequential loop never waits for a global round trip:
    //   gather table  N*tabw ints  -> int4 per thread (N*tabw <= 2048)
    //   new rows      rows*32      -> up to 16 floats per thread (rows <= 256)
    //   alpha tile    32*N         -> 4 float4 per thread (V' slices; N <= 256, Q-tile of 32 rows)
    const int n4 = N / 4;
    const bool one_qtile = Q <= kUcQ;                    // alpha of the next chunk can be prefetched whole
    int4 t_reg = make_int4(-1, -1, -1, -1);
    float nr_reg[16];
    floatx4 al_reg[4];
    float as_reg = 0.f;
    // offsets of this thread's prefetch elements relative to the chunk's base (computed once)
    int nr_off[16];                                      // -1: not mine
    const long nr_chunk = isV ? (long)rows * a.p_ld : (long)rows * a.d;
    const float* nr_base = isV ? a.Pnew + (long)l * dm + col0 : a.R + col0;
    const int nsplit = isV ? a.splitk : 1;               // only the projected rows come as split-K partial slabs
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int e = tid + u * kUcNT;
        nr_off[u] = -1;
        if (e < rows * kUcCols) {
            const int r = e / kUcCols, j = e - r * kUcCols;
            nr_off[u] = isV ? r * a.p_ld + j : r * a.d + j;
        }
    }
    int al_off[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int e = tid + u * kUcNT;
        const int r = e / n4, c4 = e - r * n4;
        al_off[u] = (r < Q && r < kUcQ) ? r * N + c4 * 4 : -1;
    }
    const long al_slot = (long)a.L * H * Q * N, as_slot = (long)a.L * H * Q;
    const float* al_base = a.alpha + ((long)l * H + h) * (long)Q * N;
    const float* as_base = a.asum + ((long)l * H + h) * (long)Q;
    // ring slots advance by one per chunk (a 64-bit modulo per chunk costs ~0.5 us of scalar division)
    int slot_cur = (int)(a.slot0 % a.ring), slot_pf = slot_cur;
    const int ring = (int)a.ring;
    auto prefetch = [&](int i) {
        const long slot = slot_pf;
        if (++slot_pf == ring) slot_pf = 0;
        if (a.gather && tid * 4 < N * tabw)
            t_reg = *reinterpret_cast<const int4*>(a.tab + slot * a.tab_slot + (long)l * N * tabw + tid * 4);
        const float* nb = nr_base + (long)i * nr_chunk;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            float v = 0.f;
            if (nr_off[u] >= 0) {
                v = nb[nr_off[u]];
                for (int k = 1; k < nsplit; ++k) v += nb[nr_off[u] + k * a.split_stride];
            }
            nr_reg[u] = v;
        }
        if (isV && one_qtile) {
            const float* ab = al_base + slot * al_slot;
#pragma unroll
            for (int u = 0; u < 4; ++u)
                al_reg[u] = (al_off[u] >= 0) ? *reinterpret_cast<const floatx4*>(ab + al_off[u]) : floatx4{0.f, 0.f, 0.f, 0.f};
            if (tid < kUcQ) as_reg = (tid < Q) ? as_base[slot * as_slot + tid] : 0.f;
        }
    };
    const bool stamp_me = a.dbg != nullptr && tid == 0 && blockIdx.x == sb && a.n_chunks > 4;   // first V' slice of layer 0
#define USTAMP(k) do { if (stamp_me && i == 3) a.dbg[k] = wall_clock64(); } while (0)
    prefetch(0);
    for (int i = 0; i < a.n_chunks; ++i) {
        USTAMP(0);
        const long slot = slot_cur;                      // ring slot of this chunk's alpha / tab
        if (++slot_cur == ring) slot_cur = 0;
        // ---- park this chunk's prefetched inputs in LDS, start fetching the next chunk's ----
        if (tid * 4 < N * tabw) *reinterpret_cast<int4*>(&tab[tid * 4]) = a.gather ? t_reg : make_int4(-1, -1, -1, -1);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int e = tid + u * kUcNT;
            if (e < rows * kUcCols) newr[e] = nr_reg[u];
        }
        if (isV && one_qtile) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = tid + u * kUcNT;
                const int r = e / n4, c4 = e - r * n4;
                if (r < kUcQ) {
                    float* d4 = &Asm[r * sstride + c4 * 4];
                    d4[0] = al_reg[u].x; d4[1] = al_reg[u].y; d4[2] = al_reg[u].z; d4[3] = al_reg[u].w;
                }
            }
            if (tid < kUcQ) asum[tid] = as_reg;
        }
        USTAMP(1);
        if (i + 1 < a.n_chunks) prefetch(i + 1);
        USTAMP(2);
        lds_barrier();
        USTAMP(3);
        // ---- memory update of the slice: column j = tid & 31, boxes (tid >> 5) + 16 u; gathers from LDS ----
        {
            const int j = tid & (kUcCols - 1), nb0 = tid >> 5;
            if (tabw == 4) {
                // all table entries first, then all gathered values, then the arithmetic: the LDS latencies overlap
#pragma unroll 1
                for (int u0 = 0; u0 < kUcMaxN / 16; u0 += 4) {
                    int4 s4[4]; float vn[4]; int br[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int n = min(nb0 + 16 * (u0 + u), N - 1);
                        s4[u] = *reinterpret_cast<const int4*>(&tab[n * 4]);
                        vn[u] = val[n]; br[u] = brow[n];
                    }
                    float g[4][4], nw[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        g[u][0] = cur[max(s4[u].x, 0) * kUcPitch + j]; g[u][1] = cur[max(s4[u].y, 0) * kUcPitch + j];
                        g[u][2] = cur[max(s4[u].z, 0) * kUcPitch + j]; g[u][3] = cur[max(s4[u].w, 0) * kUcPitch + j];
                        nw[u] = newr[max(br[u], 0) * kUcCols + j];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int n = nb0 + 16 * (u0 + u);
                        float acc = 0.f;
                        if (s4[u].x >= 0) acc = fmaf(vn[u], g[u][0], acc);
                        if (s4[u].y >= 0) acc = fmaf(vn[u], g[u][1], acc);
                        if (s4[u].z >= 0) acc = fmaf(vn[u], g[u][2], acc);
                        if (s4[u].w >= 0) acc = fmaf(vn[u], g[u][3], acc);
                        if (br[u] >= 0) acc += nw[u];
                        if (n < N) nxt[n * kUcPitch + j] = acc;
                    }
                }
            } else {
                for (int u = 0; u < kUcMaxN / 16; ++u) {
                    const int n = nb0 + 16 * u;
                    if (n < N) {
                        float acc = 0.f;
                        const float vn = val[n];
                        for (int k0 = 0; k0 < tabw; k0 += 4) {
                            const int4 s4 = *reinterpret_cast<const int4*>(&tab[n * tabw + k0]);
                            const float v0 = cur[max(s4.x, 0) * kUcPitch + j], v1 = cur[max(s4.y, 0) * kUcPitch + j];
                            const float v2 = cur[max(s4.z, 0) * kUcPitch + j], v3 = cur[max(s4.w, 0) * kUcPitch + j];
                            if (s4.x >= 0) acc = fmaf(vn, v0, acc);
                            if (s4.y >= 0) acc = fmaf(vn, v1, acc);
                            if (s4.z >= 0) acc = fmaf(vn, v2, acc);
                            if (s4.w >= 0) acc = fmaf(vn, v3, acc);
                        }
                        const int r = brow[n];
                        if (r >= 0) acc += newr[r * kUcCols + j];
                        nxt[n * kUcPitch + j] = acc;
                    }
                }
            }
        }
        lds_barrier();
        USTAMP(4);
        { float* t = cur; cur = nxt; nxt = t; }
        // ---- read-out of the V' slice: passes of 32 query rows ----
        if (isV) {
            const float* alg = a.alpha + ((slot * a.L + l) * H + h) * (long)Q * N;
            const float* asg = a.asum + ((slot * a.L + l) * H + h) * (long)Q;
            float* ctx = a.ctx + (long)i * a.L * Q * dm + (long)l * Q * dm + col0;
            for (int q0 = 0; q0 < Q; q0 += kUcQ) {
                const int qn = min(kUcQ, Q - q0);
                if (!one_qtile) {
                    for (int e = tid; e < kUcQ * N; e += kUcNT) {
                        const int r = e / N, n = e - r * N;
                        Asm[r * sstride + n] = (r < qn) ? alg[(long)(q0 + r) * N + n] : 0.f;
                    }
                    if (tid < kUcQ) asum[tid] = (tid < qn) ? asg[q0 + tid] : 0.f;
                    lds_barrier();
                }
                // 8 waves = 2 query tiles x 2 column tiles x 2 halves of the box dimension
                const int c = lane & 15, g = lane >> 4;
                const int qt = wave & 1, ct = (wave >> 1) & 1, ks = wave >> 2;
                const int per = N / 2, kb = ks * per;
                floatx4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
                for (int t = 0; t < per / 4; t += 2) {
                    const float a0 = Asm[(16 * qt + c) * sstride + kb + 4 * t + g];
                    const float b0 = cur[(kb + 4 * t + g) * kUcPitch + 16 * ct + c];
                    const float a1 = Asm[(16 * qt + c) * sstride + kb + 4 * (t + 1) + g];
                    const float b1 = cur[(kb + 4 * (t + 1) + g) * kUcPitch + 16 * ct + c];
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc1, 0, 0, 0);
                }
                USTAMP(5);
                const floatx4 accw = acc0 + acc1;
                *reinterpret_cast<floatx4*>(&red[(wave * 64 + lane) * 4]) = accw;
                lds_barrier();
                USTAMP(6);
                if (ks == 0) {
                    floatx4 tot = accw;
                    tot += *reinterpret_cast<const floatx4*>(&red[((wave + 4) * 64 + lane) * 4]);
                    const float* bv = a.bv[l] + col0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int rr = 16 * qt + 4 * g + r;
                        if (rr < qn) {
                            const int col = 16 * ct + c;
                            ctx[(long)(q0 + rr) * dm + col] = tot[r] + asum[rr] * bv[col];
                        }
                    }
                }
                if (!one_qtile) lds_barrier();
            }
        }
        // (the next iteration's LDS stores touch tab / newr / Asm / asum / nxt: all of their readers above
        //  sit before the last barrier of this iteration, except the read-out, which only reads Asm, asum,
        //  cur and red -- so one barrier is needed before Asm / asum are overwritten)
        if (isV && one_qtile) lds_barrier();
        USTAMP(7);
    }
    // ---- write the slice back ----
    for (int e = tid; e < N * kUcCols; e += kUcNT) {
        const int n = e / kUcCols, j = e - n * kUcCols;
        dst[(long)n * pitch + j] = cur[n * kUcPitch + j];
    }
}


// ------------------------------------------------------------------------------------------------------
// Fast variant for tabw == 4, Q <= 32, N in {64, 128, 192, 256}: nothing per-chunk is staged through LDS.
//   * update: a thread owns 4 consecutive columns (ds_read/write_b128) of the boxes (tid >> 3) + 64 p; its
//     gather-table rows and new-row values are prefetched from global straight into registers (the box ->
//     new-row map and 1/(count+ridge) are static, kept in registers);
//   * read-out: the alpha tile is fetched with coalesced float4 loads one chunk ahead, parked in LDS with b128
//     stores and read back as the MFMA A operand with b128 loads (lane (c, g) takes the contiguous box range
//     [kb + g*KL, kb + (g+1)*KL) of query row 16*qt + c); loading it per lane straight from global touched 64 cache
//     lines per instruction and cost 2 us per chunk;
//   * the slice lives in LDS with pitch 32 and an XOR swizzle of column bit 4 by row bit 5, so both the
//     b128 row gathers and the strided MFMA B reads are at the 2-way minimum of a 64-lane access.
// LDS: 2 * N * 128 B + 8 KB + 32 * (N + 4) * 4 B (105 KB at N = 256).
// ------------------------------------------------------------------------------------------------------
constexpr int kUfNP = 4;                  // passes of 64 boxes
constexpr int kUfKL = 32;                 // boxes per lane of the read-out (N / 8)

__device__ inline int uf_slot(int r, int j4) { return r * 8 + (j4 ^ (((r >> 5) & 1) << 2)); }   // float4 index

template <bool V16>
__global__ __launch_bounds__(kUcNT) void uc_fast_kernel(UcArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    wg_stamp_begin(a.wg_stamps);
#ifdef INFV_EXPERIMENTS
    if ((a.prio & 15) == 3) __builtin_amdgcn_s_setprio(3);          // (experiment INFV_UC_PRIO: issue priority against co-resident pooling waves)
    else if ((a.prio & 15) == 2) __builtin_amdgcn_s_setprio(2);
    else if ((a.prio & 15) == 1) __builtin_amdgcn_s_setprio(1);
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = a.N, rows = a.op.rows;
    const int apitch = N + 4;

    // a.v16: V' slices are 16 columns wide (B slices stay 32): twice as many read-out workgroups, each with half the
    // MFMA work (the fp32 read-out is matrix-pipe-bound on its CU) and half the update
    const int vcols = V16 ? 16 : kUcCols;               // (template parameter: the default kernel carries none of the narrow-slice selects)
    const int sb = a.d / kUcCols, sv = a.dm / vcols;
    const int per_layer = sb + sv;
    int l = blockIdx.x / per_layer;
    int sl = blockIdx.x - l * per_layer;
    if (V16) {
        // XCD-aware order: the four 16-column V' slices of a head read the SAME alpha tile (32 KB per chunk); workgroups are
        // dealt round-robin over the 8 XCDs, so give the four of them block ids that are equal mod 8 -- the tile then
        // reaches that XCD's L2 once instead of being fetched through four L2s (UC FETCH_SIZE: 160 MB per sub-batch, 129 of
        // them alpha).  V' workgroups first (ids 0 .. L*sv-1), then the B slices.  Speed only.
        const int spl = kHeadSize / 16;                        // slices per head
        const int heads_total = a.L * (sv / spl), nv = a.L * sv;
        if (heads_total % 8 == 0) {
            const int x = blockIdx.x;
            if (x < nv) {
                const int t = x >> 3;
                const int hh = (t / spl) * 8 + (x & 7), s_ = t % spl;
                const int hpl = sv / spl;                       // heads per layer
                l = hh / hpl;
                sl = sb + (hh - l * hpl) * spl + s_;
            } else {
                const int y = x - nv;
                l = y / sb;
                sl = y - l * sb;
            }
        }
    }
    const bool isV = sl >= sb;
    const bool narrow = V16 && isV;                     // this workgroup's slice has 4 float4 per row
    const int col0 = isV ? (sl - sb) * vcols : sl * kUcCols;
    const int dm = a.dm, H = a.H, Q = a.Q;
    const int h = col0 / kHeadSize;
    const int pitch = isV ? 2 * dm : a.d;
    const float* src = isV ? a.KV_prev + (long)l * N * 2 * dm + dm + col0 : a.B_prev + (long)l * N * a.d + col0;
    float* dst = isV ? a.KV_next + (long)l * N * 2 * dm + dm + col0 : a.B_next + (long)l * N * a.d + col0;

    // LDS: two copies of the slice (N x 16 or N x 32 floats), then -- V' slices only -- the k-partials and the alpha tile.
    // With 16-column V' slices a workgroup needs at most 73 KB (64 KB for a B slice): two of them, or one and a workgroup of
    // the projection GEMM, share a CU.
    floatx4* cur = reinterpret_cast<floatx4*>(lds);
    floatx4* nxt = cur + N * (narrow ? 4 : 8);
    float* red = lds + 2 * N * (narrow ? 16 : 32);
    float* Asm = red + 8 * 64 * 4;                      // alpha tile [32][N + 4] of the current chunk
    const int sh = narrow ? 2 : 3;                      // log2(float4 per row)
    const int bi = tid >> sh, c4 = tid & ((1 << sh) - 1);
    const int rpp = kUcNT >> sh;                        // boxes per pass (64 or 128)
    const int NP = (N + rpp - 1) / rpp;
    // LDS slot (float4 index) of (row r, float4 column j4): 32-column slices are swizzled (column bit 4 by row bit 5, see
    // above); 16-column slices are plain -- their read-out takes consecutive rows on consecutive lane groups
    auto slot = [&](int r, int j4) { return narrow ? r * 4 + j4 : uf_slot(r, j4); };
    // ---- load the slice; static operator entries of this thread's boxes ----
    float val[kUfNP]; int brow[kUfNP];
#pragma unroll
    for (int p = 0; p < kUfNP; ++p) {
        const int n = bi + rpp * p;
        val[p] = 0.f; brow[p] = -1;
        if (p < NP && n < N) {
            val[p] = a.op.box_val[n]; brow[p] = a.op.box_row[n];
            floatx4 v = {0.f, 0.f, 0.f, 0.f};
            if (a.have_state) v = *reinterpret_cast<const floatx4*>(src + (long)n * pitch + 4 * c4);
            cur[slot(n, c4)] = v;
        }
    }
    // ---- per-chunk prefetch state ----
    int4 tabv[kUfNP];
    floatx4 newv[kUfNP];
    floatx4 al_reg[4];                                  // coalesced tile loads of the NEXT chunk's alpha
    float asv[4] = {0.f, 0.f, 0.f, 0.f};
    const int nsplit = isV ? a.splitk : 1;
    const long nr_chunk = isV ? (long)rows * a.p_ld : (long)rows * a.d;
    const long nr_ld = isV ? a.p_ld : a.d;
    const float* nr_base = (isV ? a.Pnew + (long)l * dm : a.R) + col0 + 4 * c4;
    const int ring = a.ring;
    int slot_cur = (int)(a.slot0 % a.ring), slot_t = slot_cur, slot_a = slot_cur;
    auto prefetch_tab_new = [&](int i) {
        const int32_t* tb = a.tab + (long)slot_t * a.tab_slot + (long)l * N * 4;
        if (++slot_t == ring) slot_t = 0;
        const float* nb = nr_base + (long)i * nr_chunk;
#pragma unroll
        for (int p = 0; p < kUfNP; ++p) {
            tabv[p] = make_int4(-1, -1, -1, -1);
            newv[p] = floatx4{0.f, 0.f, 0.f, 0.f};
            const int n = bi + rpp * p;
            if (p < NP && n < N) {
                if (a.gather) tabv[p] = *reinterpret_cast<const int4*>(tb + n * 4);
                if (brow[p] >= 0) {
                    floatx4 v = *reinterpret_cast<const floatx4*>(nb + (long)brow[p] * nr_ld);
                    for (int k = 1; k < nsplit; ++k) v += *reinterpret_cast<const floatx4*>(nb + (long)brow[p] * nr_ld + k * a.split_stride);
                    newv[p] = v;
                }
            }
        }
    };
    // read-out geometry.  32-column slices: 8 waves = 2 query tiles x 2 column tiles x 2 halves of the box dimension, lane
    // group g takes a contiguous run of boxes.  16-column slices: 2 query tiles x 4 quarters of the box dimension, and the
    // four k of one MFMA are four CONSECUTIVE boxes (lane group g <-> box 4j + g): consecutive 64-byte rows, no conflicts.
    const int c = lane & 15, g = lane >> 4;
    const int qt = wave & 1, ct = narrow ? 0 : (wave >> 1) & 1, ks = narrow ? wave >> 1 : wave >> 2;
    const int nks = narrow ? 4 : 2;                      // partial sums per output tile
    const int per = N / nks, KL = per / 4;               // boxes per wave, per lane
    const int kb = narrow ? ks * per : ks * per + g * KL;
    const int qrow = 16 * qt + c;
    const long al_slot = (long)a.L * H * Q * N, as_slot = (long)a.L * H * Q;
    const float* al_base = a.alpha + ((long)l * H + h) * (long)Q * N;
    const float* as_base = a.asum + ((long)l * H + h) * Q + 16 * qt + 4 * g;
    const int n4 = N / 4;
    int al_off[4], al_lds[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int e = tid + u * kUcNT;
        const int r = e / n4, cc = e - r * n4;
        al_off[u] = (r < kUcQ && r < Q) ? r * N + cc * 4 : -1;
        al_lds[u] = (r < kUcQ) ? r * apitch + cc * 4 : -1;
    }
    auto prefetch_alpha = [&]() {
        const float* ab = al_base + slot_a * al_slot;
        const float* sbp = as_base + slot_a * as_slot;
        if (++slot_a == ring) slot_a = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            al_reg[u] = (al_off[u] >= 0) ? *reinterpret_cast<const floatx4*>(ab + al_off[u]) : floatx4{0.f, 0.f, 0.f, 0.f};
        if (ks == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) asv[r] = (16 * qt + 4 * g + r < Q) ? sbp[r] : 0.f;
        }
    };
    prefetch_tab_new(0);
    if (isV) prefetch_alpha();
    const float* bvp = a.bv[l] + col0;
    const float bvc = isV ? bvp[16 * ct + c] : 0.f;
    const bool stamp_me = a.dbg != nullptr && tid == 0 && blockIdx.x == sb && a.n_chunks > 4;
    lds_barrier();
    for (int i = 0; i < a.n_chunks; ++i) {
        USTAMP(0);
        float as0 = 0.f, as1 = 0.f, as2 = 0.f, as3 = 0.f;
        if (isV) {
            // park this chunk's alpha tile (readers of the previous tile are past the last barrier) and request the next
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (al_lds[u] >= 0) *reinterpret_cast<floatx4*>(&Asm[al_lds[u]]) = al_reg[u];
            as0 = asv[0]; as1 = asv[1]; as2 = asv[2]; as3 = asv[3];
            if (i + 1 < a.n_chunks && !(a.prio & 32)) prefetch_alpha();      // (prio bit 5: timing experiment, no alpha loads)
        }
        // ---- memory update: X_c[n] = val_n * sum_k X_{c-1}[tab[n][k]] + new row of n ----
#pragma unroll
        for (int p = 0; p < kUfNP; ++p) {
            const int n = bi + rpp * p;
            if (p < NP && n < N) {
                const int4 t = tabv[p];
                const floatx4 g0 = cur[slot(max(t.x, 0), c4)], g1 = cur[slot(max(t.y, 0), c4)];
                const floatx4 g2 = cur[slot(max(t.z, 0), c4)], g3 = cur[slot(max(t.w, 0), c4)];
                floatx4 acc = {0.f, 0.f, 0.f, 0.f};
                const float vn = val[p];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x = 0.f;
                    if (t.x >= 0) x = fmaf(vn, g0[e], x);
                    if (t.y >= 0) x = fmaf(vn, g1[e], x);
                    if (t.z >= 0) x = fmaf(vn, g2[e], x);
                    if (t.w >= 0) x = fmaf(vn, g3[e], x);
                    if (brow[p] >= 0) x += newv[p][e];
                    acc[e] = x;
                }
                nxt[slot(n, c4)] = acc;
            }
        }
        USTAMP(3);
        if (i + 1 < a.n_chunks && !(a.prio & 16)) prefetch_tab_new(i + 1);   // (prio bit 4: timing experiment, no table / new-row loads)
        lds_barrier();
        USTAMP(4);
        { floatx4* t = cur; cur = nxt; nxt = t; }
        if (isV) {
            const float* curf = reinterpret_cast<const float*>(cur);
            floatx4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            if (narrow) {
                // lane (c, g): A = alpha[qrow][kb + 4 j + g], B = V'[kb + 4 j + g][col c]
                const float* arow = Asm + qrow * apitch + kb + g;
                const float* bcol = curf + (long)(kb + g) * 16 + c;
#pragma unroll 8
                for (int j = 0; j < per / 4; j += 2) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[4 * j], bcol[64 * j], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[4 * j + 4], bcol[64 * j + 64], acc1, 0, 0, 0);
                }
            } else {
            const int colj = 16 * ct + c;
            const float* arow = Asm + qrow * apitch + kb;
#pragma unroll
            for (int v = 0; v < kUfKL / 4; ++v) {
                if (4 * v < KL) {
                    const floatx4 av = *reinterpret_cast<const floatx4*>(arow + 4 * v);
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {
                        const int r0 = kb + 4 * v + e, r1 = r0 + 1;
                        const float b0 = curf[r0 * 32 + (colj ^ (((r0 >> 5) & 1) << 4))];
                        const float b1 = curf[r1 * 32 + (colj ^ (((r1 >> 5) & 1) << 4))];
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], b0, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e + 1], b1, acc1, 0, 0, 0);
                    }
                }
            }
            }
            USTAMP(5);
            const floatx4 accw = acc0 + acc1;
            *reinterpret_cast<floatx4*>(&red[(wave * 64 + lane) * 4]) = accw;
            lds_barrier();
            USTAMP(6);
            if (ks == 0) {
                floatx4 tot = accw;
                if (narrow) {
                    tot += *reinterpret_cast<const floatx4*>(&red[((wave + 2) * 64 + lane) * 4]);
                    tot += *reinterpret_cast<const floatx4*>(&red[((wave + 4) * 64 + lane) * 4]);
                    tot += *reinterpret_cast<const floatx4*>(&red[((wave + 6) * 64 + lane) * 4]);
                } else {
                    tot += *reinterpret_cast<const floatx4*>(&red[((wave + 4) * 64 + lane) * 4]);
                }
                float* ctx = a.ctx + (long)i * a.L * Q * dm + (long)l * Q * dm + col0;
                const float asr[4] = {as0, as1, as2, as3};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rr = 16 * qt + 4 * g + r;
                    if (rr < Q) ctx[(long)rr * dm + 16 * ct + c] = tot[r] + asr[r] * bvc;
                }
            }
        }
        USTAMP(7);
    }
    lds_barrier();
    // ---- write the slice back ----
#pragma unroll
    for (int p = 0; p < kUfNP; ++p) {
        const int n = bi + rpp * p;
        if (p < NP && n < N) *reinterpret_cast<floatx4*>(dst + (long)n * pitch + 4 * c4) = cur[slot(n, c4)];
    }
    wg_stamp_end(a.wg_stamps);
}

bool uc_fast_supported(int N, int Q, int tabw) { return tabw == 4 && Q <= kUcQ && N % 64 == 0 && N <= 256; }

size_t uc_lds_bytes(int N, int tabw, int rows_max) { return (size_t)uc_smem(N, tabw, rows_max).total * sizeof(float); }

bool uc_supported(int N, int d, int dm, int tabw, int rows_max) {
    return N <= kUcMaxN && N % 16 == 0 && N * tabw <= 4 * kUcNT && rows_max * kUcCols <= 16 * kUcNT && d % kUcCols == 0 && dm % kHeadSize == 0 && (tabw & 3) == 0 &&
           uc_lds_bytes(N, tabw, rows_max) <= 160 * 1024;
}

hipError_t launch_uc(const UcArgs& a, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(uc_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(uc_fast_kernel<true>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    if (a.n_chunks <= 0) return hipSuccess;
    if (!uc_supported(a.N, a.d, a.dm, a.tabw, a.op.rows)) return hipErrorInvalidValue;
    const int blocks = a.L * (a.d / kUcCols + a.dm / kUcCols);
    static const bool want_fast = [] { const char* e = exp_env("INFV_UC_FAST"); return !e || atoi(e) != 0; }();
    if (want_fast && uc_fast_supported(a.N, a.Q, a.tabw)) {
        // 16-column V' slices (32-column B slices): twice the read-out workgroups of round 1's 32-column form with half the
        // MFMA work each, 73 KB of LDS (the 32-column instantiation measured 10.0 against 7.8 ms per video and is gone;
        // dm is a multiple of the head size 64, so the narrow slices always apply)
        UcArgs b = a;
        b.v16 = 1;
        const int nblk = a.L * (a.d / kUcCols + a.dm / 16);
        b.wg_stamps = exp_stamps_reserve(WG_UC, nblk);
        static const int prio = [] { const char* e = exp_env("INFV_UC_PRIO"); return e ? atoi(e) : 0; }();
        static const int skipm = [] { const char* e = exp_env("INFV_UC_SKIPLOADS"); return e ? atoi(e) : 0; }();   // timing experiments: garbage results
        b.prio = prio | (skipm << 4);
        const size_t vfl = (size_t)2 * a.N * 16 + 8 * 64 * 4 + kUcQ * (a.N + 4), bfl = (size_t)2 * a.N * 32;
        const size_t lds_floats = vfl > bfl ? vfl : bfl;
        INFV_LAUNCH(uc_fast_kernel<true>, dim3(nblk), dim3(kUcNT), lds_floats * sizeof(float), stream, b);
        return hipGetLastError();
    }
    INFV_LAUNCH(uc_kernel, dim3(blocks), dim3(kUcNT), uc_lds_bytes(a.N, a.tabw, a.op.rows), stream, a);
    return hipGetLastError();
}

}  // namespace infv
