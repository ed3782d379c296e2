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
    // XCD-aware tile order.  Workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so neighbouring ids
    // share nothing on chip.  Deal the ids of one XCD a contiguous run of the (m fastest, then n, then batch x split)
    // order instead: the tiles that read the same B rows (and, for a contraction with few column tiles, the same A
    // rows) are then resident on ONE XCD at the same time and the operand reaches that L2 once.  Before: 472 MB fetched
    // per chunk of the video Q-former's four contractions against ~130 MB of operands (profiles/r02_qformer_fetch_*).
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    {
        const unsigned nwg = gridDim.x * gridDim.y * gridDim.z;
        const unsigned orig = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const unsigned xcd = orig & 7u, q = nwg >> 3, r = nwg & 7u;            // bijective for any nwg
        const unsigned v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
        bx = (int)(v % gridDim.x);
        by = (int)((v / gridDim.x) % gridDim.y);
        bz = (int)(v / (gridDim.x * gridDim.y));
    }
    const int m0 = bx * 128, n0 = by * 128;
    const int b = bz / g.splitk, s = bz - b * g.splitk;
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

// ------------------------------------------------------------------------------------------------------
// The same contraction with 384 x 256 x 32 tiles: 8 waves as 4 (M) x 2 (N), each a 96 x 128 block of the output (3 x 4
// accumulators of 32 x 32 in AGPRs; one workgroup per CU, two waves per SIMD, 256 registers each).  H*Q = 384 query rows of the
// video Q-former are one tile: a token row reaches the chip once for all of them.
// What sets this kernel's speed is the operand stream from L2 (measured with the probes below: with every CU pulling, a CU
// receives 37-49 GB/s; the 128-column form of this kernel needed 64 KB per 72 MFMAs per SIMD and waited for them), so the tile is
// as large as LDS and the register file allow: 80 KB per 144 MFMAs per SIMD.
//   * operand tiles go global -> LDS directly (`buffer_load_dwordx4 ... lds`): no staging registers, no ds_write
//     instructions, and the loads of k-tile t + 1 fly while the MFMAs of tile t run (two stages of 80 KB: all of a CU's LDS; ONE
//     barrier per k-tile).  A load instruction lays a wave's 64 x 16 bytes down contiguously, so a tile row is 64 bytes without
//     padding; bank conflicts are avoided by a swizzle done on the GLOBAL side: the lane that fills 16-byte slot `s` of row
//     `r` fetches k-segment s ^ ((r >> 2) & 3), and a fragment read of segment g looks in slot g ^ ((r >> 2) & 3) -- any 16
//     consecutive rows then cover all 64 banks.
//   * the loads are issued one piece (1 KiB) at a time between groups of MFMAs: issued in one burst by a SIMD's only wave they
//     cost 60-180 cycles each in which the matrix pipe idles (measured on the 128-column form: loop = compute-only loop + 1 400
//     cycles per tile; interleaved: + 400); here the SIMD's other wave also has MFMAs to issue meanwhile.
// Round 2's form (384 x 128 x 64 tiles staged through registers, two barriers per tile) sat at 38 % MFMA-busy; the 384 x 128 x 32
// LDS-DMA form of this round (four waves) at 49 % (profiles/r04_*qformer*).
// ------------------------------------------------------------------------------------------------------
namespace {
constexpr int kWI = 3, kWJ = 4;                   // 32 x 32 accumulators per wave: kWI along M, kWJ along N
constexpr int kWRowsA = 4 * 32 * kWI;             // 384 rows of A per workgroup (4 waves along M)
constexpr int kWRowsB = 2 * 32 * kWJ;             // 256 rows of B (2 waves along N)
constexpr int kDBK = 32;                          // k per tile (two 32x32x16 steps)
constexpr int kDRow = 2 * kDBK;                   // 64 bytes per tile row
constexpr int kDArrA = kWRowsA * kDRow, kDArrB = kWRowsB * kDRow;
constexpr int kDStageA = 2 * kDArrA, kDStageB = 2 * kDArrB;     // hi + lo planes of one k-tile: 48 KB of A, 32 KB of B
constexpr int kDLds = 2 * (kDStageA + kDStageB);  // two stages: 163 840 B, all of a CU's LDS
constexpr int kPiecesA = 6, kPiecesB = 4;         // 1-KiB load pieces per wave and k-tile
}  // namespace

// (PROBE: experiments build only, INFV_WIDE_MODE -- 1 = no operand loads, 2 = loads only: timing probes with wrong results)
template <int PROBE>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void split_gemm_wide_kernel(SplitGemm g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    {   // XCD-aware tile order (see split_gemm_kernel)
        const unsigned nwg = gridDim.x * gridDim.y * gridDim.z;
        const unsigned orig = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const unsigned xcd = orig & 7u, q = nwg >> 3, r = nwg & 7u;
        const unsigned v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
        bx = (int)(v % gridDim.x);
        by = (int)((v / gridDim.x) % gridDim.y);
        bz = (int)(v / (gridDim.x * gridDim.y));
    }
    const int m0 = bx * kWRowsA, n0 = by * kWRowsB;
    const int b = bz / g.splitk, s = bz - b * g.splitk;
    const int kbeg = s * g.k_per_split;
    const int kend = (kbeg + g.k_per_split > g.K) ? g.K : kbeg + g.k_per_split;
    const int ntiles = kend > kbeg ? (kend - kbeg) / kDBK : 0;
    float* C = g.C + (long)b * g.strideC + (long)s * g.split_stride;

    // (M % 384 == 0, N % 256 == 0, K % 32 == 0: checked by the launcher -- every load is a whole in-range 16 bytes)
    const long a0 = (long)b * g.strideA + (long)m0 * g.lda + kbeg, b0 = (long)b * g.strideB + (long)n0 * g.ldb + kbeg;
    const int a_bytes = (int)((kWRowsA - 1) * g.lda + (kend - kbeg)) * 2, b_bytes = (int)((kWRowsB - 1) * g.ldb + (kend - kbeg)) * 2;
    // The LDS-DMA loads are written in assembly: through the builtin the compiler knows that they write LDS, cannot tell the
    // stage being filled from the stage being read, and drains vmcnt(0) in front of the first fragment read behind every load.
    // (M0 = LDS byte address of the wave's 1-KiB piece; one wait state between the M0 write and the load.)
    typedef int v4i __attribute__((ext_vector_type(4)));
    auto make_rsrc = [](const __bf16* p, int bytes) {
        const unsigned long a = reinterpret_cast<unsigned long>(p);
        v4i r = {(int)(unsigned)a, (int)(unsigned)((a >> 32) & 0xffffu), bytes, 0x00020000};
        return r;
    };
    const v4i rah = make_rsrc(g.A_hi + a0, a_bytes), ral = make_rsrc(g.A_lo + a0, a_bytes);
    const v4i rbh = make_rsrc(g.B_hi + b0, b_bytes), rbl = make_rsrc(g.B_lo + b0, b_bytes);
    auto dma16 = [](const v4i& rsrc, unsigned lds_addr, int voff, int soff) {
        // (readfirstlane: the "s" constraint alone does not move a value the compiler holds in a vector register)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                     :: "s"(__builtin_amdgcn_readfirstlane(lds_addr)), "v"(voff), "s"(rsrc), "s"(__builtin_amdgcn_readfirstlane(soff)) : "memory");
    };
    const unsigned lds0 = (unsigned)(unsigned long)(lds_ptr)smem;
    // one load instruction = 16 tile rows: lane -> (row lane >> 2, slot lane & 3), fetching k-segment slot ^ swizzle(row)
    const int lrow = lane >> 2, lseg = (lane & 3) ^ ((lane >> 4) & 3);
    const int va = (lrow * (int)g.lda + lseg * 8) * 2, vb = (lrow * (int)g.ldb + lseg * 8) * 2;
    const int blk_a = 16 * (int)g.lda * 2, blk_b = 16 * (int)g.ldb * 2;        // bytes between 16-row blocks
    // piece p of k-tile t: 48 of the 384 rows of A (3 blocks x 2 planes), then 32 of the 256 rows of B (2 blocks x 2 planes)
    auto piece = [&](int t, int p) {                             // p is a constant after unrolling
        const int kt = t * kDRow;
        if (p < kPiecesA) {
            const int blk = wave * 3 + (p >> 1);
            dma16((p & 1) ? ral : rah, lds0 + (t & 1) * kDStageA + (p & 1) * kDArrA + blk * 1024, va, blk * blk_a + kt);
        } else {
            const int blk = wave * 2 + ((p - kPiecesA) >> 1);
            dma16((p & 1) ? rbl : rbh, lds0 + 2 * kDStageA + (t & 1) * kDStageB + (p & 1) * kDArrB + blk * 1024, vb, blk * blk_b + kt);
        }
    };

    floatx16 acc[kWI][kWJ];
#pragma unroll
    for (int i = 0; i < kWI; ++i)
#pragma unroll
        for (int j = 0; j < kWJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int li = lane & 31, kh = lane >> 5;
    const int swz = (li >> 2) & 3;
    const int fo0 = li * kDRow + ((kh ^ swz) << 4), fo1 = li * kDRow + (((2 + kh) ^ swz) << 4);     // k-steps 0 and 1
    const int fa = wm * (32 * kWI) * kDRow, fb = wn * (32 * kWJ) * kDRow;
    if (PROBE != 1 && ntiles > 0) {
#pragma unroll
        for (int p = 0; p < kPiecesA + kPiecesB; ++p) piece(0, p);
    }
#ifdef INFV_EXPERIMENTS
    wg_stamp_begin(g.wg_stamps);
    const long long cyc0 = g.wg_stamps ? (long long)__builtin_amdgcn_s_memtime() : 0;
#endif
    for (int t = 0; t < ntiles; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's part of tile t has landed ...
        __syncthreads();                                         // ... everybody's has, and nobody still reads the other stage
        const unsigned char* base = smem + (t & 1) * kDStageA + fa;
        const unsigned char* base_b = smem + 2 * kDStageA + (t & 1) * kDStageB + fb;
        const bool ld = PROBE != 1 && t + 1 < ntiles;
        if (PROBE == 2) {
            if (ld) {
#pragma unroll
                for (int p = 0; p < kPiecesA + kPiecesB; ++p) piece(t + 1, p);
            }
            continue;
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int fo = ks ? fo1 : fo0;
            bf16x8 al[kWI], ah[kWI], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < kWI; ++i) {
                al[i] = *reinterpret_cast<const bf16x8*>(base + kDArrA + i * 32 * kDRow + fo);
                ah[i] = *reinterpret_cast<const bf16x8*>(base + i * 32 * kDRow + fo);
            }
            bh[0] = *reinterpret_cast<const bf16x8*>(base_b + fo);
            bl[0] = *reinterpret_cast<const bf16x8*>(base_b + kDArrB + fo);
#pragma unroll
            for (int j = 0; j < kWJ; ++j) {
                if (j + 1 < kWJ) {                               // the next 32-column block's fragments, one block ahead
                    bh[(j + 1) & 1] = *reinterpret_cast<const bf16x8*>(base_b + (j + 1) * 32 * kDRow + fo);
                    bl[(j + 1) & 1] = *reinterpret_cast<const bf16x8*>(base_b + kDArrB + (j + 1) * 32 * kDRow + fo);
                }
                // the three products on three accumulators each; behind them one load piece of the next k-tile (8 slots, 10 pieces:
                // the first two slots take two)
                const int slot = ks * kWJ + j;
#pragma unroll
                for (int i = 0; i < kWI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j & 1], acc[i][j], 0, 0, 0);
                if (ld && slot < 2) piece(t + 1, 8 + slot);
#pragma unroll
                for (int i = 0; i < kWI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j & 1], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < kWI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j & 1], acc[i][j], 0, 0, 0);
                if (ld) piece(t + 1, slot);
            }
        }
    }
#ifdef INFV_EXPERIMENTS
    if (g.wg_stamps != nullptr) {                // loop only: start, end (100 MHz) and the shader cycles between them above the CU id
        wg_stamp_end(g.wg_stamps);
        if (threadIdx.x == 0) {
            const long long cyc = (long long)__builtin_amdgcn_s_memtime() - cyc0;
            g.wg_stamps[4 * ((long)blockIdx.x + (long)gridDim.x * (blockIdx.y + (long)gridDim.y * blockIdx.z)) + 2] |= cyc << 36;
        }
    }
#endif
    // (Issuing the products as (B fragment, A fragment) transposes the 32 x 32 blocks and gives every lane four consecutive
    //  columns -- 16-byte stores, a quarter of the instructions -- but a store instruction then writes 32 rows x 32 bytes instead
    //  of 2 rows x 128 bytes: the scores launch of the 128-column form went from 404 to 487 us.  Measured, not kept.)
    // Non-temporal: the output is read by another kernel, the operands should stay in L2.
#pragma unroll
    for (int i = 0; i < kWI; ++i)
#pragma unroll
        for (int j = 0; j < kWJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 32 * kWI + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                const int o = n0 + wn * 32 * kWJ + j * 32 + li;
                __builtin_nontemporal_store(acc[i][j][r], C + (long)m * g.ldc + o);
            }
}

// the wide tiles need whole 384 x 256 tiles and enough of them to fill the chip
bool split_gemm_wide_applies(const SplitGemm& g) {
    static const int mode = [] { const char* e = exp_env("INFV_SPLIT_GEMM_WIDE"); return e ? atoi(e) : 1; }();
    if (!mode) return false;
    if (g.M % kWRowsA || g.N % kWRowsB) return false;               // whole tiles only (unconditional loads and stores)
    const long mt = g.M / kWRowsA;
    const long wgs = mt * ((g.N + kWRowsB - 1) / kWRowsB) * g.nbatch * g.splitk;
    return mode > 1 || wgs >= 192;
}

// tiles of the wide kernel for an [M x N] output per batch entry, 0 if the shape has no whole wide tiles
long split_gemm_wide_tile_count(int M, int N) {
    return (M <= 0 || N <= 0 || M % kWRowsA || N % kWRowsB) ? 0 : (long)(M / kWRowsA) * (N / kWRowsB);
}

// split-K count (and k per split, a multiple of 64) that fills the 256 CUs best with whichever tile shape will run
int split_gemm_pick_splitk(int M, int N, int K, int nbatch, int* k_per_split) {
    const int ktiles = K / kSBK;
    auto best_for = [&](long tiles, int* sk_out) {
        double best = -1.0; int bsk = 1;
        for (int sk = 1; sk <= 16 && sk <= ktiles; ++sk) {
            const int per = (ktiles + sk - 1) / sk;
            if ((long)per * (sk - 1) >= ktiles) continue;             // an empty last split
            const long wgs = tiles * sk;
            const double eff = (double)tiles * ktiles / ((double)((wgs + 255) / 256) * 256.0 * per) - 0.004 * sk;
            if (eff > best) { best = eff; bsk = sk; }
        }
        *sk_out = bsk;
        return best;
    };
    int sk = 1;
    SplitGemm probe{};
    probe.M = M; probe.N = N; probe.K = K; probe.nbatch = nbatch;
    const long tiles_w = (long)((M + kWRowsA - 1) / kWRowsA) * ((N + kWRowsB - 1) / kWRowsB) * nbatch;
    int sk_w = 1;
    best_for(tiles_w, &sk_w);
    probe.splitk = sk_w;
    if (split_gemm_wide_applies(probe)) sk = sk_w;
    else best_for((long)((M + 127) / 128) * ((N + 127) / 128) * nbatch, &sk);
    *k_per_split = ((ktiles + sk - 1) / sk) * kSBK;
    return sk;
}

hipError_t launch_split_gemm(const SplitGemm& g, hipStream_t stream, int lds_pad) {
    if (g.M <= 0 || g.N <= 0 || g.nbatch <= 0) return hipSuccess;
    if (lds_pad <= 0 && split_gemm_wide_applies(g)) {
        if (g.K % kSBK || g.k_per_split % kSBK || g.k_per_split <= 0 || g.lda % 8 || g.ldb % 8 || g.strideA % 8 || g.strideB % 8)
            return hipErrorInvalidValue;
        // (buffer-addressed loads: a tile's rows must lie within 2^31 bytes of its first)
        if ((long)kWRowsA * g.lda * 2 + 2l * g.K >= (1l << 31) || (long)kWRowsB * g.ldb * 2 + 2l * g.K >= (1l << 31)) return hipErrorInvalidValue;
        dim3 grid((g.M + kWRowsA - 1) / kWRowsA, (g.N + kWRowsB - 1) / kWRowsB, g.nbatch * g.splitk);
        SplitGemm gg = g;
        gg.wg_stamps = exp_stamps_reserve(WG_GEMM, (long)grid.x * grid.y * grid.z);
        auto go = [&](auto kernel) {
            static bool attr_w = false;                    // (one per instantiation)
            if (!attr_w) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                if (e != hipSuccess) return e;
                attr_w = true;
            }
            INFV_LAUNCH(kernel, grid, dim3(512), kDLds, stream, gg);
            return hipGetLastError();
        };
#ifdef INFV_EXPERIMENTS
        static const int probe = [] { const char* e = exp_env("INFV_WIDE_MODE"); return e ? atoi(e) : 0; }();
        if (probe == 1) return go(split_gemm_wide_kernel<1>);
        if (probe == 2) return go(split_gemm_wide_kernel<2>);
#endif
        return go(split_gemm_wide_kernel<0>);
    }
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
    INFV_LAUNCH(split_gemm_kernel, grid, dim3(256), kSLds + (lds_pad > 0 ? lds_pad : 0), stream, g);
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
    INFV_LAUNCH(split_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, ld_in, rows, cols / 4,
                       static_cast<__bf16*>(hi), static_cast<__bf16*>(lo), ld_out);
    return hipGetLastError();
}

// ======================================================================================================
// fp32-accurate contraction on the bf16 MFMA pipe ("bf16x6").  An fp32 value is carried EXACTLY as three bf16 pieces
//     x = p0 + p1 + p2,   p0 = bf16(x), p1 = bf16(x - p0), p2 = bf16(x - p0 - p1)        (8 + 8 + 8 = 24 significand bits;
// every difference is exact in fp32, only p2 may lose the last bit or two of x: <= 2^-25 relative), a product of two such
// values as the six partial products with i + j <= 2, each exact in fp32 (8 x 8 bit significands), accumulated in the MFMA's fp32
// accumulators smallest first: a2 b0 + a0 b2 + a1 b1, then a1 b0 + a0 b1, then a0 b0.  The three dropped products are <= 2^-24
// relative together.  The result differs from an fp32 FMA chain the way two fp32 summation orders differ from each other
// (tests/test_ltm_gpu.py measures both against fp64).  Six 32x32x16 bf16 MFMAs (6 x 32 clocks) replace eight 32x32x2 fp32
// MFMAs (8 x 64 clocks) per 32 x 32 x 16 block: 2.7x less matrix-pipe time -- which in the whole-video pipeline is CU time the
// projection GEMM's workgroups hold and during which a co-resident pooling workgroup streams three times slower.
//   C[m][o] = sum_k A[m][k] * B[o][k],  A, B given as three bf16 planes each; 128 x 128 x 32 tiles, 4 waves as 2 x 2, register
//   prefetch of the next k-tile, 60 KB of LDS (a pooling workgroup's 84 KB fit beside it).
// ======================================================================================================
namespace {
constexpr int kXBK = 32;                          // k per tile (two 32x32x16 steps)
constexpr int kXPitch = 2 * kXBK + 16;            // 80 bytes per LDS row
constexpr int kXArr = 128 * kXPitch;              // one operand-plane tile (128 rows)
constexpr int kXLds = 6 * kXArr;                  // A0 A1 A2 B0 B1 B2: 61 440 B
__device__ inline void split3(float x, __bf16& p0, __bf16& p1, __bf16& p2) {
    p0 = (__bf16)x;
    const float r1 = x - (float)p0;
    p1 = (__bf16)r1;
    p2 = (__bf16)(r1 - (float)p1);
}
}  // namespace

__global__ __launch_bounds__(256) void gemm_x6_kernel(SplitGemm6 g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    wg_stamp_begin(g.wg_stamps);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int bx = blockIdx.x, by = blockIdx.y;
    {   // XCD-aware tile order (see split_gemm_kernel): the tiles that share B rows run on one XCD at the same time
        const unsigned nwg = gridDim.x * gridDim.y;
        const unsigned orig = blockIdx.x + gridDim.x * blockIdx.y;
        const unsigned xcd = orig & 7u, q = nwg >> 3, r = nwg & 7u;
        const unsigned v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
        bx = (int)(v % gridDim.x);
        by = (int)(v / gridDim.x);
    }
    const int m0 = bx * 128, n0 = by * 128;
    const int ntiles = g.K / kXBK;
    // staging: thread -> (row = tid >> 1, one half of the row's 32-k segment = 16 bf16 = two 16-byte vectors)
    const int row = tid >> 1, half = tid & 1;
    const bool a_ok = m0 + row < g.M, b_ok = n0 + row < g.N;
    // k-tile-major planes: 16-k tile t of row r at ((t * rows + r) * 16); this kernel's 32-k tile = two of them (the thread's half)
    const long a_off = ((long)half * g.M + (a_ok ? m0 + row : 0)) * 16;
    const long b_off = ((long)half * g.N + (b_ok ? n0 + row : 0)) * 16;
    const long a_t = (long)g.M * 4, b_t = (long)g.N * 4;        // uint4 per 32-k tile: 2 k-tiles x rows x 2
    // Named registers, not arrays (see split_gemm_wide_kernel: hipcc leaves a staging array in scratch memory)
    const uint4* sa0 = reinterpret_cast<const uint4*>(g.A[0] + a_off); const uint4* sa1 = reinterpret_cast<const uint4*>(g.A[1] + a_off);
    const uint4* sa2 = reinterpret_cast<const uint4*>(g.A[2] + a_off); const uint4* sb0 = reinterpret_cast<const uint4*>(g.B[0] + b_off);
    const uint4* sb1 = reinterpret_cast<const uint4*>(g.B[1] + b_off); const uint4* sb2 = reinterpret_cast<const uint4*>(g.B[2] + b_off);
    uint4 ra0x, ra0y, ra1x, ra1y, ra2x, ra2y, rb0x, rb0y, rb1x, rb1y, rb2x, rb2y;
#define INFV_X6_LOAD(t) { const long oa_ = (t) * a_t, ob_ = (t) * b_t;                                                  \
        ra0x = sa0[oa_]; ra0y = sa0[oa_ + 1]; ra1x = sa1[oa_]; ra1y = sa1[oa_ + 1]; ra2x = sa2[oa_]; ra2y = sa2[oa_ + 1];    \
        rb0x = sb0[ob_]; rb0y = sb0[ob_ + 1]; rb1x = sb1[ob_]; rb1y = sb1[ob_ + 1]; rb2x = sb2[ob_]; rb2y = sb2[ob_ + 1]; }
    unsigned char* st_ = smem + row * kXPitch + half * 32;
#define INFV_X6_ST(a, x, y) *reinterpret_cast<uint4*>(st_ + (a) * kXArr) = x; *reinterpret_cast<uint4*>(st_ + (a) * kXArr + 16) = y;
#define INFV_X6_STORE() { INFV_X6_ST(0, ra0x, ra0y) INFV_X6_ST(1, ra1x, ra1y) INFV_X6_ST(2, ra2x, ra2y)               \
                          INFV_X6_ST(3, rb0x, rb0y) INFV_X6_ST(4, rb1x, rb1y) INFV_X6_ST(5, rb2x, rb2y) }
    floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int li = lane & 31, kh = lane >> 5;
    INFV_X6_LOAD(0)
    for (int t = 0; t < ntiles; ++t) {
        INFV_X6_STORE()
        __syncthreads();
        INFV_X6_LOAD(t + 1 < ntiles ? t + 1 : t)       // (the last iteration re-reads its own tile: no branch around the loads)
#pragma unroll
        for (int ks = 0; ks < kXBK / 16; ++ks) {
            bf16x8 a[2][3], b[2][3];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    a[i][pl] = *reinterpret_cast<const bf16x8*>(smem + pl * kXArr + (wm * 64 + i * 32 + li) * kXPitch + ks * 32 + kh * 16);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    b[j][pl] = *reinterpret_cast<const bf16x8*>(smem + (3 + pl) * kXArr + (wn * 64 + j * 32 + li) * kXPitch + ks * 32 + kh * 16);
            // six sweeps over the four accumulators, smallest partial product first: consecutive MFMAs never share an accumulator
#define INFV_X6_SWEEP(pa, pb)                                                                                         \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)              \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][pa], b[j][pb], acc[i][j], 0, 0, 0);
            INFV_X6_SWEEP(2, 0) INFV_X6_SWEEP(0, 2) INFV_X6_SWEEP(1, 1) INFV_X6_SWEEP(1, 0) INFV_X6_SWEEP(0, 1) INFV_X6_SWEEP(0, 0)
#undef INFV_X6_SWEEP
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
                if (m < g.M && o < g.N) __builtin_nontemporal_store(acc[i][j][r], &g.C[(long)m * g.ldc + o]);
            }
#undef INFV_X6_LOAD
#undef INFV_X6_ST
#undef INFV_X6_STORE
    wg_stamp_end(g.wg_stamps);
}

// ------------------------------------------------------------------------------------------------------
// The same six-product contraction with 384 x 256 x 16 tiles, built like split_gemm_wide_kernel: 8 waves as 4 (M) x 2 (N), each a
// 96 x 128 block of the output (12 accumulators), two waves per SIMD, operand planes streamed global -> LDS by `buffer_load ... lds`
// one 1-KiB piece at a time between the MFMAs (a piece = 32 consecutive rows of one k-tile of one plane: 1 KiB of consecutive
// memory in the k-tile-major planes), two stages of 60 KB, one barrier per k-tile.  120 KB of LDS: the workgroup has its CU to
// itself -- in the consolidation pipeline no pooling workgroup sits beside it, whose streaming loads made every vector-memory
// instruction of the 128 x 128 kernel above cost 0.2 us at issue (and whose own loads completed three times slower there), and the
// tile needs 60 KB of operands per 144 MFMAs per SIMD where the 128 x 128 tile needs 48 KB per 48.
// Same products in the same order per accumulator as gemm_x6_kernel: the same bits.
// A tile row is 32 bytes (16 bf16); bank conflicts of the fragment reads are avoided by a swizzle done on the global side: the lane
// that fills 16-byte slot s of row r fetches k-half s ^ ((r >> 3) & 1).
// ------------------------------------------------------------------------------------------------------
namespace {
constexpr int kYRowsA = 384, kYRowsB = 256;
constexpr int kYPlaneA = kYRowsA * 32, kYPlaneB = kYRowsB * 32;          // bytes of one plane's k-tile
constexpr int kYStage = 3 * kYPlaneA + 3 * kYPlaneB;                      // 61 440 B
constexpr int kYLds = 2 * kYStage;
constexpr int kYPiecesA = 3 * (kYRowsA / 32), kYPieces = kYPiecesA + 3 * (kYRowsB / 32);   // 36 + 24 pieces of 1 KiB per k-tile
}  // namespace

// one 384 x 256 output tile at (m0, n0); WRITE_THROUGH: the stores leave this XCD's L2 before the function returns (the call-long
// kernel's consumers are other kernels that run while it is still resident)
template <bool WRITE_THROUGH>
__device__ __forceinline__ void x6_wide_tile(const SplitGemm6& g, unsigned char* smem, int m0, int n0) {
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;
    const int ntiles = g.K / 16;
    typedef int v4i __attribute__((ext_vector_type(4)));
    auto make_rsrc = [](const __bf16* p, int bytes) {
        const unsigned long a = reinterpret_cast<unsigned long>(p);
        v4i r = {(int)(unsigned)a, (int)(unsigned)((a >> 32) & 0xffffu), bytes, 0x00020000};
        return r;
    };
    const int a_bytes = ntiles * g.M * 32, b_bytes = ntiles * g.N * 32;         // (below 2^31: checked by the launcher)
    const v4i ra0 = make_rsrc(g.A[0], a_bytes), ra1 = make_rsrc(g.A[1], a_bytes), ra2 = make_rsrc(g.A[2], a_bytes);
    const v4i rb0 = make_rsrc(g.B[0], b_bytes), rb1 = make_rsrc(g.B[1], b_bytes), rb2 = make_rsrc(g.B[2], b_bytes);
    auto dma16 = [](const v4i& rsrc_, unsigned lds_addr, int voff, int soff) {   // (assembly: see split_gemm_wide_kernel)
        v4i rsrc = rsrc_;
        if (WRITE_THROUGH) {
            // (in the call-long kernel's loop the allocator parks these uniform values in vector registers and hands them to the "s"
            //  operand as they are: pull them back component by component)
            rsrc.x = __builtin_amdgcn_readfirstlane(rsrc_.x); rsrc.y = __builtin_amdgcn_readfirstlane(rsrc_.y);
            rsrc.z = __builtin_amdgcn_readfirstlane(rsrc_.z); rsrc.w = __builtin_amdgcn_readfirstlane(rsrc_.w);
        }
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                     :: "s"(__builtin_amdgcn_readfirstlane(lds_addr)), "v"(voff), "s"(rsrc), "s"(__builtin_amdgcn_readfirstlane(soff)) : "memory");
    };
    const unsigned lds0 = (unsigned)(unsigned long)(lds_ptr)smem;
    // one load instruction = 32 tile rows: lane -> (row lane >> 1, slot lane & 1), fetching k-half slot ^ ((row >> 3) & 1)
    const int lv = (lane >> 1) * 32 + (((lane & 1) ^ ((lane >> 4) & 1)) << 4);
    // Load pieces of k-tile t, by slot (a compile-time constant): slots 0..2 = planes 0..2 of B's 32-row block `wave`; 3..5 = of A's
    // block `wave`; 6..8 = of A's block 8 + wave (waves 0..3: 12 blocks of A on 8 waves; the two waves of a SIMD, w and w + 4, issue 15
    // pieces between them).  M is a multiple of 32 (launcher): a block of A rows is wholly inside the operand or wholly outside
    // (then skipped: its outputs are never stored).
    auto piece = [&](int t, int slot) {
        const unsigned stage = lds0 + (t & 1) * kYStage;
        if (slot < 3) {
            dma16(slot == 0 ? rb0 : slot == 1 ? rb1 : rb2, stage + 3 * kYPlaneA + slot * kYPlaneB + wave * 1024, lv, (t * g.N + n0 + wave * 32) * 32);
        } else {
            const int pl = slot < 6 ? slot - 3 : slot - 6, blk = slot < 6 ? wave : 8 + wave;
            if (blk < 12 && m0 + blk * 32 < g.M)
                dma16(pl == 0 ? ra0 : pl == 1 ? ra1 : ra2, stage + pl * kYPlaneA + blk * 1024, lv, (t * g.M + m0 + blk * 32) * 32);
        }
    };

    floatx16 acc[3][4];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int li = lane & 31, kh = lane >> 5;
    const int fo = li * 32 + ((kh ^ ((li >> 3) & 1)) << 4);
    const int fa = wm * 96 * 32 + fo, fb = 3 * kYPlaneA + wn * 128 * 32 + fo;
    if (ntiles > 0) {
#pragma unroll
        for (int sl = 0; sl < 9; ++sl) piece(0, sl);
    }
    for (int t = 0; t < ntiles; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's pieces of tile t have landed ...
        __syncthreads();                                         // ... everybody's have, and nobody still reads the other stage
        const unsigned char* base = smem + (t & 1) * kYStage;
        const bool ld = t + 1 < ntiles;
        // row block i of A against column blocks 2 jp, 2 jp + 1 of B: the six products, smallest partial product first (the order of
        // gemm_x6_kernel), on two accumulators alternately; between the MFMAs the load pieces of the next tile
        // (6 groups of twelve MFMAs, 9 pieces: one or two per group).  The B fragments are read again for every i (45 fragment reads per k-tile instead of 21): holding
        // all of A's (36 registers) beside the 192 accumulators does not fit two waves per SIMD.
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            bf16x8 a[3], b[2][3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) a[pl] = *reinterpret_cast<const bf16x8*>(base + pl * kYPlaneA + fa + i * 32 * 32);
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) b[jj][pl] = *reinterpret_cast<const bf16x8*>(base + pl * kYPlaneB + fb + (2 * jp + jj) * 32 * 32);
#define INFV_Y_SWEEP(pa, pb) _Pragma("unroll") for (int jj = 0; jj < 2; ++jj) acc[i][2 * jp + jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[pa], b[jj][pb], acc[i][2 * jp + jj], 0, 0, 0);
                INFV_Y_SWEEP(2, 0) INFV_Y_SWEEP(0, 2) INFV_Y_SWEEP(1, 1)
                if (ld) piece(t + 1, (2 * i + jp) * 3 / 2);
                INFV_Y_SWEEP(1, 0) INFV_Y_SWEEP(0, 1) INFV_Y_SWEEP(0, 0)
                if (ld && ((2 * i + jp) & 1)) piece(t + 1, (2 * i + jp) * 3 / 2 + 1);
#undef INFV_Y_SWEEP
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 96 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                const int o = n0 + wn * 128 + j * 32 + li;
                if (m < g.M) __builtin_nontemporal_store(acc[i][j][r], &g.C[(long)m * g.ldc + o]);
            }
    if (WRITE_THROUGH) {
        // every store of the tile has reached this XCD's L2 (vmcnt), then one wave pushes the L2's dirty lines out (release)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (wave == 0) asm volatile("buffer_wbl2 sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
    }
}

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_x6_wide_kernel(SplitGemm6 g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    wg_stamp_begin(g.wg_stamps);
    int bx = blockIdx.x, by = blockIdx.y;
    {   // XCD-aware tile order (see split_gemm_kernel): the tiles that share B rows run on one XCD at the same time
        const unsigned nwg = gridDim.x * gridDim.y;
        const unsigned orig = blockIdx.x + gridDim.x * blockIdx.y;
        const unsigned xcd = orig & 7u, q = nwg >> 3, r = nwg & 7u;
        const unsigned v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
        bx = (int)(v % gridDim.x);
        by = (int)(v / gridDim.x);
    }
    x6_wide_tile<false>(g, smem, bx * kYRowsA, by * kYRowsB);
    wg_stamp_end(g.wg_stamps);
}

#ifdef INFV_EXPERIMENTS
// ------------------------------------------------------------------------------------------------------
// The projection GEMM of a WHOLE consolidate call as ONE resident launch (round 5): a few workgroups per XCD (the same number on
// every XCD: the static block -> XCD deal of every other launch then sees eight equal XCDs) claim 384 x 256 tiles from a
// queue -- sub-batch by sub-batch, a sub-batch's S' column tiles (what role S waits for) before its V' tiles -- wait for the
// pooling launch to have written the sub-batch's rows (its completion count) and for the UC kernel that last read the output
// set, run the tile, push it out of the L2 and count it into the sub-batch's S' / V' word.  Per-sub-batch launches of
// gemm_x6_wide_kernel each needed an EMPTY CU 63 times per sub-batch (~100 us of empty CU-time per workgroup, residency stamps).
// ------------------------------------------------------------------------------------------------------
__global__ void gemm_call_desc_kernel(GemmCallDesc* dst, GemmCallDesc v) { if (threadIdx.x == 0) *dst = v; }

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_x6_call_kernel(const GemmCallDesc* __restrict__ dp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ unsigned int s_tile;
    const int tid = threadIdx.x;
    const GemmCallDesc& d = *dp;                                               // (device memory: the tile body needs the scalar registers)
    const int n_batches = __builtin_amdgcn_readfirstlane(d.n_batches), sub_rows = __builtin_amdgcn_readfirstlane(d.sub_rows);
    const int n_sets = __builtin_amdgcn_readfirstlane(d.n_sets), s_col_tile0 = __builtin_amdgcn_readfirstlane(d.s_col_tile0);
    const int ct_n = __builtin_amdgcn_readfirstlane(d.N) / kYRowsB;            // column tiles of the output
    const int rt_full = (sub_rows + kYRowsA - 1) / kYRowsA;
    const int tiles_full = rt_full * ct_n;
    const int rows_last = __builtin_amdgcn_readfirstlane((int)(d.total_rows - (long)(n_batches - 1) * sub_rows));
    const int rt_last = (rows_last + kYRowsA - 1) / kYRowsA;
    const unsigned int total = (unsigned int)((n_batches - 1) * tiles_full + rt_last * ct_n);
    bool failed = false;
    for (;;) {
        __syncthreads();                                                       // (s_tile of the previous round has been read)
        if (tid == 0) s_tile = atomicAdd(d.tile_ctr, 1u);
        __syncthreads();
        const unsigned int t = __builtin_amdgcn_readfirstlane(s_tile);     // (uniform: the operand resources below live in scalar registers)
        if (t >= total) break;
        int b = (int)(t / (unsigned)tiles_full);
        if (b > n_batches - 1) b = n_batches - 1;
        const int lt = (int)(t - (unsigned)b * tiles_full);
        const int Mb = (b == n_batches - 1) ? rows_last : sub_rows;
        const int rt_b = (b == n_batches - 1) ? rt_last : rt_full;
        const int n_s = (ct_n - s_col_tile0) * rt_b;                          // S' tiles of the sub-batch come first
        int ct, rt; bool is_s;
        if (lt < n_s) { is_s = true; ct = s_col_tile0 + lt / rt_b; rt = lt % rt_b; }
        else { is_s = false; const int lv = lt - n_s; ct = lv / rt_b; rt = lv % rt_b; }
        if (tid == 0 && !failed) {
            // the sub-batch's rows are written (pooling launch) and the output set is free (the UC kernel of sub-batch b - n_sets is done)
            int spins = 0;
            while (__hip_atomic_load(d.pool_done + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned int)Mb ||
                   (b >= n_sets && __hip_atomic_load(d.uc_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned int)(b - n_sets + 1))) {
                __builtin_amdgcn_s_sleep(16);
                if (++spins > d.spin_limit) { __hip_atomic_store(d.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); failed = true; break; }
            }
        }
        __syncthreads();
        // (the descriptor is read through vector loads -- the kernel stores to memory, so the compiler will not use scalar loads --
        //  and the tile body needs its operand resources in scalar registers: make every field uniform by hand)
        auto uni_ptr = [](const void* p) {
            const unsigned long a = reinterpret_cast<unsigned long>(p);
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
            return reinterpret_cast<void*>(((unsigned long)hi << 32) | lo);
        };
        SplitGemm6 g;
        const int Kd = __builtin_amdgcn_readfirstlane(d.K), Nd = __builtin_amdgcn_readfirstlane(d.N);
        const long a_off = (long)b * __builtin_amdgcn_readfirstlane(d.sub_rows) * Kd;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            g.A[i] = static_cast<const __bf16*>(uni_ptr(d.A_all[i])) + a_off;
            g.B[i] = static_cast<const __bf16*>(uni_ptr(d.B[i]));
        }
        g.lda = Kd; g.ldb = Kd; g.C = static_cast<float*>(uni_ptr(d.C_set[b % n_sets]));
        g.ldc = (long)__builtin_amdgcn_readfirstlane((int)d.ldc); g.M = (int)Mb; g.N = Nd; g.K = Kd; g.wg_stamps = nullptr; g.narrow = 0;
        x6_wide_tile<true>(g, smem, rt * kYRowsA, ct * kYRowsB);
        if (tid == 0) __hip_atomic_fetch_add((is_s ? d.done_s : d.done_v) + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

hipError_t launch_gemm_x6_call(const GemmCallDesc& d, GemmCallDesc* d_dev, int n_wgs, hipStream_t stream) {
    if (d.n_batches <= 0 || n_wgs <= 0) return hipSuccess;
    if (d.sub_rows % 32 || d.total_rows % 32 || d.N % kYRowsB || d.K % 16 || (long)(d.K / 16) * d.sub_rows * 32 >= (1l << 31) ||
        (long)(d.K / 16) * d.N * 32 >= (1l << 31) || d.n_sets <= 0 || d.n_sets > 8) return hipErrorInvalidValue;
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_x6_call_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kYLds);   // (+ the static tile word: below the CU's 160 KB)
        if (e != hipSuccess) return e;
        attr = true;
    }
    INFV_LAUNCH(gemm_call_desc_kernel, dim3(1), dim3(64), 0, stream, d_dev, d);
    INFV_LAUNCH(gemm_x6_call_kernel, dim3(n_wgs), dim3(512), kYLds, stream, d_dev);
    return hipGetLastError();
}

#else
hipError_t launch_gemm_x6_call(const GemmCallDesc&, GemmCallDesc*, int, hipStream_t) { return hipErrorNotSupported; }
#endif

bool gemm_x6_wide_applies(const SplitGemm6& g) {
    return g.M > 0 && g.M % 32 == 0 && g.N % kYRowsB == 0 && g.K % 16 == 0 && (long)(g.K / 16) * g.M * 32 < (1l << 31) &&
           (long)(g.K / 16) * g.N * 32 < (1l << 31) && g.M >= 1024;
}

hipError_t launch_gemm_x6(const SplitGemm6& g, hipStream_t stream) {
    if (g.M <= 0 || g.N <= 0) return hipSuccess;
    if (!g.narrow && gemm_x6_wide_applies(g)) {
        static bool attr_y = false;
        if (!attr_y) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_x6_wide_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            attr_y = true;
        }
        dim3 grid((g.M + kYRowsA - 1) / kYRowsA, g.N / kYRowsB);
        SplitGemm6 gg = g;
        gg.wg_stamps = exp_stamps_reserve(WG_GEMM, (long)grid.x * grid.y);
        INFV_LAUNCH(gemm_x6_wide_kernel, grid, dim3(512), kYLds, stream, gg);
        return hipGetLastError();
    }
    if (g.K % kXBK) return hipErrorInvalidValue;
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_x6_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        if (e != hipSuccess) return e;
        attr = true;
    }
    dim3 grid((g.M + 127) / 128, (g.N + 127) / 128);
    SplitGemm6 gg = g;
    gg.wg_stamps = exp_stamps_reserve(WG_GEMM, (long)grid.x * grid.y);
    // (experiment INFV_X6_LDS: total dynamic LDS; above 80 KB only one of these workgroups fits a CU)
    static const size_t lds_x = [] { const char* e = exp_env("INFV_X6_LDS"); return e ? (size_t)atol(e) : (size_t)0; }();
    INFV_LAUNCH(gemm_x6_kernel, grid, dim3(256), lds_x > (size_t)kXLds && lds_x <= 96 * 1024 ? lds_x : (size_t)kXLds, stream, gg);
    return hipGetLastError();
}

// x [rows][cols] fp32 (rows row0 .. row0 + rows - 1 of an operand of rows_total rows) -> the three bf16 planes with x = p0 + p1 + p2,
// K-TILE-MAJOR: element (r, c) of a plane sits at ((c / 16) * rows_total + r) * 16 + c % 16.  A k-tile of 128 consecutive rows is
// then 4 KB of consecutive memory: a wave of the GEMM's staging loads reads 8 whole 128-byte lines instead of 32 bytes of each of 32
// (row-major planes: the kernel spent its time in the CU's memory pipe, 61 us per workgroup for 15 us of MFMAs).
__global__ __launch_bounds__(256) void split3_rows_kernel(const float* __restrict__ x, long ld_in, long rows, int cols4,
                                                          __bf16* __restrict__ p0, __bf16* __restrict__ p1, __bf16* __restrict__ p2,
                                                          long row0, long rows_total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * cols4) return;
    const long r = i / cols4;
    const int c = (int)(i - r * cols4) * 4;
    const floatx4 v = *reinterpret_cast<const floatx4*>(x + r * ld_in + c);
    __bf16 a[4], b[4], d[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) split3(v[e], a[e], b[e], d[e]);
    const long o = ((long)(c >> 4) * rows_total + row0 + r) * 16 + (c & 15);
    *reinterpret_cast<uint2*>(p0 + o) = make_uint2(pack2(a[0], a[1]), pack2(a[2], a[3]));
    *reinterpret_cast<uint2*>(p1 + o) = make_uint2(pack2(b[0], b[1]), pack2(b[2], b[3]));
    *reinterpret_cast<uint2*>(p2 + o) = make_uint2(pack2(d[0], d[1]), pack2(d[2], d[3]));
}

hipError_t launch_split3_rows(const float* x, long ld_in, long rows, int cols, void* p0, void* p1, void* p2, long row0, long rows_total,
                              hipStream_t stream) {
    if (rows <= 0) return hipSuccess;
    if (cols % 16 || ld_in % 4) return hipErrorInvalidValue;
    const long n = rows * (cols / 4);
    INFV_LAUNCH(split3_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, ld_in, rows, cols / 4,
                       static_cast<__bf16*>(p0), static_cast<__bf16*>(p1), static_cast<__bf16*>(p2), row0, rows_total);
    return hipGetLastError();
}

#ifdef INFV_EXPERIMENTS
}  // namespace infv
// test hook (experiments build only, tests/test_ltm_gpu.py): C [M][N] = A [M][K] . B [N][K]^T through the bf16x6 path (which = 0) or
// through the fp32-MFMA kernel the whole-video path used before (which = 1); which = 2: the bf16x6 path's 128 x 128 kernel even where
// the 384 x 256 one applies; which = 3 / 4: the split-bf16 (three-product) contraction, wide tiles / 128 x 128 tiles; device pointers, synchronous
extern "C" int infv_exp_gemm(int which, const float* A, const float* B, float* C, int M, int N, int K) {
    using namespace infv;
    if (which == 1) {
        if (launch_project_scores(M, K, N, B, A, C, N, nullptr, 0) != hipSuccess) return -1;
        return hipDeviceSynchronize() == hipSuccess ? 0 : -1;
    }
    if (which == 3 || which == 4) {
        // split-bf16 (hi/lo, three products) contraction of the video Q-former: 3 = whichever kernel the launcher picks (the
        // 384 x 256 one where whole tiles fit), 4 = the 128 x 128 kernel on the same operands
        __bf16* sp[4] = {};
        for (int i = 0; i < 4; ++i)
            if (hipMalloc(&sp[i], (size_t)(i < 2 ? M : N) * K * sizeof(__bf16)) != hipSuccess) return -1;
        int rc3 = 0;
        if (launch_split_rows(A, K, M, K, sp[0], sp[1], K, nullptr) != hipSuccess) rc3 = -1;
        if (launch_split_rows(B, K, N, K, sp[2], sp[3], K, nullptr) != hipSuccess) rc3 = -1;
        SplitGemm sg{};
        sg.A_hi = sp[0]; sg.A_lo = sp[1]; sg.lda = K; sg.strideA = 0; sg.B_hi = sp[2]; sg.B_lo = sp[3]; sg.ldb = K; sg.strideB = 0;
        sg.C = C; sg.ldc = N; sg.strideC = 0; sg.split_stride = 0; sg.M = M; sg.N = N; sg.K = K; sg.k_per_split = K; sg.splitk = 1; sg.nbatch = 1;
        if (rc3 == 0 && which == 3 && !split_gemm_wide_applies(sg)) rc3 = -2;        // (the test wants to know that the wide kernel ran)
        if (rc3 == 0 && launch_split_gemm(sg, nullptr, which == 4 ? 1 : 0) != hipSuccess) rc3 = -1;
        if (hipDeviceSynchronize() != hipSuccess) rc3 = -1;
        for (int i = 0; i < 4; ++i) (void)hipFree(sp[i]);
        return rc3;
    }
    __bf16* pl[6] = {};
    for (int i = 0; i < 6; ++i)
        if (hipMalloc(&pl[i], (size_t)(i < 3 ? M : N) * K * sizeof(__bf16)) != hipSuccess) return -1;
    int rc = 0;
    if (launch_split3_rows(A, K, M, K, pl[0], pl[1], pl[2], 0, M, nullptr) != hipSuccess) rc = -1;
    if (launch_split3_rows(B, K, N, K, pl[3], pl[4], pl[5], 0, N, nullptr) != hipSuccess) rc = -1;
    SplitGemm6 g{};
    for (int i = 0; i < 3; ++i) { g.A[i] = pl[i]; g.B[i] = pl[3 + i]; }
    g.lda = K; g.ldb = K; g.C = C; g.ldc = N; g.M = M; g.N = N; g.K = K; g.narrow = which == 2;
    if (rc == 0 && launch_gemm_x6(g, nullptr) != hipSuccess) rc = -1;
    if (hipDeviceSynchronize() != hipSuccess) rc = -1;
    for (int i = 0; i < 6; ++i) (void)hipFree(pl[i]);
    return rc;
}
namespace infv {
#endif

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
    INFV_LAUNCH(split_transpose_kernel, dim3((n + 63) / 64, d / 64, nb), dim3(256), 0, stream, F, n, d,
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

// The same softmax for rows of up to 1024 * NV floats, each row read ONCE: a thread keeps its NV float4 in registers
// between the max, the sum and the write (same per-thread order of operations as the three-pass kernel above: same bits).
template <int NV>
__global__ __launch_bounds__(256) void softmax_rows_split_reg_kernel(const float* __restrict__ S, int len, long ld,
                                                                     __bf16* __restrict__ Ph, __bf16* __restrict__ Pl, long ld_out) {
    __shared__ float red[4];
    const float* row = S + (long)blockIdx.x * ld;
    const int tid = threadIdx.x;
    floatx4 v[NV];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = tid * 4 + 1024 * i;
        if (c < len) {
            v[i] = __builtin_nontemporal_load(reinterpret_cast<const floatx4*>(row + c));
            mx = fmaxf(mx, fmaxf(fmaxf(v[i][0], v[i][1]), fmaxf(v[i][2], v[i][3])));
        }
    }
    mx = wave_max(mx);
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = tid * 4 + 1024 * i;
        if (c < len) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[i][k] = expf(v[i][k] - mx);
            sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        }
    }
    sum = wave_sum(sum);
    if ((tid & 63) == 0) red[tid >> 6] = sum;
    __syncthreads();
    const float inv = 1.0f / ((red[0] + red[1]) + (red[2] + red[3]));
    __bf16* ph = Ph + (long)blockIdx.x * ld_out;
    __bf16* pl = Pl + (long)blockIdx.x * ld_out;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = tid * 4 + 1024 * i;
        if (c < len) {
            __bf16 h[4], l[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) split2(v[i][k] * inv, h[k], l[k]);
            *reinterpret_cast<uint2*>(ph + c) = make_uint2(pack2(h[0], h[1]), pack2(h[2], h[3]));
            *reinterpret_cast<uint2*>(pl + c) = make_uint2(pack2(l[0], l[1]), pack2(l[2], l[3]));
        }
    }
}

hipError_t launch_softmax_rows_split(const float* S, long n_rows, int len, long ld, void* Ph, void* Pl, long ld_out,
                                     hipStream_t stream) {
    if (len % 4 || ld % 4 || ld_out % 4) return hipErrorInvalidValue;
    if (len <= 2048) {
        INFV_LAUNCH(softmax_rows_split_reg_kernel<2>, dim3((unsigned)n_rows), dim3(256), 0, stream, S, len, ld,
                           static_cast<__bf16*>(Ph), static_cast<__bf16*>(Pl), ld_out);
        return hipGetLastError();
    }
    if (len <= 8192) {
        INFV_LAUNCH(softmax_rows_split_reg_kernel<8>, dim3((unsigned)n_rows), dim3(256), 0, stream, S, len, ld,
                           static_cast<__bf16*>(Ph), static_cast<__bf16*>(Pl), ld_out);
        return hipGetLastError();
    }
    INFV_LAUNCH(softmax_rows_split_kernel, dim3((unsigned)n_rows), dim3(256), 0, stream, S, len, ld,
                       static_cast<__bf16*>(Ph), static_cast<__bf16*>(Pl), ld_out);
    return hipGetLastError();
}

}  // namespace infv
