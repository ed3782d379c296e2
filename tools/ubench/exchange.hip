// Micro-benchmarks behind the role-S step (tools/ubench/README.md).  Standalone: hipcc --offload-arch=gfx950 -O3 exchange.hip
//   (1) all-reduce of 127 values over G workgroups, once per "step", 3000 steps, three forms:
//       A  fixed-point u64 atomics into NSH shards + arrival count in the word, sc1 poll (the shipped exchange)
//       B  mailboxes: every workgroup stores its 127 fp32 values as 8-byte {value, step tag} granules (one 16-byte sc1 store per
//          lane) into its own 1-KB slot; every workgroup reads all G slots with 16-byte sc1 loads (wave w: slots w, w+8, ..),
//          re-reads a slot until its tags match, sums in slot order, waves meet in LDS
//   (2) the sequential fp32 running sum of 127 values: systolic DPP scan vs one lane's register chain through LDS vs the same
//       without LDS traffic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef unsigned int uintx4_t __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
constexpr int kBins = 128, kShift = 54, kMaxSh = 8;
__device__ inline int acc_word(int j) { return j < 64 ? 2 * j : 2 * (j - 64) + 1; }

// ---------------------------------------------------------------- A: sharded atomics
template <int NSH>
__global__ __launch_bounds__(512) void xchg_atomic(unsigned long long* acc /*[3][kMaxSh][128]*/, int steps, int work_sleep, long long* out, float* sink) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x, G = gridDim.x;
    const int shard = b % NSH;
    const bool owner = b < NSH;
    unsigned int need[NSH];
#pragma unroll
    for (int r = 0; r < NSH; ++r) need[r] = (G - r + NSH - 1) / NSH;
    uintx4_t pv[NSH];
    double tot = 0.0;
    long long t0 = 0;
    for (int i = 0; i < steps; ++i) {
        if (i == 16 && tid == 0) t0 = wall_clock64();
        unsigned long long* prev = acc + (long)((i + 2) % 3) * kMaxSh * kBins;
        unsigned long long* cur = acc + (long)(i % 3) * kMaxSh * kBins;
        unsigned long long* clr = acc + (long)((i + 1) % 3) * kMaxSh * kBins;
        if (wave == 0 && i > 0) {
            int spins = 0;
            for (;;) {
                bool ok = true;
                unsigned long long m0 = 0, m1 = 0;
#pragma unroll
                for (int r = 0; r < NSH; ++r) {
                    const unsigned long long w0 = ((unsigned long long)pv[r].y << 32) | pv[r].x, w1 = ((unsigned long long)pv[r].w << 32) | pv[r].z;
                    ok = ok && (unsigned)(w0 >> kShift) >= need[r] && (lane == 63 || (unsigned)(w1 >> kShift) >= need[r]);
                    m0 += w0 & ((1ull << kShift) - 1); m1 += w1 & ((1ull << kShift) - 1);
                }
                if (ok) { tot += (double)m0 + (double)m1; break; }
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 22)) { if (lane == 0) printf("timeout b %d step %d\n", b, i); break; }
                __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(prev, 0, kMaxSh * kBins * 8, 0x00020000);
#pragma unroll
                for (int r = 0; r < NSH; ++r) pv[r] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, r * kBins * 8, 16);
            }
        }
        __syncthreads();
        if (owner && tid >= 128 && tid < 256) atomicExch(clr + shard * kBins + (tid - 128), 0ull);
        for (int w_ = 0; w_ < work_sleep; ++w_) __builtin_amdgcn_s_sleep(1);     // stands in for the rest of the step (64 clocks per unit)
        if (owner && (wave == 2 || wave == 3)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid < kBins - 1) atomicAdd(&cur[shard * kBins + acc_word(tid)], (unsigned long long)(tid + 1 + i) + (1ull << kShift));
        if (wave == 0) {
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(cur, 0, kMaxSh * kBins * 8, 0x00020000);
#pragma unroll
            for (int r = 0; r < NSH; ++r) pv[r] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, r * kBins * 8, 16);
        }
    }
    if (tid == 0 && b == 0) out[0] = wall_clock64() - t0;
    if (wave == 0) sink[b * 64 + lane] = (float)tot;
}

// ---------------------------------------------------------------- B: mailboxes (no atomics)
__global__ __launch_bounds__(512) void xchg_mailbox(uintx4_t* box /*[2][G][64] 16-byte {v(bin i), tag, v(bin i+64), tag}*/, int steps, int work_sleep,
                                                    long long* out, float* sink) {
    __shared__ float part[8][kBins];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x, G = gridDim.x;
    float tot = 0.f;
    long long t0 = 0;
    for (int i = 0; i < steps; ++i) {
        if (i == 16 && tid == 0) t0 = wall_clock64();
        // ---- read step i-1's mailboxes: wave w takes slots w, w+8, ...
        if (i > 0) {
            const uintx4_t* bx = box + (long)((i - 1) & 1) * G * 64;
            float s0 = 0.f, s1 = 0.f;
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uintx4_t*>(bx), 0, G * 1024, 0x00020000);
            uintx4_t v[4];
            const unsigned tag = (unsigned)i;       // step i-1 was written with tag i
#pragma unroll
            for (int k = 0; k < 4; ++k) { const int g = wave + 8 * k; if (g < G) v[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, g * 1024, 16); }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int g = wave + 8 * k;
                if (g < G) {
                    int spins = 0;
                    while (__any(v[k].y != tag || v[k].w != tag)) {
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > (1 << 22)) { if (lane == 0) printf("mailbox timeout b %d step %d slot %d\n", b, i, g); break; }
                        v[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, g * 1024, 16);
                    }
                    s0 += __uint_as_float(v[k].x); s1 += __uint_as_float(v[k].z);
                }
            }
            part[wave][lane] = s0; part[wave][lane + 64] = s1;
        }
        __syncthreads();
        if (i > 0 && tid < kBins) { float t = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) t += part[w][tid];
            tot += t; }
        for (int w_ = 0; w_ < work_sleep; ++w_) __builtin_amdgcn_s_sleep(1);
        __syncthreads();
        // ---- publish step i: one 16-byte write-through store per lane of wave 0
        if (wave == 0) {
            uintx4_t w;
            w.x = __float_as_uint((float)(lane + 1 + i)); w.y = (unsigned)(i + 1);
            w.z = __float_as_uint((float)(lane + 65 + i)); w.w = (unsigned)(i + 1);
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(box + (long)(i & 1) * G * 64, 0, G * 1024, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b128(w, rs, lane * 16, b * 1024, 16 /* sc1 */);
        }
    }
    if (tid == 0 && b == 0) out[0] = wall_clock64() - t0;
    if (tid < kBins) sink[b * kBins + tid] = tot;
}

// ---------------------------------------------------------------- C: mailboxes inside ONE XCD's L2
// Grid = 8 * G blocks; only blocks with blockIdx % 8 == pick work (round-robin placement puts them on one XCD: checked
// through HW_REG_XCC_ID, reported).  Producer: PLAIN 16-byte stores (write-through L1, the line stays in the XCD's L2);
// consumer: sc1 loads (bypass L1, served by that L2).  Every step's totals are checked.
__device__ inline int xcc_id() { int v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 15; }

template <bool PLAIN_STORE>
__global__ __launch_bounds__(512) void xchg_mailbox_l2(uintx4_t* box, int G, int pick, int steps, int work_sleep, long long* out, float* sink, int* xcc_out, int* bad) {
    __shared__ float part[8][kBins];
    if ((int)(blockIdx.x & 7) != pick) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x >> 3;
    if (tid == 0) xcc_out[b] = xcc_id();
    float tot = 0.f;
    long long t0 = 0;
    int nbad = 0;
    for (int i = 0; i < steps; ++i) {
        if (i == 16 && tid == 0) t0 = wall_clock64();
        if (i > 0) {
            const uintx4_t* bx = box + (long)((i - 1) & 1) * G * 64;
            float s0 = 0.f, s1 = 0.f;
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uintx4_t*>(bx), 0, G * 1024, 0x00020000);
            uintx4_t v[4];
            const unsigned tag = (unsigned)i;
#pragma unroll
            for (int k = 0; k < 4; ++k) { const int g = wave + 8 * k; if (g < G) v[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, g * 1024, 16); }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int g = wave + 8 * k;
                if (g < G) {
                    int spins = 0;
                    while (__any(v[k].y != tag || v[k].w != tag)) {
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > (1 << 20)) { if (lane == 0) printf("L2 mailbox timeout b %d step %d slot %d\n", b, i, g); break; }
                        v[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, g * 1024, 16);
                    }
                    s0 += __uint_as_float(v[k].x); s1 += __uint_as_float(v[k].z);
                }
            }
            part[wave][lane] = s0; part[wave][lane + 64] = s1;
        }
        __syncthreads();
        if (i > 0 && tid < kBins) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) t += part[w][tid];
            const float expect = (float)G * (float)(tid + 1 + (i - 1));     // every producer wrote lane+1+(i-1) (bin tid)
            if (t != expect) ++nbad;
            tot += t;
        }
        for (int w_ = 0; w_ < work_sleep; ++w_) __builtin_amdgcn_s_sleep(1);
        __syncthreads();
        if (wave == 0) {
            uintx4_t w;
            w.x = __float_as_uint((float)(lane + 1 + i)); w.y = (unsigned)(i + 1);
            w.z = __float_as_uint((float)(lane + 65 + i)); w.w = (unsigned)(i + 1);
            uintx4_t* dst = box + (long)(i & 1) * G * 64 + (long)b * 64 + lane;
            if (PLAIN_STORE) *dst = w;
            else {
                __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(box + (long)(i & 1) * G * 64, 0, G * 1024, 0x00020000);
                __builtin_amdgcn_raw_buffer_store_b128(w, rs, lane * 16, b * 1024, 16 /* sc1 */);
            }
        }
    }
    if (tid == 0 && b == 0) out[0] = wall_clock64() - t0;
    if (tid < kBins) { sink[b * kBins + tid] = tot; if (nbad) atomicAdd(bad, nbad); }
}

// ---------------------------------------------------------------- scans
template <int CTRL> __device__ inline float dpp_f32(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true)); }
__device__ inline float readlane_f32(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

__global__ void scan_bench(const float* p_in, float* out, long long* cyc, int mode, int reps) {
    __shared__ __attribute__((aligned(16))) float praw[kBins];
    __shared__ __attribute__((aligned(16))) float craw[kBins];
    const int lane = threadIdx.x;
    float p0 = p_in[lane], p1 = lane + 64 < 127 ? p_in[lane + 64] : 0.f;
    float r0 = 0.f, r1 = 0.f;
    long long t0 = clock64();
    for (int rep = 0; rep < reps; ++rep) {
        if (mode == 0) {                                   // systolic DPP scan (round 1-3)
            float c0 = p0;
#pragma unroll
            for (int t = 0; t < 63; ++t) c0 = dpp_f32<0x138>(c0) + p0;
            const float carry = readlane_f32(c0, 63);
            const float q1s = (lane == 0) ? carry + p1 : p1;
            float c1 = q1s;
#pragma unroll
            for (int t = 0; t < 62; ++t) c1 = dpp_f32<0x138>(c1) + q1s;
            r0 = c0; r1 = c1;
        } else if (mode == 1) {                            // one lane's register chain through LDS
            praw[lane] = p0; praw[lane + 64] = p1;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (lane == 0) {
                const floatx4* p4 = reinterpret_cast<const floatx4*>(praw);
                floatx4* c4 = reinterpret_cast<floatx4*>(craw);
                float c = 0.f;
                floatx4 pg[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) pg[e] = p4[e];
#pragma unroll
                for (int g = 0; g < 32; ++g) {
                    const floatx4 p = pg[g & 7];
                    floatx4 o;
                    o.x = c + p.x; o.y = o.x + p.y; o.z = o.y + p.z; o.w = o.z + p.w;
                    c = o.w;
                    c4[g] = o;
                    if (g + 8 < 32) pg[g & 7] = p4[g + 8];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            r0 = craw[lane]; r1 = craw[lane + 64];
        } else if (mode == 2) {                            // the bare dependent chain: 127 v_add on one register (no LDS)
            float c = p0;
#pragma unroll
            for (int t = 0; t < 127; ++t) { c = c + p1; asm volatile("" : "+v"(c)); }
            r0 = c; r1 = c;
        } else if (mode == 3) {                            // chain with SGPR operands after 127 readlanes
            float c = 0.f;
#pragma unroll
            for (int t = 0; t < 64; ++t) { c = c + readlane_f32(p0, t); asm volatile("" : "+v"(c)); }
#pragma unroll
            for (int t = 0; t < 63; ++t) { c = c + readlane_f32(p1, t); asm volatile("" : "+v"(c)); }
            r0 = c; r1 = c;
        } else if (mode == 5) {                            // LDS atomic: 64 lanes add to ONE word, each gets the sum before its own add
            if (lane == 0) craw[0] = 0.f;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const float b0 = __hip_atomic_fetch_add(&craw[0], p0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const float b1 = __hip_atomic_fetch_add(&craw[0], p1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            r0 = b0 + p0; r1 = b1 + p1;
        } else if (mode == 4) {                            // DPP row_shr scan (16-lane rows) -- hazard cost of the plain DPP form
            float c0 = p0;
#pragma unroll
            for (int t = 0; t < 63; ++t) c0 = dpp_f32<0x111>(c0) + p0;
            r0 = c0; r1 = c0;
        }
        p0 += r0 * 1e-30f; p1 += r1 * 1e-30f;              // keep the repetitions dependent
    }
    long long t1 = clock64();
    out[lane] = r0; out[lane + 64] = r1;
    if (lane == 0) cyc[0] = t1 - t0;
}

int main(int argc, char** argv) {
    const int steps = 3000;
    unsigned long long* acc; uintx4_t* box; long long* out; float* sink;
    CK(hipMalloc(&acc, 3 * kMaxSh * kBins * 8)); CK(hipMalloc(&box, 2 * 64 * 1024)); CK(hipMalloc(&out, 64)); CK(hipMalloc(&sink, 64 * 128 * 4));
    auto run = [&](const char* name, int G, int nsh, int work, auto launch) {
        std::vector<double> us;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipMemset(acc, 0, 3 * kMaxSh * kBins * 8)); CK(hipMemset(box, 0, 2 * 64 * 1024));
            launch();
            CK(hipDeviceSynchronize());
            long long t; CK(hipMemcpy(&t, out, 8, hipMemcpyDeviceToHost));
            us.push_back((double)t * 0.01 / (steps - 16));
        }
        std::sort(us.begin(), us.end());
        printf("%-10s G=%2d shards=%d work=%4d clk : %.3f us/step (min %.3f max %.3f)\n", name, G, nsh, work * 64, us[2], us[0], us[4]);
    };
    for (int work : {0, 32}) {
        for (int G : {24, 48}) {
            run("atomic", G, 1, work, [&] { hipLaunchKernelGGL(xchg_atomic<1>, dim3(G), dim3(512), 0, 0, acc, steps, work, out, sink); });
            run("atomic", G, 2, work, [&] { hipLaunchKernelGGL(xchg_atomic<2>, dim3(G), dim3(512), 0, 0, acc, steps, work, out, sink); });
            run("atomic", G, 4, work, [&] { hipLaunchKernelGGL(xchg_atomic<4>, dim3(G), dim3(512), 0, 0, acc, steps, work, out, sink); });
            run("atomic", G, 8, work, [&] { hipLaunchKernelGGL(xchg_atomic<8>, dim3(G), dim3(512), 0, 0, acc, steps, work, out, sink); });
            if (G <= 32) run("mailbox", G, 0, work, [&] { hipLaunchKernelGGL(xchg_mailbox, dim3(G), dim3(512), 0, 0, box, steps, work, out, sink); });
        }
    }
    {
        int* xcc; int* bad;
        CK(hipMalloc(&xcc, 64 * 4)); CK(hipMalloc(&bad, 4));
        for (int plain = 1; plain >= 0; --plain)
            for (int work : {0, 32})
                for (int pick : {0, 4}) {
                    const int G = 24;
                    std::vector<double> us;
                    int hbad = 0, hx[64];
                    for (int rep = 0; rep < 5; ++rep) {
                        CK(hipMemset(box, 0, 2 * 64 * 1024)); CK(hipMemset(bad, 0, 4)); CK(hipMemset(xcc, 0xff, 256));
                        if (plain) hipLaunchKernelGGL(xchg_mailbox_l2<true>, dim3(8 * G), dim3(512), 0, 0, box, G, pick, steps, work, out, sink, xcc, bad);
                        else hipLaunchKernelGGL(xchg_mailbox_l2<false>, dim3(8 * G), dim3(512), 0, 0, box, G, pick, steps, work, out, sink, xcc, bad);
                        CK(hipDeviceSynchronize());
                        long long t; CK(hipMemcpy(&t, out, 8, hipMemcpyDeviceToHost));
                        int bb; CK(hipMemcpy(&bb, bad, 4, hipMemcpyDeviceToHost)); hbad += bb;
                        CK(hipMemcpy(hx, xcc, 256, hipMemcpyDeviceToHost));
                        us.push_back((double)t * 0.01 / (steps - 16));
                    }
                    std::sort(us.begin(), us.end());
                    int mask = 0; for (int g = 0; g < G; ++g) mask |= 1 << hx[g];
                    printf("mailbox-L2 %s G=%d pick=%d work=%4d clk : %.3f us/step (min %.3f max %.3f)  xcc mask 0x%x  wrong totals %d\n",
                           plain ? "plain-store" : "sc1-store  ", G, pick, work * 64, us[2], us[0], us[4], mask, hbad);
                }
    }
    // scans
    float* p; float* o; long long* cyc;
    CK(hipMalloc(&p, 512)); CK(hipMalloc(&o, 512)); CK(hipMalloc(&cyc, 8));
    std::vector<float> hp(128);
    double s = 0; for (int i = 0; i < 127; ++i) { hp[i] = 1.0f / 127 + 1e-4f * (i % 7); s += hp[i]; } hp[127] = 0;
    CK(hipMemcpy(p, hp.data(), 512, hipMemcpyHostToDevice));
    const char* names[] = {"dpp wave_shr systolic", "lane-0 chain via LDS", "bare 127 dependent v_add", "readlane + v_add chain", "dpp row_shr (63 links)", "LDS atomic add, one word"};
    std::vector<float> ref(128);
    for (int mode = 0; mode < 6; ++mode) {
        hipLaunchKernelGGL(scan_bench, dim3(1), dim3(64), 0, 0, p, o, cyc, mode, 1); CK(hipDeviceSynchronize());
        hipLaunchKernelGGL(scan_bench, dim3(1), dim3(64), 0, 0, p, o, cyc, mode, 200); CK(hipDeviceSynchronize());
        long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
        std::vector<float> ho(128); 
        hipLaunchKernelGGL(scan_bench, dim3(1), dim3(64), 0, 0, p, o, cyc, mode, 1); CK(hipDeviceSynchronize());
        CK(hipMemcpy(ho.data(), o, 512, hipMemcpyDeviceToHost));
        if (mode == 0) ref = ho;
        int same = 0; for (int i = 0; i < 127; ++i) same += (ho[i] == ref[i]);
        printf("scan %-26s : %.0f clocks per scan; %d/127 entries equal the systolic scan's\n", names[mode], (double)c / 200, same);
    }
    {   // does the LDS atomic serve the lanes in lane order?  500 random probability vectors against the systolic scan, bit for bit
        int all_same = 0, trials = 500; long worst = 0;
        std::vector<float> a0(128), a5(128);
        srand(7);
        for (int t = 0; t < trials; ++t) {
            double s2 = 0; for (int i = 0; i < 127; ++i) { hp[i] = (float)((rand() % 100000 + 1) * ((t & 1) ? 1e-7 : 1e-5) * ((rand() % 7 == 0) ? 50.0 : 1.0)); s2 += hp[i]; }
            for (int i = 0; i < 127; ++i) hp[i] = (float)(hp[i] / s2);
            CK(hipMemcpy(p, hp.data(), 512, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(scan_bench, dim3(1), dim3(64), 0, 0, p, o, cyc, 0, 1); CK(hipDeviceSynchronize());
            CK(hipMemcpy(a0.data(), o, 512, hipMemcpyDeviceToHost));
            hipLaunchKernelGGL(scan_bench, dim3(1), dim3(64), 0, 0, p, o, cyc, 5, 1); CK(hipDeviceSynchronize());
            CK(hipMemcpy(a5.data(), o, 512, hipMemcpyDeviceToHost));
            int same = 0; for (int i = 0; i < 127; ++i) same += (a0[i] == a5[i]);
            all_same += (same == 127); if (127 - same > worst) worst = 127 - same;
        }
        printf("LDS-atomic scan == systolic scan on %d of %d random vectors (worst: %ld entries differ)\n", all_same, trials, worst);
    }
    return 0;
}
