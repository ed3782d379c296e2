// Workgroup residency stamps (experiments build only, INFV_WG_STAMPS=1; tools/residency.py): every workgroup of the pipeline's
// kernels records when it started, when it ended and which CU it ran on.  The shipped library compiles none of this.
#pragma once
#include <hip/hip_runtime.h>

namespace infv {
enum WgKind { WG_POOL = 1, WG_GEMM = 2, WG_UC = 3, WG_CHAIN = 4, WG_ALPHA = 5 };

#ifdef INFV_EXPERIMENTS
// device buffer of n_wgs records [start, end, (xcc << 32) | hw_id, kind], appended per launch; nullptr when stamps are off
long long* exp_stamps_reserve(int kind, long n_wgs);

__device__ inline void wg_stamp_begin(long long* s) {
    if (s != nullptr && threadIdx.x == 0) {
        long long* r = s + 4 * ((long)blockIdx.x + (long)gridDim.x * (blockIdx.y + (long)gridDim.y * blockIdx.z));
        r[0] = wall_clock64();
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        r[2] = ((long long)(xcc & 0xf) << 32) | hw;
    }
}
// call where every wave of the workgroup has arrived (after a __syncthreads(), or from the last wave to finish)
__device__ inline void wg_stamp_end(long long* s) {
    if (s != nullptr && threadIdx.x == 0)
        s[4 * ((long)blockIdx.x + (long)gridDim.x * (blockIdx.y + (long)gridDim.y * blockIdx.z)) + 1] = wall_clock64();
}
#else
inline long long* exp_stamps_reserve(int, long) { return nullptr; }
__device__ inline void wg_stamp_begin(long long*) {}
__device__ inline void wg_stamp_end(long long*) {}
#endif
}  // namespace infv
