// Microbenchmark: duration (rocprofv3 kernel-trace) of a kernel whose workgroups return at once,
// as a function of workgroup size, dynamic LDS, grid size and kernarg bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
struct Big { int v[180]; };   // 720 B of kernarg
template <int NT> __global__ __launch_bounds__(NT) void noop_small(int flag, int* out) { if (flag) out[threadIdx.x] = 1; }
template <int NT> __global__ __launch_bounds__(NT) void noop_big(Big b, int* out) { if (b.v[179]) out[threadIdx.x] = 1; }
template <int NT> __global__ __launch_bounds__(NT) void noop_lds(int flag, int* out) { extern __shared__ float l[]; if (flag) out[threadIdx.x] = (int)l[flag]; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    int* out; CK(hipMalloc(&out, 4096));
    Big b{}; 
    CK(hipFuncSetAttribute((const void*)noop_lds<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)noop_lds<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int rep = 0; rep < 200; ++rep) {
        hipLaunchKernelGGL(noop_small<256>, dim3(160), dim3(256), 0, 0, 0, out);
        hipLaunchKernelGGL(noop_small<1024>, dim3(160), dim3(1024), 0, 0, 0, out);
        hipLaunchKernelGGL(noop_big<256>, dim3(160), dim3(256), 0, 0, b, out);
        hipLaunchKernelGGL(noop_big<1024>, dim3(160), dim3(1024), 0, 0, b, out);
        hipLaunchKernelGGL(noop_lds<256>, dim3(160), dim3(256), 100 * 1024, 0, 0, out);
        hipLaunchKernelGGL(noop_lds<1024>, dim3(160), dim3(1024), 100 * 1024, 0, 0, out);
        hipLaunchKernelGGL(noop_lds<1024>, dim3(48), dim3(1024), 100 * 1024, 0, 0, out);
        hipLaunchKernelGGL(noop_lds<1024>, dim3(160), dim3(1024), 16 * 1024, 0, 0, out);
        hipLaunchKernelGGL(noop_small<256>, dim3(1024), dim3(256), 0, 0, 0, out);
    }
    CK(hipDeviceSynchronize());
    printf("done\n");
    return 0;
}
