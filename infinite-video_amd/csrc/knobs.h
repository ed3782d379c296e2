// Environment knobs of libinfv_ltm.so.
//
// The shipped library reads five documented options from the environment (all parity-tested; see README.md):
//   INFV_PROJ_X6, INFV_VPROJ_SPLIT, INFV_VQF_FP32, INFV_VQF_FUSE, INFV_VQF_SPLIT_CACHE_GB   -> plain getenv at their call sites.
// Everything else -- timing experiments that produce garbage (INFV_SKIP, INFV_S_FLAGS), in-kernel stamps, fault
// injection, occupancy pads, stream priorities and the A/B selectors of variants that are not the default -- goes through
// exp_env(), which only looks at the environment in the experiments build (-DINFV_EXPERIMENTS:
// libinfv_ltm_exp.so, what tools/ and the variant / fault-injection tests load through INFV_LTM_LIBRARY=exp).  In the
// shipped library exp_env() is a constant nullptr: no hidden work-skipping or tuning switch can change what it runs.
#pragma once
#include <cstdlib>

namespace infv {
inline const char* exp_env(const char* name) {
#ifdef INFV_EXPERIMENTS
    return std::getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}
}  // namespace infv

// ---- launch census: every kernel launch of the library goes through INFV_LAUNCH (same argument order as hipLaunchKernelGGL); it
// counts them on the host (one relaxed atomic add per launch) so that a benchmark can state "launches per chunk" of a path without
// a profiler (infv_ltm_launch_count, include/infv_ltm.h).  tests/test_host_cpu.py checks that no source launches a kernel any other
// way (no raw <<< >>>, no hipLaunchKernelGGL, no hipModuleLaunchKernel).
#include <hip/hip_runtime.h>
#include <atomic>
namespace infv { extern std::atomic<long long> g_kernel_launches; }
#define INFV_LAUNCH(kernelName, numBlocks, numThreads, memPerBlock, streamId, ...)                                \
    do {                                                                                                          \
        ::infv::g_kernel_launches.fetch_add(1, std::memory_order_relaxed);                                        \
        (kernelName)<<<(numBlocks), (numThreads), (memPerBlock), (streamId)>>>(__VA_ARGS__);                      \
    } while (0)
