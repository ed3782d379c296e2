// Shared by the C-ABI translation units (ltm_capi.hip, vqf_capi.hip): error reporting and device buffers.
#pragma once
#include "knobs.h"
#include <hip/hip_runtime.h>
#include <stddef.h>

#include "../../include/infv_ltm.h"

namespace infv {

// Records the message returned by infv_ltm_last_error() (thread-local) and returns `code`.
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return ::infv::fail(INFV_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

struct DeviceBuf {
    void* p = nullptr;
    size_t bytes = 0;
    DeviceBuf() = default;
    DeviceBuf(const DeviceBuf&) = delete;
    DeviceBuf& operator=(const DeviceBuf&) = delete;
    ~DeviceBuf() { if (p) (void)hipFree(p); }
    hipError_t reserve(size_t n) {
        if (n <= bytes) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); p = nullptr; bytes = 0; if (e != hipSuccess) return e; }
        hipError_t e = hipMalloc(&p, n);
        if (e == hipSuccess) bytes = n;
        return e;
    }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

template <class T>
hipError_t upload(DeviceBuf& buf, const T* host, size_t n) {
    hipError_t e = buf.reserve((n ? n : 1) * sizeof(T));
    if (e != hipSuccess || n == 0) return e;
    return hipMemcpy(buf.p, host, n * sizeof(T), hipMemcpyHostToDevice);
}

}  // namespace infv
