// common.hip.h — error reporting and device-buffer helpers shared by the C ABI sources.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>

#include "../../include/zett_hip.h"

namespace zett {

inline thread_local std::string g_err;

inline int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                               \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess)                                                                       \
            return ::zett::fail(ZETT_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

// Every ABI entry point works on its handle's device and leaves the calling thread's current device as it found it
// (the caller may be a torch process holding several GPUs; zett_destroy can run from a garbage collector).
struct DeviceScope {
    int prev = -1;
    hipError_t err = hipSuccess;
    explicit DeviceScope(int device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) err = hipSetDevice(device); else prev = -1;
    }
    ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceScope(const DeviceScope&) = delete;
    DeviceScope& operator=(const DeviceScope&) = delete;
};
#define ZETT_ON_DEVICE(dev)                                                                          \
    ::zett::DeviceScope _scope(dev);                                                                 \
    if (_scope.err != hipSuccess) return ::zett::fail(ZETT_E_HIP, "hipSetDevice(%d) failed: %s", (int)(dev), hipGetErrorString(_scope.err))

struct DevBuf {   // grow-only device allocation
    void* p = nullptr;
    size_t bytes = 0;
    int reserve(size_t need) {
        if (need <= bytes) return 0;
        if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
        need = (need + 255) & ~(size_t)255;
        hipError_t e = hipMalloc(&p, need);
        if (e != hipSuccess) return fail(ZETT_E_HIP, "hipMalloc(%zu) failed: %s", need, hipGetErrorString(e));
        bytes = need;
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
    template <typename U> U* as() const { return (U*)p; }
};

}  // namespace zett
