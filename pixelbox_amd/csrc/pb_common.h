// pb_common.h -- error plumbing shared by the HIP translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "../../include/pixelbox_hip.h"

namespace pb {

inline char *tls_error() {
    static thread_local char buf[512] = {0};
    return buf;
}

inline int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(tls_error(), 512, fmt, ap);
    va_end(ap);
    return code;
}

#define PB_HIP(expr)                                                                             \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess)                                                                    \
            return pb::fail(PB_ERR_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #expr,             \
                            hipGetErrorString(_e));                                              \
    } while (0)

#define PB_CHECK(cond, code, ...)                          \
    do {                                                   \
        if (!(cond)) return pb::fail(code, __VA_ARGS__);   \
    } while (0)

// Switch to the handle's device for the duration of a call and restore on exit.
struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

}  // namespace pb
