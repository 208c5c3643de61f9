// Shared host-side helpers of libtvc_hip.so (error reporting, HIP call checking).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "tvc_native.h"

namespace tvc {

char* last_error_buf();  // thread-local, 512 bytes
inline int set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(last_error_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

#define TVC_HIP_CHECK(expr)                                                                          \
    do {                                                                                             \
        hipError_t _e = (expr);                                                                      \
        if (_e != hipSuccess)                                                                        \
            return tvc::set_error(TVC_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),   \
                                  __FILE__, __LINE__);                                               \
    } while (0)

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

}  // namespace tvc
