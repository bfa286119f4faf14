// bd_host.h -- host-side helpers shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include "../../include/bigdreamer_hip.h"

namespace bd {

char* err_buf();   // thread-local message buffer (cabi.hip)

inline int fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return -1;
}

constexpr int kMaxLds = 160 * 1024;   // LDS per CU / per workgroup on gfx950

// Allow a kernel to use more than the default 64 KiB of dynamic LDS (once per kernel symbol).
template <class K>
inline int allow_big_lds(K kernel) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds);
    if (e != hipSuccess) return fail("hipFuncSetAttribute(MaxDynamicSharedMemorySize): %s", hipGetErrorString(e));
    return 0;
}

}  // namespace bd

#define BD_CHECK_LAUNCH(name)                                                             \
    do {                                                                                  \
        hipError_t e_ = hipGetLastError();                                                \
        if (e_ != hipSuccess) return bd::fail("%s: launch failed: %s", name, hipGetErrorString(e_)); \
    } while (0)

#define BD_REQUIRE(cond, ...)                  \
    do {                                       \
        if (!(cond)) return bd::fail(__VA_ARGS__); \
    } while (0)
