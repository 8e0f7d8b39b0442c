// common.h -- error plumbing shared by the translation units of libothello_mi355x.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/othello_mi355x.h"

namespace oth {

void set_error(const char* fmt, ...);
bool device_ok();  // a gfx950 device is present and usable

#define OTH_HIP(expr)                                                                      \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess) {                                                            \
            oth::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                           __LINE__);                                                      \
            return OTH_E_HIP;                                                              \
        }                                                                                  \
    } while (0)

#define OTH_NEED_DEVICE()                                                              \
    do {                                                                               \
        if (!oth::device_ok()) {                                                       \
            oth::set_error("no gfx950 (MI355X) device available: the HIP path cannot " \
                           "run and there is no CPU fallback");                        \
            return OTH_E_NO_DEVICE;                                                    \
        }                                                                              \
    } while (0)

#define OTH_CHECK(cond, ...)             \
    do {                                 \
        if (!(cond)) {                   \
            oth::set_error(__VA_ARGS__); \
            return OTH_E_INVALID;        \
        }                                \
    } while (0)

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace oth
