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

// The HIP current device is per host thread and new threads start on device 0: every entry point that works on an
// object (engine, net) rebinds the calling thread to the device the object lives on; the stateless batch calls bind
// to the device that owns their first device pointer.  One process per GPU is the deployment, but a worker thread
// of rank r must not silently run on GPU 0.
int bind_device(int dev);
int bind_pointer_device(const void* p);
int current_device();
#define OTH_BIND(dev)                    \
    do {                                 \
        int _r = oth::bind_device(dev);  \
        if (_r) return _r;               \
    } while (0)
#define OTH_BIND_PTR(p)                          \
    do {                                         \
        int _r = oth::bind_pointer_device(p);    \
        if (_r) return _r;                       \
    } while (0)

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace oth
