// Shared host/device helpers for libcim_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>

namespace cim {

void set_error(const char* fmt, ...);

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Python float threshold -> the binary16 value PyTorch compares an fp16 tensor against
// (round-to-nearest-even), widened back to f32 so device code can compare in f32 exactly.
static inline float round_to_f16(float thr) { return (float)(_Float16)thr; }

#define CIM_CHECK_ARG(cond)                                                     \
    do {                                                                        \
        if (!(cond)) {                                                          \
            cim::set_error("%s: bad argument: %s", __func__, #cond);            \
            return -1;                                                          \
        }                                                                       \
    } while (0)

#define CIM_CHECK_LAUNCH()                                                      \
    do {                                                                        \
        hipError_t e_ = hipGetLastError();                                      \
        if (e_ != hipSuccess) {                                                 \
            cim::set_error("%s: %s", __func__, hipGetErrorString(e_));          \
            return (int)e_;                                                     \
        }                                                                       \
    } while (0)

#define CIM_CHECK_HIP(call)                                                     \
    do {                                                                        \
        hipError_t e_ = (call);                                                 \
        if (e_ != hipSuccess) {                                                 \
            cim::set_error("%s: %s", __func__, hipGetErrorString(e_));          \
            return (int)e_;                                                     \
        }                                                                       \
    } while (0)

__device__ __forceinline__ float h2f(uint16_t bits) {
    return __half2float(__ushort_as_half(bits));
}

}  // namespace cim
