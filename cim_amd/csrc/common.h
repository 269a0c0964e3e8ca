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

// ---- pair images (f16x2p GEMM engine, gemm_pair.hip): x * s = h + l, two fp16 terms ------------------------------------
typedef _Float16 pair_f16x2 __attribute__((ext_vector_type(2)));
typedef float pair_f32x2 __attribute__((ext_vector_type(2)));
// (a, b) already scaled -> packed f16 pairs (a in the low half) of the two terms
__device__ __forceinline__ void pair_split2(float a, float b, unsigned& h, unsigned& l) {
    const pair_f32x2 v = {a, b};
    const pair_f16x2 hv = __builtin_convertvector(v, pair_f16x2);          // v_cvt_pk_f16_f32 (RNE)
    const pair_f32x2 r = {a - (float)hv.x, b - (float)hv.y};               // exact
    h = __builtin_bit_cast(unsigned, hv);
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(r, pair_f16x2));
}
// |max| bit pattern -> power-of-two scale 2^(14 - e): |x| s < 2^15 (biased exponent clamped so that s and 1 / s are normal)
__device__ __forceinline__ float pair_scale_of(unsigned amax_bits) {
    const int e = (int)((amax_bits >> 23) & 0xffu);
    int b = 268 - e;
    if (b > 253) b = 253;
    if (e == 0 || e == 255) b = 127;
    return __uint_as_float((unsigned)b << 23);
}

// Running maximum of bit patterns in ONE device word (max |x| of a tensor: the scale source of a pair image).  Thousands of
// atomics on one address serialise in L2 (measured: +80 us on a 135 us kernel with one atomicMax per wave), so a wave
// first looks at the word - it only grows - and skips the atomic when it cannot raise it: after the first few waves
// almost all do.
__device__ __forceinline__ void amax_publish(unsigned* word, unsigned wave_max) {
    if (wave_max > __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(word, wave_max);      // (agent scope: past the CU's L1)
}

__device__ __forceinline__ float h2f(uint16_t bits) {
    return __half2float(__ushort_as_half(bits));
}

}  // namespace cim
