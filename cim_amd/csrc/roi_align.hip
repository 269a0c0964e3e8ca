// ROIAlign forward / backward for gfx950, channels-last, optionally fused with the
// MaskFuse prologue (mask multiply + channel concat).
//
// Replaces mmcv.ops.RoIAlign as used at /root/reference/lib/modeling/model_builder.py:229-231
// and the elementwise prologue of MaskFuse.forward (/root/reference/lib/modeling/resnet50.py:131-134).
// Arithmetic follows SURVEY.md App. D; the forward is evaluated with FP contraction off and
// in the reference's sample order so it is bit-identical to oracle/roi_align_ref.c.
//
// Layout: feat [B,H,W,C], out [K,P,P,C] (or cat [K,P,P,2C]).  A workgroup owns one
// (roi, bin-row); its lanes run along C, so every global access is a contiguous
// 16 B/lane x 64-lane (1 KiB per wave) segment of one feature pixel's channel vector.
#include "common.h"
#include "../../include/cim_hip.h"

namespace {

struct RoiGeom {
    float x1, y1, bw, bh, count;
    int gw, gh, b;
};

#pragma clang fp contract(off)
__device__ __forceinline__ RoiGeom roi_geom(const float* __restrict__ roi, float scale, int P, int sampling_ratio,
                                            int aligned) {
    RoiGeom g;
    const float off = aligned ? 0.5f : 0.0f;
    g.b = (int)roi[0];
    g.x1 = roi[1] * scale - off;
    g.y1 = roi[2] * scale - off;
    const float x2 = roi[3] * scale - off;
    const float y2 = roi[4] * scale - off;
    float rw = x2 - g.x1, rh = y2 - g.y1;
    if (!aligned) {
        rw = fmaxf(rw, 1.0f);
        rh = fmaxf(rh, 1.0f);
    }
    g.bh = rh / (float)P;
    g.bw = rw / (float)P;
    g.gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)P);
    g.gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)P);
    const int c = g.gh * g.gw;
    g.count = (float)(c > 1 ? c : 1);
    return g;
}

struct Tap {  // one axis of a bilinear sample
    int lo, hi;
    float l, h;
    bool valid;
};

__device__ __forceinline__ Tap make_tap(float v, int size) {
    Tap t;
    t.valid = !(v < -1.0f || v > (float)size);
    if (v <= 0.0f) v = 0.0f;
    t.lo = (int)v;
    if (t.lo >= size - 1) {
        t.hi = t.lo = size - 1;
        v = (float)t.lo;
    } else {
        t.hi = t.lo + 1;
    }
    t.l = v - (float)t.lo;
    t.h = 1.0f - t.l;
    return t;
}

template <int VEC> struct VecT;
template <> struct VecT<1> { using type = float; };
template <> struct VecT<4> { using type = float4; };

__device__ __forceinline__ float vmul(float a, float b) { return a * b; }
__device__ __forceinline__ float4 vmul(float a, float4 b) { return make_float4(a * b.x, a * b.y, a * b.z, a * b.w); }
__device__ __forceinline__ float vadd(float a, float b) { return a + b; }
__device__ __forceinline__ float4 vadd(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float vdiv(float a, float b) { return a / b; }
__device__ __forceinline__ float4 vdiv(float4 a, float b) { return make_float4(a.x / b, a.y / b, a.z / b, a.w / b); }
__device__ __forceinline__ void vzero(float& a) { a = 0.0f; }
__device__ __forceinline__ void vzero(float4& a) { a = make_float4(0.f, 0.f, 0.f, 0.f); }

// grid = (K, P); block = 256 lanes along C.
template <int VEC, bool MASKCAT>
__global__ __launch_bounds__(256) void roi_align_fwd_kernel(const float* __restrict__ feat,
                                                            const float* __restrict__ rois,
                                                            const float* __restrict__ masks,
                                                            float* __restrict__ out, int C, int H, int W, int P,
                                                            float scale, int sampling_ratio, int aligned) {
    using V = typename VecT<VEC>::type;
    const int k = blockIdx.x, ph = blockIdx.y;
    const RoiGeom g = roi_geom(rois + 5 * (size_t)k, scale, P, sampling_ratio, aligned);
    const float* __restrict__ fb = feat + (size_t)g.b * H * W * C;
    const int OC = MASKCAT ? 2 * C : C;
    for (int c = threadIdx.x * VEC; c < C; c += blockDim.x * VEC) {
        for (int pw = 0; pw < P; ++pw) {
            V acc;
            vzero(acc);
            for (int iy = 0; iy < g.gh; ++iy) {
                const float y = g.y1 + ph * g.bh + (iy + 0.5f) * g.bh / (float)g.gh;
                const Tap ty = make_tap(y, H);
                for (int ix = 0; ix < g.gw; ++ix) {
                    const float x = g.x1 + pw * g.bw + (ix + 0.5f) * g.bw / (float)g.gw;
                    const Tap tx = make_tap(x, W);
                    if (!(ty.valid && tx.valid)) continue;   // contributes 0
                    const V v1 = *reinterpret_cast<const V*>(fb + ((size_t)ty.lo * W + tx.lo) * C + c);
                    const V v2 = *reinterpret_cast<const V*>(fb + ((size_t)ty.lo * W + tx.hi) * C + c);
                    const V v3 = *reinterpret_cast<const V*>(fb + ((size_t)ty.hi * W + tx.lo) * C + c);
                    const V v4 = *reinterpret_cast<const V*>(fb + ((size_t)ty.hi * W + tx.hi) * C + c);
                    const float w1 = ty.h * tx.h, w2 = ty.h * tx.l, w3 = ty.l * tx.h, w4 = ty.l * tx.l;
                    const V val = vadd(vadd(vadd(vmul(w1, v1), vmul(w2, v2)), vmul(w3, v3)), vmul(w4, v4));
                    acc = vadd(acc, val);
                }
            }
            const V o = vdiv(acc, g.count);
            float* dst = out + (((size_t)k * P + ph) * P + pw) * OC + c;
            *reinterpret_cast<V*>(dst) = o;
            if (MASKCAT) {
                const float m = masks[((size_t)k * P + ph) * P + pw];
                *reinterpret_cast<V*>(dst + C) = vmul(m, o);
            }
        }
    }
}

__device__ __forceinline__ void atomic_add_vec(float* p, float v) { atomicAdd(p, v); }
__device__ __forceinline__ void atomic_add_vec(float* p, float4 v) {
    atomicAdd(p + 0, v.x);
    atomicAdd(p + 1, v.y);
    atomicAdd(p + 2, v.z);
    atomicAdd(p + 3, v.w);
}

// Backward: scatter g*w/count to the 4 neighbours of every in-range sample
// (roi_align_kernel.cu:237-266 of the in-tree variant; mmcv does the same).
template <int VEC, bool MASKCAT>
__global__ __launch_bounds__(256) void roi_align_bwd_kernel(const float* __restrict__ grad_out,
                                                            const float* __restrict__ rois,
                                                            const float* __restrict__ masks,
                                                            float* __restrict__ grad_in, int C, int H, int W, int P,
                                                            float scale, int sampling_ratio, int aligned) {
    using V = typename VecT<VEC>::type;
    const int k = blockIdx.x, ph = blockIdx.y;
    const RoiGeom g = roi_geom(rois + 5 * (size_t)k, scale, P, sampling_ratio, aligned);
    float* __restrict__ gb = grad_in + (size_t)g.b * H * W * C;
    const int OC = MASKCAT ? 2 * C : C;
    for (int c = threadIdx.x * VEC; c < C; c += blockDim.x * VEC) {
        for (int pw = 0; pw < P; ++pw) {
            const float* src = grad_out + (((size_t)k * P + ph) * P + pw) * OC + c;
            V go = *reinterpret_cast<const V*>(src);
            if (MASKCAT) {
                const float m = masks[((size_t)k * P + ph) * P + pw];
                go = vadd(go, vmul(m, *reinterpret_cast<const V*>(src + C)));
            }
            for (int iy = 0; iy < g.gh; ++iy) {
                const float y = g.y1 + ph * g.bh + (iy + 0.5f) * g.bh / (float)g.gh;
                const Tap ty = make_tap(y, H);
                for (int ix = 0; ix < g.gw; ++ix) {
                    const float x = g.x1 + pw * g.bw + (ix + 0.5f) * g.bw / (float)g.gw;
                    const Tap tx = make_tap(x, W);
                    if (!(ty.valid && tx.valid)) continue;
                    const float w1 = ty.h * tx.h, w2 = ty.h * tx.l, w3 = ty.l * tx.h, w4 = ty.l * tx.l;
                    atomic_add_vec(gb + ((size_t)ty.lo * W + tx.lo) * C + c, vdiv(vmul(w1, go), g.count));
                    atomic_add_vec(gb + ((size_t)ty.lo * W + tx.hi) * C + c, vdiv(vmul(w2, go), g.count));
                    atomic_add_vec(gb + ((size_t)ty.hi * W + tx.lo) * C + c, vdiv(vmul(w3, go), g.count));
                    atomic_add_vec(gb + ((size_t)ty.hi * W + tx.hi) * C + c, vdiv(vmul(w4, go), g.count));
                }
            }
        }
    }
}

template <bool MASKCAT>
int launch_fwd(const float* feat, const float* rois, const float* masks, float* out, int B, int C, int H, int W, int K,
               int P, float scale, int sr, int aligned, hipStream_t st) {
    if (K == 0) return 0;
    dim3 grid(K, P), block(256);
    if (C % 4 == 0)
        hipLaunchKernelGGL((roi_align_fwd_kernel<4, MASKCAT>), grid, block, 0, st, feat, rois, masks, out, C, H, W, P,
                           scale, sr, aligned);
    else
        hipLaunchKernelGGL((roi_align_fwd_kernel<1, MASKCAT>), grid, block, 0, st, feat, rois, masks, out, C, H, W, P,
                           scale, sr, aligned);
    return 0;
}

template <bool MASKCAT>
int launch_bwd(const float* go, const float* rois, const float* masks, float* gin, int B, int C, int H, int W, int K,
               int P, float scale, int sr, int aligned, hipStream_t st) {
    hipError_t e = hipMemsetAsync(gin, 0, sizeof(float) * (size_t)B * H * W * C, st);
    if (e != hipSuccess) return (int)e;
    if (K == 0) return 0;
    dim3 grid(K, P), block(256);
    if (C % 4 == 0)
        hipLaunchKernelGGL((roi_align_bwd_kernel<4, MASKCAT>), grid, block, 0, st, go, rois, masks, gin, C, H, W, P,
                           scale, sr, aligned);
    else
        hipLaunchKernelGGL((roi_align_bwd_kernel<1, MASKCAT>), grid, block, 0, st, go, rois, masks, gin, C, H, W, P,
                           scale, sr, aligned);
    return 0;
}

}  // namespace

#define ROI_ARGS_OK()                                                                     \
    CIM_CHECK_ARG(B > 0 && C > 0 && H > 0 && W > 0 && K >= 0 && P > 0 && P <= 65535);     \
    CIM_CHECK_ARG(rois != nullptr || K == 0)

extern "C" int cim_roi_align_fwd(const float* feat, const float* rois, float* out, int B, int C, int H, int W, int K,
                                 int P, float spatial_scale, int sampling_ratio, int aligned, void* stream) {
    ROI_ARGS_OK();
    CIM_CHECK_ARG(feat && (out || K == 0));
    int rc = launch_fwd<false>(feat, rois, nullptr, out, B, C, H, W, K, P, spatial_scale, sampling_ratio, aligned,
                               cim::as_stream(stream));
    if (rc) return rc;
    CIM_CHECK_LAUNCH();
    return 0;
}

extern "C" int cim_roi_align_bwd(const float* grad_out, const float* rois, float* grad_in, int B, int C, int H, int W,
                                 int K, int P, float spatial_scale, int sampling_ratio, int aligned, void* stream) {
    ROI_ARGS_OK();
    CIM_CHECK_ARG(grad_in && (grad_out || K == 0));
    int rc = launch_bwd<false>(grad_out, rois, nullptr, grad_in, B, C, H, W, K, P, spatial_scale, sampling_ratio,
                               aligned, cim::as_stream(stream));
    if (rc) return rc;
    CIM_CHECK_LAUNCH();
    return 0;
}

extern "C" int cim_roi_align_maskcat_fwd(const float* feat, const float* rois, const float* masks, float* cat, int B,
                                         int C, int H, int W, int K, int P, float spatial_scale, int sampling_ratio,
                                         int aligned, void* stream) {
    ROI_ARGS_OK();
    CIM_CHECK_ARG(feat && ((cat && masks) || K == 0));
    int rc = launch_fwd<true>(feat, rois, masks, cat, B, C, H, W, K, P, spatial_scale, sampling_ratio, aligned,
                              cim::as_stream(stream));
    if (rc) return rc;
    CIM_CHECK_LAUNCH();
    return 0;
}

extern "C" int cim_roi_align_maskcat_bwd(const float* grad_cat, const float* rois, const float* masks, float* grad_in,
                                         int B, int C, int H, int W, int K, int P, float spatial_scale,
                                         int sampling_ratio, int aligned, void* stream) {
    ROI_ARGS_OK();
    CIM_CHECK_ARG(grad_in && ((grad_cat && masks) || K == 0));
    int rc = launch_bwd<true>(grad_cat, rois, masks, grad_in, B, C, H, W, K, P, spatial_scale, sampling_ratio, aligned,
                              cim::as_stream(stream));
    if (rc) return rc;
    CIM_CHECK_LAUNCH();
    return 0;
}
