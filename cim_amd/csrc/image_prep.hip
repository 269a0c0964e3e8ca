// Network-input image preparation on the device (the data side of the training / inference step).
//
// Replaces, for cfg.transform_mode == "ToTensor" (all shipped configs), the host chain of
// /root/reference/lib/utils/blob.py:93-147 (prep_im_for_blob) as driven by lib/roi_data/minibatch.py:109-150 and
// lib/core/test.py:464-473:  BGR uint8 image (cv2.imread) [-> horizontal flip] -> float32 -> cv2.resize(fx = fy = scale,
// INTER_LINEAR) -> np.uint8 (truncation) -> BGR2RGB -> ToTensor (/255) -> Normalize(mean, std) -> CHW float32.
// One launch, one thread per output pixel; the source image is read through L2 (4 neighbours x 3 bytes).
//
// cv2 is a third-party dependency that is not part of the reference tree (nor of this image): the arithmetic follows
// OpenCV 4.x modules/imgproc/src/resize.cpp, INTER_LINEAR on CV_32FC3 (resizeGeneric_ / HResizeLinear / VResizeLinear):
//   scale = 1 / fx (double);  f = (float)((d + 0.5) * scale - 0.5);  s = floor(f);  f -= s;
//   columns: s < 0 -> (s, f) = (0, 0);  s >= w - 1 -> D = S[w - 1];  rows: indices clipped to [0, h - 1], weights kept;
//   horizontal pass first (D = S[s] * (1 - f) + S[s + 1] * f), then vertical (dst = R0 * (1 - fy) + R1 * fy),
//   every product and sum rounded to float32 (no fused multiply-add).
#include "common.h"
#include "../../include/cim_hip.h"

namespace {

#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void image_prep_kernel(const uint8_t* __restrict__ src, int h, int w, float* __restrict__ dst,
                                                         int H, int W, long long plane_stride, int row_stride, double scale,
                                                         int hflip, float m0, float m1, float m2, float s0, float s1, float s2) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    float fx = (float)(((double)x + 0.5) * scale - 0.5);
    int sx = (int)floorf(fx);
    fx -= (float)sx;
    if (sx < 0) { fx = 0.0f; sx = 0; }
    const bool edge = sx >= w - 1;
    if (edge) { fx = 0.0f; sx = w - 1; }
    float fy = (float)(((double)y + 0.5) * scale - 0.5);
    const int sy = (int)floorf(fy);
    fy -= (float)sy;
    const int y0 = min(max(sy, 0), h - 1), y1 = min(max(sy + 1, 0), h - 1);
    const int xa = hflip ? (w - 1 - sx) : sx;
    const int xb = edge ? xa : (hflip ? xa - 1 : xa + 1);
    const float a0 = 1.0f - fx, a1 = fx, b0 = 1.0f - fy, b1 = fy;
    const uint8_t* r0 = src + (size_t)y0 * w * 3;
    const uint8_t* r1 = src + (size_t)y1 * w * 3;
    const float mean[3] = {m0, m1, m2}, stdv[3] = {s0, s1, s2};
#pragma unroll
    for (int c = 0; c < 3; ++c) {                        // c = output (RGB) channel; source is BGR
        const int sc = 2 - c;
        float t0, t1;
        if (edge) {
            t0 = (float)r0[xa * 3 + sc];
            t1 = (float)r1[xa * 3 + sc];
        } else {
            const float p = (float)r0[xa * 3 + sc] * a0, q = (float)r0[xb * 3 + sc] * a1;
            t0 = p + q;
            const float p1 = (float)r1[xa * 3 + sc] * a0, q1 = (float)r1[xb * 3 + sc] * a1;
            t1 = p1 + q1;
        }
        const float u = t0 * b0, v = t1 * b1;
        float val = u + v;
        val = fminf(fmaxf(val, 0.0f), 255.0f);
        const float byte = (float)(int)val;              // np.uint8(): truncation
        const float unit = byte / 255.0f;                // ToTensor
        const float d = unit - mean[c];
        dst[(size_t)c * plane_stride + (size_t)y * row_stride + x] = d / stdv[c];     // Normalize
    }
}

}  // namespace

extern "C" int cim_image_prep(const uint8_t* src_bgr, int h, int w, float* dst, int H, int W, long long plane_stride,
                              int row_stride, double inv_scale, int hflip, const float* mean_std6_host, void* stream) {
    CIM_CHECK_ARG(src_bgr && dst && mean_std6_host && h > 0 && w > 0 && H > 0 && W > 0 && row_stride >= W && inv_scale > 0.0);
    const float* ms = mean_std6_host;
    hipLaunchKernelGGL(image_prep_kernel, dim3((W + 63) / 64, (H + 3) / 4), dim3(256), 0, cim::as_stream(stream), src_bgr, h, w,
                       dst, H, W, plane_stride, row_stride, inv_scale, hflip, ms[0], ms[1], ms[2], ms[3], ms[4], ms[5]);
    CIM_CHECK_LAUNCH();
    return 0;
}
