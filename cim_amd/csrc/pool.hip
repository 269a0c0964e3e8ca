// Max-pooling and nearest-neighbour up-sampling of the bodies: the last per-step operators that ran on ATen.
//
// Replaces nn.MaxPool2d(3, 2, 1) of the ResNet stem (/root/reference/lib/modeling/resnet50.py:29 through torchvision's resnet50),
// nn.MaxPool2d(2, 2) at the end of VGG16's conv1 / conv2 / conv3 (/root/reference/lib/modeling/vgg16.py:43,50,60) and
// nn.Upsample(scale_factor = 2^k, mode = 'nearest') of HRNet's fuse layers (/root/reference/lib/modeling/HRNet.py:201).
// NCHW fp32 planes (the bodies' layout: a 1 x 1 convolution is W . X[Cin, HW]); HBM streaming, one thread per output element,
// lanes along W (block = 64 columns x 4 rows: no per-element 64-bit divisions).  Semantics are ATen's: the window is scanned rows first
// and a value replaces the running maximum when it is greater OR NaN (the first maximum of a window wins a tie; NaN propagates); floor output size, no dilation.  The backward is a
// GATHER (a thread owns an input pixel and adds the gradients of the <= ceil(k / s)^2 windows whose arg-max it is, in window order):
// no atomics, deterministic - ATen scatters with atomicAdd.
#include "common.h"
#include "../../include/cim_hip.h"

namespace {

__global__ __launch_bounds__(256) void maxpool2d_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int* __restrict__ idx,
                                                            long long rows, int H, int W, int Ho, int Wo, int k, int s, int p) {
    // block = (64 columns, 4 output rows); grid = (row blocks, column blocks): one 32-bit division per thread
    const long long row = (long long)blockIdx.x * 4 + threadIdx.y;
    const int ow = blockIdx.y * 64 + threadIdx.x;
    if (row >= rows || ow >= Wo) return;
    const long long nc = (unsigned)row / (unsigned)Ho;           // (rows < 2^31: checked by the launcher)
    const int oh = (int)(row - nc * Ho);
    const long long i = row * Wo + ow;
    const float* __restrict__ src = x + nc * (long long)H * W;
    const int h0 = max(oh * s - p, 0), h1 = min(oh * s - p + k, H);
    const int w0 = max(ow * s - p, 0), w1 = min(ow * s - p + k, W);
    float best = -INFINITY;
    int at = h0 * W + w0;
    for (int h = h0; h < h1; ++h)
        for (int w = w0; w < w1; ++w) {
            const float v = src[h * W + w];
            if (v > best || v != v) {
                best = v;
                at = h * W + w;
            }
        }
    y[i] = best;
    if (idx != nullptr) idx[i] = at;
}

__global__ __launch_bounds__(256) void maxpool2d_bwd_kernel(const float* __restrict__ dy, const int* __restrict__ idx,
                                                            float* __restrict__ dx, long long rows, int H, int W, int Ho, int Wo,
                                                            int k, int s, int p) {
    const long long row = (long long)blockIdx.x * 4 + threadIdx.y;
    const int w = blockIdx.y * 64 + threadIdx.x;
    if (row >= rows || w >= W) return;
    const long long nc = (unsigned)row / (unsigned)H;            // (rows < 2^31: checked by the launcher)
    const int h = (int)(row - nc * H);
    const long long i = row * W + w;
    // windows that contain (h, w): oh * s - p <= h < oh * s - p + k
    const int oh0 = max((h + p - k + s) / s, 0), oh1 = min((h + p) / s, Ho - 1);
    const int ow0 = max((w + p - k + s) / s, 0), ow1 = min((w + p) / s, Wo - 1);
    const int self = h * W + w;
    const long long base = nc * (long long)Ho * Wo;
    float g = 0.0f;
    for (int oh = oh0; oh <= oh1; ++oh)
        for (int ow = ow0; ow <= ow1; ++ow)
            if (idx[base + (long long)oh * Wo + ow] == self) g += dy[base + (long long)oh * Wo + ow];
    dx[i] = g;
}

// y[nc][oh][ow] = x[nc][oh / s][ow / s]; ADD: accumulate into y (the fuse layers sum their branches)
template <bool ADD>
__global__ __launch_bounds__(256) void upsample_nearest_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long long rows,
                                                                   int H, int W, int s) {
    const int Wo = W * s, Ho = H * s;
    const long long row = (long long)blockIdx.x * 4 + threadIdx.y;
    const int ow = blockIdx.y * 64 + threadIdx.x;
    if (row >= rows || ow >= Wo) return;
    const long long nc = (unsigned)row / (unsigned)Ho;           // (rows < 2^31: checked by the launcher)
    const int oh = (int)(row - nc * Ho);
    const long long i = row * Wo + ow;
    const float v = x[(nc * H + oh / s) * W + ow / s];
    y[i] = ADD ? y[i] + v : v;
}

// dx[nc][h][w] = sum of the s x s block of dy, rows first (ATen's order)
__global__ __launch_bounds__(256) void upsample_nearest_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, long long rows,
                                                                   int H, int W, int s) {
    const long long row = (long long)blockIdx.x * 4 + threadIdx.y;
    const int w = blockIdx.y * 64 + threadIdx.x;
    if (row >= rows || w >= W) return;
    const long long nc = (unsigned)row / (unsigned)H;            // (rows < 2^31: checked by the launcher)
    const int h = (int)(row - nc * H);
    const long long i = row * W + w;
    const int Wo = W * s;
    const float* __restrict__ src = dy + (nc * H * s + (long long)h * s) * Wo + (long long)w * s;
    float g = 0.0f;
    for (int a = 0; a < s; ++a)
        for (int b = 0; b < s; ++b) g += src[(long long)a * Wo + b];
    dx[i] = g;
}

inline dim3 grid_of(long long rows, int cols) { return dim3((unsigned)((rows + 3) / 4), (unsigned)((cols + 63) / 64)); }
const dim3 kBlock(64, 4);

}  // namespace

extern "C" int cim_maxpool2d_out_size(int in, int k, int stride, int pad) { return (in + 2 * pad - k) / stride + 1; }

extern "C" int cim_maxpool2d_fwd(const float* x, float* y, int* idx, int NC, int H, int W, int k, int stride, int pad, void* stream) {
    CIM_CHECK_ARG(x && y && NC > 0 && H > 0 && W > 0 && k >= 1 && k <= 7 && stride >= 1 && pad >= 0 && 2 * pad <= k);
    CIM_CHECK_ARG((long long)H * W < (1ll << 31) && H + 2 * pad >= k && W + 2 * pad >= k);
    const int Ho = cim_maxpool2d_out_size(H, k, stride, pad), Wo = cim_maxpool2d_out_size(W, k, stride, pad);
    const long long rows = (long long)NC * Ho;
    CIM_CHECK_ARG(rows < (1ll << 31) && Wo / 64 < 65535);
    hipLaunchKernelGGL(maxpool2d_fwd_kernel, grid_of(rows, Wo), kBlock, 0, cim::as_stream(stream), x, y, idx, rows, H, W, Ho, Wo, k,
                       stride, pad);
    CIM_CHECK_LAUNCH();
    return 0;
}

extern "C" int cim_maxpool2d_bwd(const float* dy, const int* idx, float* dx, int NC, int H, int W, int k, int stride, int pad,
                                 void* stream) {
    CIM_CHECK_ARG(dy && idx && dx && NC > 0 && H > 0 && W > 0 && k >= 1 && k <= 7 && stride >= 1 && pad >= 0 && 2 * pad <= k);
    CIM_CHECK_ARG((long long)H * W < (1ll << 31) && H + 2 * pad >= k && W + 2 * pad >= k);
    const int Ho = cim_maxpool2d_out_size(H, k, stride, pad), Wo = cim_maxpool2d_out_size(W, k, stride, pad);
    const long long rows = (long long)NC * H;
    CIM_CHECK_ARG(rows < (1ll << 31) && W / 64 < 65535);
    hipLaunchKernelGGL(maxpool2d_bwd_kernel, grid_of(rows, W), kBlock, 0, cim::as_stream(stream), dy, idx, dx, rows, H, W, Ho, Wo, k,
                       stride, pad);
    CIM_CHECK_LAUNCH();
    return 0;
}

extern "C" int cim_upsample_nearest_fwd(const float* x, float* y, int NC, int H, int W, int scale, int accumulate, void* stream) {
    CIM_CHECK_ARG(x && y && NC > 0 && H > 0 && W > 0 && scale >= 1 && scale <= 64);
    const long long rows = (long long)NC * H * scale;
    CIM_CHECK_ARG(rows < (1ll << 31) && (long long)W * scale / 64 < 65535);
    if (accumulate)
        hipLaunchKernelGGL(upsample_nearest_fwd_kernel<true>, grid_of(rows, W * scale), kBlock, 0, cim::as_stream(stream), x, y, rows, H, W,
                           scale);
    else
        hipLaunchKernelGGL(upsample_nearest_fwd_kernel<false>, grid_of(rows, W * scale), kBlock, 0, cim::as_stream(stream), x, y, rows, H, W,
                           scale);
    CIM_CHECK_LAUNCH();
    return 0;
}

extern "C" int cim_upsample_nearest_bwd(const float* dy, float* dx, int NC, int H, int W, int scale, void* stream) {
    CIM_CHECK_ARG(dy && dx && NC > 0 && H > 0 && W > 0 && scale >= 1 && scale <= 64);
    const long long rows = (long long)NC * H;
    CIM_CHECK_ARG(rows < (1ll << 31) && W / 64 < 65535);
    hipLaunchKernelGGL(upsample_nearest_bwd_kernel, grid_of(rows, W), kBlock, 0, cim::as_stream(stream), dy, dx, rows, H, W, scale);
    CIM_CHECK_LAUNCH();
    return 0;
}
