// Frozen-statistics BatchNorm + residual add + ReLU of the ResNet / HRNet bodies, fused, forward and backward.
//
// Replaces the nn.BatchNorm2d (eval mode: running statistics, trainable affine - /root/reference/lib/modeling/
// resnet50.py:53-77 keeps every BN in eval()) -> (+ identity) -> nn.ReLU chains of torchvision's Bottleneck as wrapped
// by /root/reference/lib/modeling/resnet50.py:17-44.  In ATen these are 2-3 launches forward and 3-4 backward per
// BN, each a full pass over the activation; here one launch each way:
//   forward :  y = relu?(x * a[c] + b[c] (+ res)),   a = gamma * rsqrt(var + eps),  b = beta - mean * a
//   backward:  dz = dy * (y > 0) (ReLU) ; dx = dz * a[c] ; dres = dz ; dgamma[c] = sum dz * (x - mean) * rstd ;
//              dbeta[c] = sum dz
// Layout NCHW (what MIOpen's fp32 convolutions produce here), one workgroup per (image, channel) plane chunk forward,
// one workgroup per channel backward (the plane sums stay in the workgroup: no atomics, deterministic).
// HBM-streaming kernels: forward 8-12 B / element, backward 16-20 B / element.
#include "common.h"
#include "../../include/cim_hip.h"

namespace {

__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const float* __restrict__ mean, const float* __restrict__ var,
                                                         float eps, float* __restrict__ y, int C, int HW, int relu) {
    const int plane = blockIdx.x;                 // n * C + c
    const int c = plane % C;
    const float a = gamma[c] * rsqrtf(var[c] + eps);
    const float b = beta[c] - mean[c] * a;
    const size_t base = (size_t)plane * HW;
    for (int i = blockIdx.y * 256 + threadIdx.x; i < HW; i += gridDim.y * 256) {
        float v = fmaf(x[base + i], a, b);
        if (res != nullptr) v += res[base + i];
        y[base + i] = relu ? fmaxf(v, 0.0f) : v;
    }
}

// grid = (C, chunks); block = 512.  A channel's N x HW elements are cut into `chunks` contiguous ranges so that narrow,
// high-resolution layers (HRNet: 48 channels x 24k pixels) still fill the chip; with chunks > 1 the per-channel sums
// meet through atomicAdd into caller-zeroed dgamma / dbeta.
__global__ __launch_bounds__(512) void bn_act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                         const float* __restrict__ x, const float* __restrict__ gamma,
                                                         const float* __restrict__ mean, const float* __restrict__ var,
                                                         float eps, float* __restrict__ dx, float* __restrict__ dres,
                                                         float* __restrict__ dgamma, float* __restrict__ dbeta, int N,
                                                         int C, int HW, int relu) {
    __shared__ float s1[8], s2[8];
    const int c = blockIdx.x, tid = threadIdx.x;
    const float rstd = rsqrtf(var[c] + eps), mu = mean[c];
    const float a = gamma[c] * rstd;
    const long long total = (long long)N * HW;
    const long long per = (total + gridDim.y - 1) / gridDim.y;
    const long long e0 = (long long)blockIdx.y * per, e1 = min(total, e0 + per);
    float sum_dz = 0.0f, sum_dzx = 0.0f;
    for (long long e = e0 + tid; e < e1; e += 512) {
        const int n = (int)(e / HW);
        const size_t i = ((size_t)n * C + c) * HW + (size_t)(e - (long long)n * HW);
        float dz = dy[i];
        if (relu && !(y[i] > 0.0f)) dz = 0.0f;
        if (dx != nullptr) dx[i] = dz * a;
        if (dres != nullptr) dres[i] = dz;
        sum_dz += dz;
        sum_dzx = fmaf(dz, x[i] - mu, sum_dzx);
    }
    if (dgamma == nullptr) return;                // frozen affine: block-uniform
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        sum_dz += __shfl_xor(sum_dz, o);
        sum_dzx += __shfl_xor(sum_dzx, o);
    }
    if ((tid & 63) == 0) { s1[tid >> 6] = sum_dz; s2[tid >> 6] = sum_dzx; }
    __syncthreads();
    if (tid == 0) {
        float t1 = 0.0f, t2 = 0.0f;
        for (int w = 0; w < 8; ++w) { t1 += s1[w]; t2 += s2[w]; }
        if (gridDim.y > 1) {
            atomicAdd(dbeta + c, t1);
            atomicAdd(dgamma + c, t2 * rstd);
        } else {
            dbeta[c] = t1;
            dgamma[c] = t2 * rstd;
        }
    }
}

}  // namespace

extern "C" int cim_bn_act_fwd(const float* x, const float* res, const float* gamma, const float* beta, const float* mean,
                              const float* var, float eps, float* y, int N, int C, int HW, int relu, void* stream) {
    CIM_CHECK_ARG(x && y && gamma && beta && mean && var && N > 0 && C > 0 && HW > 0);
    CIM_CHECK_ARG((long long)N * C <= 2147483647LL);
    int chunks = (HW + 256 * 8 - 1) / (256 * 8);          // ~8 elements per lane
    if (chunks > 65535) chunks = 65535;
    hipLaunchKernelGGL(bn_act_fwd_kernel, dim3(N * C, chunks), dim3(256), 0, cim::as_stream(stream), x, res, gamma, beta, mean,
                       var, eps, y, C, HW, relu);
    CIM_CHECK_LAUNCH();
    return 0;
}

// > 1: the launch accumulates dgamma / dbeta with atomicAdd and the caller must zero them first
extern "C" int cim_bn_act_bwd_chunks(int N, int C, int HW) {
    if (N <= 0 || C <= 0 || HW <= 0) return 1;
    const long long total = (long long)N * HW;
    if (C >= 128) return 1;                                    // enough workgroups: plain stores, no zero-fill launch
    long long want = (1024 + C - 1) / C;                       // ~1024 workgroups
    const long long cap = (total + 4095) / 4096;               // >= 8 elements per lane
    if (want > cap) want = cap;
    return (int)(want < 1 ? 1 : want);
}

extern "C" int cim_bn_act_bwd(const float* dy, const float* y, const float* x, const float* gamma, const float* mean,
                              const float* var, float eps, float* dx, float* dres, float* dgamma, float* dbeta, int N, int C,
                              int HW, int relu, void* stream) {
    CIM_CHECK_ARG(dy && x && gamma && mean && var && N > 0 && C > 0 && HW > 0);
    CIM_CHECK_ARG((y != nullptr || !relu) && ((dgamma == nullptr) == (dbeta == nullptr)));
    const int chunks = cim_bn_act_bwd_chunks(N, C, HW);
    hipLaunchKernelGGL(bn_act_bwd_kernel, dim3(C, chunks), dim3(512), 0, cim::as_stream(stream), dy, y, x, gamma, mean, var,
                       eps, dx, dres, dgamma, dbeta, N, C, HW, relu);
    CIM_CHECK_LAUNCH();
    return 0;
}
