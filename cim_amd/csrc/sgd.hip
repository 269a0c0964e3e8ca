// Fused multi-tensor SGD with momentum and weight decay for the 250-580 M parameters of the CIM models
// (SURVEY.md 8 f-4).  Replaces torch.optim.SGD(params, momentum=...) as constructed at
// /root/reference/tools/train.py:282-311 (two groups: weights, and biases at 2 x lr without weight decay) and stepped
// at :438; the update rule is torch.optim.SGD's (dampening 0, no Nesterov):
//     g' = g + wd * p ;   buf = momentum * buf + g' ;   p = p - lr * buf
// with buf zero-initialised (identical to torch's "first step: buf = g'").  The momentum buffers stay ordinary tensors
// in optimizer.state[p]['momentum_buffer'], so lib/utils/net.py:47-83 (update_learning_rate, _CorrectMomentum) and the
// checkpoint code work unchanged.
//
// One launch for ALL tensors.  Two device tables: `chunks` (tensor index + element offset of every 16384-element chunk;
// built once, it only depends on the tensor sizes) and `tensors` (pointers, element count, lr, wd per tensor: 40 bytes
// each, refreshed per step - autograd hands out new gradient tensors every step, the chunk layout never changes).
// A workgroup streams one chunk with 16 B accesses: 12 B read + 8 B written per parameter, HBM-bound.
#include "common.h"
#include "../../include/cim_hip.h"

namespace {

__global__ __launch_bounds__(256) void sgd_multi_kernel(const cim_sgd_tensor* __restrict__ tensors,
                                                        const cim_sgd_chunk* __restrict__ chunks, float momentum) {
    const cim_sgd_chunk ch = chunks[blockIdx.x];
    const cim_sgd_tensor t = tensors[ch.tensor];
    float* __restrict__ p = reinterpret_cast<float*>(t.p) + ch.offset;
    const float* __restrict__ g = reinterpret_cast<const float*>(t.g) + ch.offset;
    float* __restrict__ b = reinterpret_cast<float*>(t.buf) + ch.offset;
    const float lr = t.lr, wd = t.wd;
    const int n = min(ch.n, (int)(t.n - ch.offset));
    // 16-byte accesses need the three chunk bases aligned (chunk offsets are multiples of 4 elements)
    const bool aligned = ((t.p | t.g | t.buf) & 15) == 0;
    const int n4 = aligned ? (n & ~3) : 0;
    for (int i = threadIdx.x * 4; i < n4; i += 256 * 4) {
        float4 pv = *reinterpret_cast<const float4*>(p + i);
        const float4 gv = *reinterpret_cast<const float4*>(g + i);
        float4 bv = *reinterpret_cast<const float4*>(b + i);
        bv.x = fmaf(momentum, bv.x, fmaf(wd, pv.x, gv.x));
        bv.y = fmaf(momentum, bv.y, fmaf(wd, pv.y, gv.y));
        bv.z = fmaf(momentum, bv.z, fmaf(wd, pv.z, gv.z));
        bv.w = fmaf(momentum, bv.w, fmaf(wd, pv.w, gv.w));
        pv.x = fmaf(-lr, bv.x, pv.x);
        pv.y = fmaf(-lr, bv.y, pv.y);
        pv.z = fmaf(-lr, bv.z, pv.z);
        pv.w = fmaf(-lr, bv.w, pv.w);
        *reinterpret_cast<float4*>(b + i) = bv;
        *reinterpret_cast<float4*>(p + i) = pv;
    }
    for (int i = n4 + threadIdx.x; i < n; i += 256) {      // unaligned tensors / tails
        const float bv = fmaf(momentum, b[i], fmaf(wd, p[i], g[i]));
        b[i] = bv;
        p[i] = fmaf(-lr, bv, p[i]);
    }
}

}  // namespace

extern "C" int cim_sgd_multi(const cim_sgd_tensor* tensors, const cim_sgd_chunk* chunks, int n_chunks, float momentum,
                             void* stream) {
    CIM_CHECK_ARG((tensors != nullptr && chunks != nullptr) || n_chunks == 0);
    CIM_CHECK_ARG(n_chunks >= 0);
    if (n_chunks == 0) return 0;
    hipLaunchKernelGGL(sgd_multi_kernel, dim3(n_chunks), dim3(256), 0, cim::as_stream(stream), tensors, chunks, momentum);
    CIM_CHECK_LAUNCH();
    return 0;
}
