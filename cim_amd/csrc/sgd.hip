// Fused multi-tensor SGD with momentum and weight decay for the 250-580 M parameters of the CIM models
// (SURVEY.md 8 f-4).  Replaces torch.optim.SGD(params, momentum=...) as constructed at
// /root/reference/tools/train.py:282-311 (two groups: weights, and biases at 2 x lr without weight decay) and stepped
// at :438; the update rule is torch.optim.SGD's (dampening 0, no Nesterov):
//     g' = g + wd * p ;   buf = momentum * buf + g' ;   p = p - lr * buf
// with buf zero-initialised (identical to torch's "first step: buf = g'").  The momentum buffers stay ordinary tensors
// in optimizer.state[p]['momentum_buffer'], so lib/utils/net.py:47-83 (update_learning_rate, _CorrectMomentum) and the
// checkpoint code work unchanged.
//
// One launch for ALL tensors: the host builds a table of fixed-size chunks (tensor pointers + element range + the
// group's lr / wd) once per step; a workgroup streams one chunk with 16 B accesses: 12 B read + 8 B written per
// parameter, HBM-bound.
#include "common.h"
#include "../../include/cim_hip.h"

namespace {

__global__ __launch_bounds__(256) void sgd_multi_kernel(const cim_sgd_chunk* __restrict__ table, float momentum) {
    const cim_sgd_chunk ch = table[blockIdx.x];
    float* __restrict__ p = reinterpret_cast<float*>(ch.p);
    const float* __restrict__ g = reinterpret_cast<const float*>(ch.g);
    float* __restrict__ b = reinterpret_cast<float*>(ch.buf);
    const float lr = ch.lr, wd = ch.wd;
    const int n4 = ((ch.aligned != 0) ? ch.n : 0) & ~3;
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
    for (int i = n4 + threadIdx.x; i < ch.n; i += 256) {      // unaligned tensors / tails
        const float bv = fmaf(momentum, b[i], fmaf(wd, p[i], g[i]));
        b[i] = bv;
        p[i] = fmaf(-lr, bv, p[i]);
    }
}

}  // namespace

extern "C" int cim_sgd_multi(const cim_sgd_chunk* table, int n_chunks, float momentum, void* stream) {
    CIM_CHECK_ARG(table != nullptr || n_chunks == 0);
    CIM_CHECK_ARG(n_chunks >= 0);
    if (n_chunks == 0) return 0;
    hipLaunchKernelGGL(sgd_multi_kernel, dim3(n_chunks), dim3(256), 0, cim::as_stream(stream), table, momentum);
    CIM_CHECK_LAUNCH();
    return 0;
}
