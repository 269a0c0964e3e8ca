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
// built once, it only depends on the tensor sizes) and `tensors` (cim_sgd_tensor: pointers, element count, lr, wd, matrix shape and |max|
// array pointers per tensor: 64 bytes each, refreshed per step - autograd hands out new gradient tensors every step, the chunk layout never changes).
// A workgroup streams one chunk with 16 B accesses: 12 B read + 8 B written per parameter, HBM-bound.
#include "common.h"
#include "../../include/cim_hip.h"

namespace {

// wave-wide maximum on the VALU (see amax_wave in gemm_f32.hip): valid in lane 63
#ifndef CIM_SGD_NT
#define CIM_SGD_NT 1            // 1 = nontemporal stores of the updated parameter / history (matrix mode)
#endif
typedef float sgd_v4 __attribute__((ext_vector_type(4)));
#ifndef CIM_SGD_NTL
#define CIM_SGD_NTL 0           // 1 = nontemporal loads too: measured slower (0.919 vs 0.864 ms at cfg2)
#endif
__device__ __forceinline__ float4 sgd_load(const float* p) {
#if CIM_SGD_NTL
    const sgd_v4 v = __builtin_nontemporal_load(reinterpret_cast<const sgd_v4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
#else
    return *reinterpret_cast<const float4*>(p);
#endif
}
__device__ __forceinline__ void sgd_store(float* p, const float4& v) {
#if CIM_SGD_NT
    __builtin_nontemporal_store(sgd_v4{v.x, v.y, v.z, v.w}, reinterpret_cast<sgd_v4*>(p));
#else
    *reinterpret_cast<float4*>(p) = v;
#endif
}

__device__ __forceinline__ unsigned sgd_wave_max(unsigned v) {
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, true));
    return v;
}

__device__ __forceinline__ float4 sgd_update4(float4& pv, const float4& gv, float4& bv, float momentum, float lr, float wd) {
    bv.x = fmaf(momentum, bv.x, fmaf(wd, pv.x, gv.x));
    bv.y = fmaf(momentum, bv.y, fmaf(wd, pv.y, gv.y));
    bv.z = fmaf(momentum, bv.z, fmaf(wd, pv.z, gv.z));
    bv.w = fmaf(momentum, bv.w, fmaf(wd, pv.w, gv.w));
    pv.x = fmaf(-lr, bv.x, pv.x);
    pv.y = fmaf(-lr, bv.y, pv.y);
    pv.z = fmaf(-lr, bv.z, pv.z);
    pv.w = fmaf(-lr, bv.w, pv.w);
    return pv;
}

// A launch of fewer workgroups than chunks WALKS over them (workgroup b takes chunks b, b + grid, ...): cim_sgd_multi's
// max_workgroups - an update that runs beside another stream's latency-bound launches (cim_amd.optim.SGD.overlap_update: under the next
// step's backbone forward) then holds ONE or TWO of a CU's workgroup slots instead of all of them, and those launches' workgroups
// are placed at once instead of queueing behind 15 k chunk workgroups.
__global__ __launch_bounds__(256) void sgd_multi_kernel(const cim_sgd_tensor* __restrict__ tensors,
                                                        const cim_sgd_chunk* __restrict__ chunks, int n_chunks, float momentum) {
  for (int ci = blockIdx.x; ci < n_chunks; ci += gridDim.x) {
    const cim_sgd_chunk ch = chunks[ci];
    const cim_sgd_tensor t = tensors[ch.tensor];
    const float lr = t.lr, wd = t.wd;
    if (t.cols > 0) {
        // matrix mode ([rows][cols] weights of the MaskFuse contractions): a 64-row x 1024-column tile per workgroup, a lane
        // owns 4 consecutive columns - the update of the flat mode plus the per-row / per-column max |w_new| the f16x2 GEMM
        // engine needs as operand scales (saves its pass over the 822 MB fc1 weight).  ch.offset = first row, ch.n = first column.
        const int r0 = (int)ch.offset, r1 = min((int)t.rows, r0 + 64);
        const int c = ch.n + threadIdx.x * 4;
        const bool cin = c < t.cols;
        const size_t base = (size_t)r0 * t.cols + (cin ? c : 0);
        float* __restrict__ p = reinterpret_cast<float*>(t.p) + base;
        const float* __restrict__ g = reinterpret_cast<const float*>(t.g) + base;
        float* __restrict__ b = reinterpret_cast<float*>(t.buf) + base;
        unsigned* row_amax = reinterpret_cast<unsigned*>(t.row_amax);
        unsigned* col_amax = reinterpret_cast<unsigned*>(t.col_amax);
        const bool lead = (threadIdx.x & 63) == 63;
        uint4 cm = make_uint4(0, 0, 0, 0);
        const size_t ld = (size_t)t.cols;
        int r = r0;
        for (; r + 4 <= r1; r += 4, p += 4 * ld, g += 4 * ld, b += 4 * ld) {      // 4 rows (12 x 16 B loads) in flight per lane
            float4 pv[4], gv[4], bv[4];
            unsigned m[4] = {0, 0, 0, 0};
            if (cin) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    pv[i] = sgd_load(p + i * ld);
                    gv[i] = sgd_load(g + i * ld);
                    bv[i] = sgd_load(b + i * ld);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    sgd_update4(pv[i], gv[i], bv[i], momentum, lr, wd);
                    sgd_store(b + i * ld, bv[i]);
                    sgd_store(p + i * ld, pv[i]);
                    const unsigned ax = __float_as_uint(pv[i].x) & 0x7fffffffu, ay = __float_as_uint(pv[i].y) & 0x7fffffffu;
                    const unsigned az = __float_as_uint(pv[i].z) & 0x7fffffffu, aw = __float_as_uint(pv[i].w) & 0x7fffffffu;
                    cm.x = max(cm.x, ax); cm.y = max(cm.y, ay); cm.z = max(cm.z, az); cm.w = max(cm.w, aw);
                    m[i] = max(max(ax, ay), max(az, aw));
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned mm = sgd_wave_max(m[i]);
                if (lead) atomicMax(row_amax + r + i, mm);
            }
        }
        for (; r < r1; ++r, p += ld, g += ld, b += ld) {
            unsigned m = 0;
            if (cin) {
                float4 pv = sgd_load(p);
                const float4 gv = sgd_load(g);
                float4 bv = sgd_load(b);
                sgd_update4(pv, gv, bv, momentum, lr, wd);
                sgd_store(b, bv);
                sgd_store(p, pv);
                const unsigned ax = __float_as_uint(pv.x) & 0x7fffffffu, ay = __float_as_uint(pv.y) & 0x7fffffffu;
                const unsigned az = __float_as_uint(pv.z) & 0x7fffffffu, aw = __float_as_uint(pv.w) & 0x7fffffffu;
                cm.x = max(cm.x, ax); cm.y = max(cm.y, ay); cm.z = max(cm.z, az); cm.w = max(cm.w, aw);
                m = max(max(ax, ay), max(az, aw));
            }
            m = sgd_wave_max(m);
            if (lead) atomicMax(row_amax + r, m);
        }
        if (cin) {
            atomicMax(col_amax + c, cm.x); atomicMax(col_amax + c + 1, cm.y);
            atomicMax(col_amax + c + 2, cm.z); atomicMax(col_amax + c + 3, cm.w);
        }
        continue;
    }
    float* __restrict__ p = reinterpret_cast<float*>(t.p) + ch.offset;
    const float* __restrict__ g = reinterpret_cast<const float*>(t.g) + ch.offset;
    float* __restrict__ b = reinterpret_cast<float*>(t.buf) + ch.offset;
    const int n = min(ch.n, (int)(t.n - ch.offset));
    // 16-byte accesses need the three chunk bases aligned (chunk offsets are multiples of 4 elements)
    const bool aligned = ((t.p | t.g | t.buf) & 15) == 0;
    const int n4 = aligned ? (n & ~3) : 0;
    for (int i = threadIdx.x * 4; i < n4; i += 256 * 4) {
        float4 pv = *reinterpret_cast<const float4*>(p + i);
        const float4 gv = *reinterpret_cast<const float4*>(g + i);
        float4 bv = *reinterpret_cast<const float4*>(b + i);
        sgd_update4(pv, gv, bv, momentum, lr, wd);
        *reinterpret_cast<float4*>(b + i) = bv;
        *reinterpret_cast<float4*>(p + i) = pv;
    }
    for (int i = n4 + threadIdx.x; i < n; i += 256) {      // unaligned tensors / tails
        const float bv = fmaf(momentum, b[i], fmaf(wd, p[i], g[i]));
        b[i] = bv;
        p[i] = fmaf(-lr, bv, p[i]);
    }
  }
}

}  // namespace

extern "C" int cim_sgd_multi(const cim_sgd_tensor* tensors, const cim_sgd_chunk* chunks, int n_chunks, float momentum,
                             int max_workgroups, void* stream) {
    CIM_CHECK_ARG((tensors != nullptr && chunks != nullptr) || n_chunks == 0);
    CIM_CHECK_ARG(n_chunks >= 0 && max_workgroups >= 0);
    if (n_chunks == 0) return 0;
    const int grid = max_workgroups > 0 && max_workgroups < n_chunks ? max_workgroups : n_chunks;
    hipLaunchKernelGGL(sgd_multi_kernel, dim3(grid), dim3(256), 0, cim::as_stream(stream), tensors, chunks, n_chunks, momentum);
    CIM_CHECK_LAUNCH();
    return 0;
}
