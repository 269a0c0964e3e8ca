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
#include <cstdlib>
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

// Forward, tap-table form (default for C % 4 == 0, P <= FW_MAXP and sampling grids <= FW_MAXG):
// the sample positions of a block's bin row are wave-uniform, so their bilinear taps (pixel offsets
// and weights - ~40 instructions incl. an IEEE division per sample in the kernel above) are computed
// ONCE per block into LDS and read back as broadcasts, and the loads of two x-samples (8 x 16 B per
// lane) are issued before either is consumed.  Same operations in the same order as the direct
// kernel and the oracle: results are bit-identical.
constexpr int FW_MAXG = 32, FW_MAXP = 8;

template <bool MASKCAT>
__global__ __launch_bounds__(256) void roi_align_fwd_tab_kernel(const float* __restrict__ feat,
                                                                const float* __restrict__ rois,
                                                                const float* __restrict__ masks,
                                                                float* __restrict__ out, int C, int H, int W, int P,
                                                                float scale, int sampling_ratio, int aligned) {
    __shared__ int4 ytab[FW_MAXG];
    __shared__ int4 xtab[FW_MAXP * FW_MAXG];
    const int k = blockIdx.x, ph = blockIdx.y, tid = threadIdx.x;
    const RoiGeom g = roi_geom(rois + 5 * (size_t)k, scale, P, sampling_ratio, aligned);
    const float* __restrict__ fb = feat + (size_t)g.b * H * W * C;
    const int OC = MASKCAT ? 2 * C : C;
    if (g.gh > FW_MAXG || g.gw > FW_MAXG) {      // block-uniform: oversized sampling grid -> direct evaluation
        for (int c = tid * 4; c < C; c += 256 * 4) {
            for (int pw = 0; pw < P; ++pw) {
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int iy = 0; iy < g.gh; ++iy) {
                    const float y = g.y1 + ph * g.bh + (iy + 0.5f) * g.bh / (float)g.gh;
                    const Tap ty = make_tap(y, H);
                    for (int ix = 0; ix < g.gw; ++ix) {
                        const float x = g.x1 + pw * g.bw + (ix + 0.5f) * g.bw / (float)g.gw;
                        const Tap tx = make_tap(x, W);
                        if (!(ty.valid && tx.valid)) continue;
                        const float4 v1 = *reinterpret_cast<const float4*>(fb + ((size_t)ty.lo * W + tx.lo) * C + c);
                        const float4 v2 = *reinterpret_cast<const float4*>(fb + ((size_t)ty.lo * W + tx.hi) * C + c);
                        const float4 v3 = *reinterpret_cast<const float4*>(fb + ((size_t)ty.hi * W + tx.lo) * C + c);
                        const float4 v4 = *reinterpret_cast<const float4*>(fb + ((size_t)ty.hi * W + tx.hi) * C + c);
                        const float w1 = ty.h * tx.h, w2 = ty.h * tx.l, w3 = ty.l * tx.h, w4 = ty.l * tx.l;
                        acc = vadd(acc, vadd(vadd(vadd(vmul(w1, v1), vmul(w2, v2)), vmul(w3, v3)), vmul(w4, v4)));
                    }
                }
                const float4 o = vdiv(acc, g.count);
                float* dst = out + (((size_t)k * P + ph) * P + pw) * OC + c;
                *reinterpret_cast<float4*>(dst) = o;
                if (MASKCAT) *reinterpret_cast<float4*>(dst + C) = vmul(masks[((size_t)k * P + ph) * P + pw], o);
            }
        }
        return;
    }
    // taps: {element offset of the low pixel (-1: sample out of range), of the high pixel, l, h}
    if (tid < g.gh) {
        const float y = g.y1 + ph * g.bh + (tid + 0.5f) * g.bh / (float)g.gh;
        const Tap t = make_tap(y, H);
        ytab[tid] = make_int4(t.valid ? t.lo * W * C : -1, t.hi * W * C, __float_as_int(t.l), __float_as_int(t.h));
    }
    for (int e = tid; e < P * g.gw; e += 256) {
        const int pw = e / g.gw, ix = e - pw * g.gw;
        const float x = g.x1 + pw * g.bw + (ix + 0.5f) * g.bw / (float)g.gw;
        const Tap t = make_tap(x, W);
        xtab[pw * FW_MAXG + ix] = make_int4(t.valid ? t.lo * C : -1, t.hi * C, __float_as_int(t.l), __float_as_int(t.h));
    }
    __syncthreads();
#define FW_SAMPLE(TX, V1, V2, V3, V4)                                                                         \
    {                                                                                                         \
        const float xl = __int_as_float(TX.z), xh = __int_as_float(TX.w);                                     \
        const float w1 = yh * xh, w2 = yh * xl, w3 = yl * xh, w4 = yl * xl;                                   \
        acc = vadd(acc, vadd(vadd(vadd(vmul(w1, V1), vmul(w2, V2)), vmul(w3, V3)), vmul(w4, V4)));            \
    }
    for (int c = tid * 4; c < C; c += 256 * 4) {
        const float* __restrict__ fc = fb + c;
        for (int pw = 0; pw < P; ++pw) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            const int4* xt = xtab + pw * FW_MAXG;
            for (int iy = 0; iy < g.gh; ++iy) {
                const int4 ty = ytab[iy];
                if (ty.x < 0) continue;                                   // wave-uniform
                const float yl = __int_as_float(ty.z), yh = __int_as_float(ty.w);
                const float* __restrict__ r0 = fc + ty.x;
                const float* __restrict__ r1 = fc + ty.y;
                int ix = 0;
                for (; ix + 1 < g.gw; ix += 2) {
                    const int4 ta = xt[ix], tb = xt[ix + 1];
                    if (ta.x >= 0 && tb.x >= 0) {                         // both loads sets in flight together
                        const float4 a1 = *reinterpret_cast<const float4*>(r0 + ta.x), a2 = *reinterpret_cast<const float4*>(r0 + ta.y);
                        const float4 a3 = *reinterpret_cast<const float4*>(r1 + ta.x), a4 = *reinterpret_cast<const float4*>(r1 + ta.y);
                        const float4 b1 = *reinterpret_cast<const float4*>(r0 + tb.x), b2 = *reinterpret_cast<const float4*>(r0 + tb.y);
                        const float4 b3 = *reinterpret_cast<const float4*>(r1 + tb.x), b4 = *reinterpret_cast<const float4*>(r1 + tb.y);
                        FW_SAMPLE(ta, a1, a2, a3, a4)
                        FW_SAMPLE(tb, b1, b2, b3, b4)
                    } else {
                        if (ta.x >= 0) {
                            const float4 a1 = *reinterpret_cast<const float4*>(r0 + ta.x), a2 = *reinterpret_cast<const float4*>(r0 + ta.y);
                            const float4 a3 = *reinterpret_cast<const float4*>(r1 + ta.x), a4 = *reinterpret_cast<const float4*>(r1 + ta.y);
                            FW_SAMPLE(ta, a1, a2, a3, a4)
                        }
                        if (tb.x >= 0) {
                            const float4 b1 = *reinterpret_cast<const float4*>(r0 + tb.x), b2 = *reinterpret_cast<const float4*>(r0 + tb.y);
                            const float4 b3 = *reinterpret_cast<const float4*>(r1 + tb.x), b4 = *reinterpret_cast<const float4*>(r1 + tb.y);
                            FW_SAMPLE(tb, b1, b2, b3, b4)
                        }
                    }
                }
                if (ix < g.gw) {
                    const int4 ta = xt[ix];
                    if (ta.x >= 0) {
                        const float4 a1 = *reinterpret_cast<const float4*>(r0 + ta.x), a2 = *reinterpret_cast<const float4*>(r0 + ta.y);
                        const float4 a3 = *reinterpret_cast<const float4*>(r1 + ta.x), a4 = *reinterpret_cast<const float4*>(r1 + ta.y);
                        FW_SAMPLE(ta, a1, a2, a3, a4)
                    }
                }
            }
            const float4 o = vdiv(acc, g.count);
            float* dst = out + (((size_t)k * P + ph) * P + pw) * OC + c;
            *reinterpret_cast<float4*>(dst) = o;
            if (MASKCAT) *reinterpret_cast<float4*>(dst + C) = vmul(masks[((size_t)k * P + ph) * P + pw], o);
        }
    }
#undef FW_SAMPLE
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


// ---------------------------------------------------------------------------------------------
// Backward, LDS-resident formulation (the default).
//
// ROIAlign is linear and separable: with per-ROI weight tables
//   WY[ph][y] = sum over the bin-row's y-samples of the bilinear weight that sample puts on row y
//   WX[pw][x] = likewise for columns
// (a sample is dropped when either coordinate is out of range, which factorises too),
//   grad_in[y,x,c] += (1/count) * sum_ph WY[ph][y] * sum_pw WX[pw][x] * g[ph,pw,c].
// A workgroup OWNS a chunk of CH channels of the WHOLE feature map as an fp32 tile in LDS
// (33x43x16x4 B = 91 KB of the CU's 160 KB), walks its share of the ROIs, accumulates into the
// tile with plain LDS read-modify-writes (every (pixel, channel) is touched by exactly one lane
// per ROI), and flushes the tile once.  HBM traffic = grad_out read once + the map written
// once per ROI group; no global atomics in the ROI loop (the v1 kernel issued ~1.8 G of them).
// grid = (C/CH, RG); block = 256.
constexpr int TILE_THREADS = 1024;   // 16 waves: the tile pins one workgroup per CU, so hide LDS latency inside it

// conservative [lo, hi] range of bins whose samples can touch position `pos` along one axis
__device__ __forceinline__ void bin_range(float start, float bin, int P, int pos, int& lo, int& hi) {
    if (bin > 0.0f) {
        const float inv = 1.0f / bin;
        lo = (int)floorf(((float)pos - 1.0f - start) * inv) - 1;
        hi = (int)floorf(((float)pos + 1.0f - start) * inv) + 1;
        lo = max(lo, 0);
        hi = min(hi, P - 1);
    } else {  // degenerate ROI: every bin sits at `start`
        lo = 0;
        hi = P - 1;
    }
}

// Per-ROI record written once per launch by roi_tables_kernel (when the caller provides a workspace):
//   wy [P][H] | wx [P][W] | yr [H] | xr [W] | ylo yhi xlo xhi | count | batch      (4-byte words)
// yr / xr: packed range of bins with a non-zero weight on that row / column: lo | hi << 8 | none << 16.
__host__ __device__ __forceinline__ int roi_rec_words(int P, int H, int W) { return ((P + 1) * (H + W) + 6 + 3) & ~3; }

// The record is assembled in LDS (tables, packed ranges, bounding box through LDS atomics) and written to global memory
// once, as one contiguous run: the first version went through global memory between its three phases (three dependent
// round trips per workgroup: 16.7 us for 1000 ROIs; now ~5).  Dynamic LDS: roi_rec_words(P, H, W) words.
__global__ __launch_bounds__(256) void roi_tables_kernel(const float* __restrict__ rois, float* __restrict__ rec_all, int K,
                                                         int P, int H, int W, float scale, int sampling_ratio, int aligned) {
    extern __shared__ __attribute__((aligned(16))) float tsm[];
    const int k = blockIdx.x, tid = threadIdx.x;
    const RoiGeom g = roi_geom(rois + 5 * (size_t)k, scale, P, sampling_ratio, aligned);
    const int words = roi_rec_words(P, H, W);
    float* wy = tsm;
    float* wx = wy + P * H;
    int* yr = reinterpret_cast<int*>(wx + P * W);
    int* xr = yr + H;
    int* box = xr + W;
    if (tid < 4) box[tid] = (tid == 0) ? H : (tid == 2) ? W : -1;      // ylo yhi xlo xhi
    for (int e = tid; e < P * (H + W); e += 256) {
        const bool isy = e < P * H;
        const int e2 = isy ? e : e - P * H;
        const int size = isy ? H : W;
        const int pb = e2 / size, pos = e2 % size;
        const int gn = isy ? g.gh : g.gw;
        const float start = isy ? g.y1 : g.x1, bin = isy ? g.bh : g.bw;
        float acc = 0.0f;
        for (int is = 0; is < gn; ++is) {
            const float v = start + pb * bin + (is + 0.5f) * bin / (float)gn;
            const Tap t = make_tap(v, size);
            if (!t.valid) continue;
            if (t.lo == pos) acc += t.h;
            if (t.hi == pos) acc += t.l;
        }
        (isy ? wy : wx)[e2] = acc;
    }
    __syncthreads();
    for (int e = tid; e < H + W; e += 256) {
        const bool isy = e < H;
        const int pos = isy ? e : e - H, size = isy ? H : W;
        const float* tabp = isy ? wy : wx;
        int lo = P, hi = -1;
        for (int pb = 0; pb < P; ++pb)
            if (tabp[pb * size + pos] != 0.0f) { lo = min(lo, pb); hi = pb; }
        (isy ? yr : xr)[pos] = (hi < 0) ? 0x10000 : (lo | (hi << 8));
        if (hi >= 0) {
            atomicMin(&box[isy ? 0 : 2], pos);
            atomicMax(&box[isy ? 1 : 3], pos);
        }
    }
    if (tid == 0) {
        reinterpret_cast<float*>(box)[4] = g.count;
        box[5] = g.b;
    }
    __syncthreads();
    float* rec = rec_all + (size_t)k * words;
    for (int e = tid; e < (P + 1) * (H + W) + 6; e += 256) rec[e] = tsm[e];
}

// ---------------------------------------------------------------------------------------------
// Forward, aggregated-weight form (default when the caller provides the table workspace).
//
// out[ph,pw,c] = (1/count) sum_y sum_x WY[ph][y] WX[pw][x] feat[y,x,c]   with the per-ROI tables of
// roi_tables_kernel (the same ones the backward uses).  A bin touches (rows of bin ph) x (columns of bin pw)
// pixels ONCE each - (gh+1)(gw+1) 16 B loads per lane instead of the 4*gh*gw of the sample-by-sample kernels
// (2.5-3x fewer at the benchmark's ROI sizes: the tap kernels move 16x the algorithmic bytes through L1).
// The sum is reassociated, so the result differs from the oracle's sample order by a few ulp (tests state
// 1e-6 relative); the sample-order kernels above stay behind cim_roi_align_fwd / CIM_ROI_FWD_EXACT=1.
// The weight tables are built with FP contraction off (same values as the backward's).
// A workgroup owns one (roi, bin row): it flattens the bin row's (pixel offset, weight) pairs into LDS once,
// then every lane runs a flat, 4-way unrolled loop over them for its 4 channels.
typedef float ga_f2 __attribute__((ext_vector_type(2)));       // v_pk_fma_f32: two fp32 FMAs per instruction
__device__ __forceinline__ ga_f2 ga_lo(const float4& v) { return ga_f2{v.x, v.y}; }
__device__ __forceinline__ ga_f2 ga_hi(const float4& v) { return ga_f2{v.z, v.w}; }
__device__ __forceinline__ ga_f2 ga_fma(float w, ga_f2 v, ga_f2 a) { return __builtin_elementwise_fma(ga_f2{w, w}, v, a); }
#ifndef CIM_ROI_FU
#define CIM_ROI_FU 8             // loads in flight per lane in the aggregated forward (4 or 8)
#endif
#ifndef CIM_ROI_FEXP
#define CIM_ROI_FEXP 0           // ablations: 1 = no stores (0.131 ms incl. tables), 2 = no loads (0.100); product 0.157
#endif
#ifndef CIM_ROI_FZ
#define CIM_ROI_FZ 1             // channel slices of the aggregated forward (grid.z); 2 / 4 (L2-sized slices) measured equal
#endif
#ifndef CIM_ROI_FNT
#define CIM_ROI_FNT 1            // 1 = nontemporal stores of the pooled output
#endif
constexpr int AG_MAXE = 64;     // entries per (ph, pw) list kept in LDS; larger bins take the sample-order kernel

template <bool MASKCAT>
__global__ __launch_bounds__(256) void roi_align_fwd_agg_kernel(const float* __restrict__ feat,
                                                                const float* __restrict__ masks,
                                                                float* __restrict__ out, int C, int H, int W, int P,
                                                                const float* __restrict__ rec_all,
                                                                const float* __restrict__ rois, float scale,
                                                                int sampling_ratio, int aligned) {
    __shared__ int2 ent[FW_MAXP * AG_MAXE];      // {element offset of the pixel, weight bits}
    __shared__ int s_over;
    __shared__ int s_n[FW_MAXP], s_xlo[FW_MAXP], s_nx[FW_MAXP];
    __shared__ int s_rows[64];                   // rows with a non-zero weight in this bin row
    __shared__ int s_nrows;
    // grid.z splits the channels (CIM_ROI_FZ slices): with 2 slices one slice of the cfg2 map (2.9 MB) fits an XCD's 4 MB L2
    const int k = blockIdx.x, ph = blockIdx.y, tid = threadIdx.x, NTH = blockDim.x;
    const int cslice = ((C / 4 + gridDim.z - 1) / gridDim.z) * 4;
    const int c_first = blockIdx.z * cslice + tid * 4, c_end = min(C, (int)(blockIdx.z + 1) * cslice);
    const float* rec = rec_all + (size_t)k * roi_rec_words(P, H, W);
    const float* wy = rec + ph * H;
    const float* wx = rec + P * H;
    const int* box = reinterpret_cast<const int*>(rec + (P + 1) * (H + W));
    const int ylo = box[0], yhi = box[1], xlo = box[2], xhi = box[3];
    const float inv_count = 1.0f / reinterpret_cast<const float*>(box)[4];
    const float* __restrict__ fb = feat + (size_t)box[5] * H * W * C;
    if (tid == 0) {
        int n = 0;
        for (int y = ylo; y <= yhi; ++y)
            if (wy[y] != 0.0f && n < 64) s_rows[n++] = y;
        s_nrows = n;
        s_over = 0;
    } else if (tid >= 64 && tid < 64 + P) {
        const int pw = tid - 64;
        int lo = xhi + 1, hi = xlo - 1;
        for (int x = xlo; x <= xhi; ++x)
            if (wx[pw * W + x] != 0.0f) { lo = min(lo, x); hi = x; }
        s_xlo[pw] = lo;
        s_nx[pw] = max(hi - lo + 1, 0);
    }
    __syncthreads();
    const int nrows = s_nrows;
    if (tid < P && nrows * s_nx[tid] > AG_MAXE) s_over = 1;      // benign race: every writer stores 1
    __syncthreads();
    const int OC = MASKCAT ? 2 * C : C;
    if (s_over) {      // block-uniform: a bin larger than the LDS list (ROI far larger than the map) -> sample by sample
        const RoiGeom g = roi_geom(rois + 5 * (size_t)k, scale, P, sampling_ratio, aligned);
        for (int c = c_first; c < c_end; c += NTH * 4) {
            for (int pw = 0; pw < P; ++pw) {
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int iy = 0; iy < g.gh; ++iy) {
                    const float y = g.y1 + ph * g.bh + (iy + 0.5f) * g.bh / (float)g.gh;
                    const Tap ty = make_tap(y, H);
                    for (int ix = 0; ix < g.gw; ++ix) {
                        const float x = g.x1 + pw * g.bw + (ix + 0.5f) * g.bw / (float)g.gw;
                        const Tap tx = make_tap(x, W);
                        if (!(ty.valid && tx.valid)) continue;
                        const float4 v1 = *reinterpret_cast<const float4*>(fb + ((size_t)ty.lo * W + tx.lo) * C + c);
                        const float4 v2 = *reinterpret_cast<const float4*>(fb + ((size_t)ty.lo * W + tx.hi) * C + c);
                        const float4 v3 = *reinterpret_cast<const float4*>(fb + ((size_t)ty.hi * W + tx.lo) * C + c);
                        const float4 v4 = *reinterpret_cast<const float4*>(fb + ((size_t)ty.hi * W + tx.hi) * C + c);
                        const float w1 = ty.h * tx.h, w2 = ty.h * tx.l, w3 = ty.l * tx.h, w4 = ty.l * tx.l;
                        acc = vadd(acc, vadd(vadd(vadd(vmul(w1, v1), vmul(w2, v2)), vmul(w3, v3)), vmul(w4, v4)));
                    }
                }
                const float4 o = vdiv(acc, g.count);
                float* dst = out + (((size_t)k * P + ph) * P + pw) * OC + c;
                *reinterpret_cast<float4*>(dst) = o;
                if (MASKCAT) *reinterpret_cast<float4*>(dst + C) = vmul(masks[((size_t)k * P + ph) * P + pw], o);
            }
        }
        return;
    }
    for (int e = tid; e < P * AG_MAXE; e += NTH) {
        const int pw = e / AG_MAXE, i = e % AG_MAXE;
        const int nx = s_nx[pw];
        const int n = nrows * nx;
        if (i == 0) s_n[pw] = n;
        if (i < n) {
            const int r = i / nx, x = s_xlo[pw] + (i - r * nx), y = s_rows[r];
            ent[e] = make_int2((y * W + x) * C, __float_as_int(wy[y] * inv_count * wx[pw * W + x]));
        }
    }
    __syncthreads();
    for (int c = c_first; c < c_end; c += NTH * 4) {
        const float* __restrict__ fc = fb + c;
        for (int pw = 0; pw < P; ++pw) {
            const int2* el = ent + pw * AG_MAXE;
#if CIM_ROI_FEXP == 2
            const int n = 0;
#else
            const int n = s_n[pw];
#endif
            ga_f2 al = {0.f, 0.f}, ah = {0.f, 0.f};
            int i = 0;
#if CIM_ROI_FU == 8
            for (; i + 8 <= n; i += 8) {
                int2 e[8];
                float4 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) e[j] = el[i + j];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const float4*>(fc + e[j].x);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float w = __int_as_float(e[j].y);
                    al = ga_fma(w, ga_lo(v[j]), al);
                    ah = ga_fma(w, ga_hi(v[j]), ah);
                }
            }
#endif
            for (; i + 4 <= n; i += 4) {
                const int2 e0 = el[i], e1 = el[i + 1], e2 = el[i + 2], e3 = el[i + 3];
                const float4 v0 = *reinterpret_cast<const float4*>(fc + e0.x);
                const float4 v1 = *reinterpret_cast<const float4*>(fc + e1.x);
                const float4 v2 = *reinterpret_cast<const float4*>(fc + e2.x);
                const float4 v3 = *reinterpret_cast<const float4*>(fc + e3.x);
                const float w0 = __int_as_float(e0.y), w1 = __int_as_float(e1.y), w2 = __int_as_float(e2.y), w3 = __int_as_float(e3.y);
                al = ga_fma(w0, ga_lo(v0), al); ah = ga_fma(w0, ga_hi(v0), ah);
                al = ga_fma(w1, ga_lo(v1), al); ah = ga_fma(w1, ga_hi(v1), ah);
                al = ga_fma(w2, ga_lo(v2), al); ah = ga_fma(w2, ga_hi(v2), ah);
                al = ga_fma(w3, ga_lo(v3), al); ah = ga_fma(w3, ga_hi(v3), ah);
            }
            for (; i < n; ++i) {
                const int2 e0 = el[i];
                const float4 v0 = *reinterpret_cast<const float4*>(fc + e0.x);
                const float w0 = __int_as_float(e0.y);
                al = ga_fma(w0, ga_lo(v0), al); ah = ga_fma(w0, ga_hi(v0), ah);
            }
            const float4 acc = make_float4(al.x, al.y, ah.x, ah.y);
            float* dst = out + (((size_t)k * P + ph) * P + pw) * OC + c;
#if CIM_ROI_FEXP == 1      /* ablation: no stores */
            if (acc.x != 123.456f) continue;
#endif
#if CIM_ROI_FEXP == 2      /* ablation: no loads (stores only) */
#endif
#if CIM_ROI_FNT
            typedef float ga_f4 __attribute__((ext_vector_type(4)));
            __builtin_nontemporal_store(ga_f4{acc.x, acc.y, acc.z, acc.w}, reinterpret_cast<ga_f4*>(dst));
            if (MASKCAT) {
                const float4 mo = vmul(masks[((size_t)k * P + ph) * P + pw], acc);
                __builtin_nontemporal_store(ga_f4{mo.x, mo.y, mo.z, mo.w}, reinterpret_cast<ga_f4*>(dst + C));
            }
#else
            *reinterpret_cast<float4*>(dst) = acc;
            if (MASKCAT) *reinterpret_cast<float4*>(dst + C) = vmul(masks[((size_t)k * P + ph) * P + pw], acc);
#endif
        }
    }
}

// Forward, row-sum form of the same separable sum (default when P <= 7 and the map is at most 64 x 64):
//   out[ph,pw,c] = sum_x WX[pw][x] * ( sum_y WY[ph][y] / count * feat[y,x,c] ).
// A workgroup still owns one (roi, bin row), but walks the COLUMNS of the ROI once: the inner sum t(x) over the bin row's
// rows is formed once per column and fed to every bin whose WX[pw][x] is non-zero, so the pixels that neighbouring bins
// share (one or two columns per bin boundary) are loaded once - rows x (columns of the ROI) loads instead of
// rows x (sum of the bins' column counts), ~23 % fewer at the benchmark's ROI sizes.  Two columns are in flight per
// iteration (up to 8 loads per lane).  The 7 column weights of a column sit in LDS as one padded row (zero outside the
// bin's range), so the bin update is 7 unconditional packed FMAs.
#ifndef CIM_ROI_FROW
#define CIM_ROI_FROW 1           // 0 = the flat entry-list kernel above
#endif
constexpr int RS_MAXD = 64;      // rows / columns of the map
template <bool MASKCAT>
__global__ __launch_bounds__(256) void roi_align_fwd_rowsum_kernel(const float* __restrict__ feat,
                                                                   const float* __restrict__ masks,
                                                                   float* __restrict__ out, int C, int H, int W, int P,
                                                                   const float* __restrict__ rec_all) {
    __shared__ __attribute__((aligned(16))) float s_wx[RS_MAXD][8];     // [column - xlo][pw], pw = 7 is padding
    __shared__ float s_wy[RS_MAXD];
    __shared__ int s_rows[RS_MAXD];
    __shared__ int s_nrows;
    const int k = blockIdx.x, ph = blockIdx.y, tid = threadIdx.x, NTH = blockDim.x;      // NTH = min(256, C / 4 rounded up to a wave)
    const float* rec = rec_all + (size_t)k * roi_rec_words(P, H, W);
    const float* wy = rec + ph * H;
    const float* wx = rec + P * H;
    const int* box = reinterpret_cast<const int*>(rec + (P + 1) * (H + W));
    const int ylo = box[0], yhi = box[1], xlo = box[2], xhi = box[3];
    const float inv_count = 1.0f / reinterpret_cast<const float*>(box)[4];
    const float* __restrict__ fb = feat + (size_t)box[5] * H * W * C;
    const int ncols = max(xhi - xlo + 1, 0);
    if (tid == 0) {
        int n = 0;
        for (int y = ylo; y <= yhi; ++y) {
            const float w = wy[y];
            if (w != 0.0f) { s_rows[n] = y * W; s_wy[n] = w * inv_count; ++n; }
        }
        s_nrows = n;
    }
    for (int e = tid; e < ncols * 8; e += NTH) {
        const int xi = e >> 3, pw = e & 7;
        s_wx[xi][pw] = pw < P ? wx[pw * W + xlo + xi] : 0.0f;
    }
    __syncthreads();
    const int nrows = s_nrows;
    const int OC = MASKCAT ? 2 * C : C;
    for (int c = tid * 4; c < C; c += NTH * 4) {
        const float* __restrict__ fc = fb + c;
        ga_f2 al[7], ah[7];
#pragma unroll
        for (int pw = 0; pw < 7; ++pw) { al[pw] = ga_f2{0.f, 0.f}; ah[pw] = ga_f2{0.f, 0.f}; }
        int xi = 0;
        for (; xi + 2 <= ncols; xi += 2) {
            const int x0 = (xlo + xi) * C;
            ga_f2 t0l = {0.f, 0.f}, t0h = {0.f, 0.f}, t1l = {0.f, 0.f}, t1h = {0.f, 0.f};
            int r = 0;
            for (; r + 4 <= nrows; r += 4) {
                float4 v0[4], v1[4];
                float w[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float* src = fc + (size_t)s_rows[r + j] * C + x0;
                    v0[j] = *reinterpret_cast<const float4*>(src);
                    v1[j] = *reinterpret_cast<const float4*>(src + C);
                    w[j] = s_wy[r + j];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    t0l = ga_fma(w[j], ga_lo(v0[j]), t0l); t0h = ga_fma(w[j], ga_hi(v0[j]), t0h);
                    t1l = ga_fma(w[j], ga_lo(v1[j]), t1l); t1h = ga_fma(w[j], ga_hi(v1[j]), t1h);
                }
            }
            for (; r < nrows; ++r) {
                const float* src = fc + (size_t)s_rows[r] * C + x0;
                const float4 v0 = *reinterpret_cast<const float4*>(src);
                const float4 v1 = *reinterpret_cast<const float4*>(src + C);
                const float w = s_wy[r];
                t0l = ga_fma(w, ga_lo(v0), t0l); t0h = ga_fma(w, ga_hi(v0), t0h);
                t1l = ga_fma(w, ga_lo(v1), t1l); t1h = ga_fma(w, ga_hi(v1), t1h);
            }
            const float4 wa0 = *reinterpret_cast<const float4*>(&s_wx[xi][0]), wb0 = *reinterpret_cast<const float4*>(&s_wx[xi][4]);
            const float4 wa1 = *reinterpret_cast<const float4*>(&s_wx[xi + 1][0]), wb1 = *reinterpret_cast<const float4*>(&s_wx[xi + 1][4]);
            const float w0[7] = {wa0.x, wa0.y, wa0.z, wa0.w, wb0.x, wb0.y, wb0.z};
            const float w1[7] = {wa1.x, wa1.y, wa1.z, wa1.w, wb1.x, wb1.y, wb1.z};
#pragma unroll
            for (int pw = 0; pw < 7; ++pw) {
                al[pw] = ga_fma(w0[pw], t0l, al[pw]); ah[pw] = ga_fma(w0[pw], t0h, ah[pw]);
                al[pw] = ga_fma(w1[pw], t1l, al[pw]); ah[pw] = ga_fma(w1[pw], t1h, ah[pw]);
            }
        }
        if (xi < ncols) {
            const int x0 = (xlo + xi) * C;
            ga_f2 t0l = {0.f, 0.f}, t0h = {0.f, 0.f};
            for (int r = 0; r < nrows; ++r) {
                const float4 v0 = *reinterpret_cast<const float4*>(fc + (size_t)s_rows[r] * C + x0);
                const float w = s_wy[r];
                t0l = ga_fma(w, ga_lo(v0), t0l); t0h = ga_fma(w, ga_hi(v0), t0h);
            }
            const float4 wa0 = *reinterpret_cast<const float4*>(&s_wx[xi][0]), wb0 = *reinterpret_cast<const float4*>(&s_wx[xi][4]);
            const float w0[7] = {wa0.x, wa0.y, wa0.z, wa0.w, wb0.x, wb0.y, wb0.z};
#pragma unroll
            for (int pw = 0; pw < 7; ++pw) { al[pw] = ga_fma(w0[pw], t0l, al[pw]); ah[pw] = ga_fma(w0[pw], t0h, ah[pw]); }
        }
#pragma unroll
        for (int pw = 0; pw < 7; ++pw) {
            if (pw >= P) break;
            typedef float ga_f4 __attribute__((ext_vector_type(4)));
            float* dst = out + (((size_t)k * P + ph) * P + pw) * OC + c;
            __builtin_nontemporal_store(ga_f4{al[pw].x, al[pw].y, ah[pw].x, ah[pw].y}, reinterpret_cast<ga_f4*>(dst));
            if (MASKCAT) {
                const float m = masks[((size_t)k * P + ph) * P + pw];
                __builtin_nontemporal_store(ga_f4{m * al[pw].x, m * al[pw].y, m * ah[pw].x, m * ah[pw].y}, reinterpret_cast<ga_f4*>(dst + C));
            }
        }
    }
}

// Two bin rows per workgroup (CIM_ROI_FROW2): the rows of the map that bin rows ph and ph + 1 share (one or two of ~3.6)
// are loaded once as well; grid = (K, ceil(P / 2)).
#ifndef CIM_ROI_FROW2
#define CIM_ROI_FROW2 1           // 0.121 -> 0.118 ms per call at cfg2, bit-identical to the one-row kernel
#endif
template <bool MASKCAT>
__global__ __launch_bounds__(256) void roi_align_fwd_rowsum2_kernel(const float* __restrict__ feat,
                                                                    const float* __restrict__ masks,
                                                                    float* __restrict__ out, int C, int H, int W, int P,
                                                                    const float* __restrict__ rec_all) {
    __shared__ __attribute__((aligned(16))) float s_wx[RS_MAXD][8];
    __shared__ float s_wa[RS_MAXD], s_wb[RS_MAXD];
    __shared__ int s_rows[RS_MAXD];
    __shared__ int s_nrows;
    const int k = blockIdx.x, ph0 = blockIdx.y * 2, tid = threadIdx.x, NTH = blockDim.x;
    const bool two = ph0 + 1 < P;
    const float* rec = rec_all + (size_t)k * roi_rec_words(P, H, W);
    const float* wy0 = rec + ph0 * H;
    const float* wy1 = wy0 + H;
    const float* wx = rec + P * H;
    const int* box = reinterpret_cast<const int*>(rec + (P + 1) * (H + W));
    const int ylo = box[0], yhi = box[1], xlo = box[2], xhi = box[3];
    const float inv_count = 1.0f / reinterpret_cast<const float*>(box)[4];
    const float* __restrict__ fb = feat + (size_t)box[5] * H * W * C;
    const int ncols = max(xhi - xlo + 1, 0);
    if (tid == 0) {
        int n = 0;
        for (int y = ylo; y <= yhi; ++y) {
            const float a = wy0[y], b = two ? wy1[y] : 0.0f;
            if (a != 0.0f || b != 0.0f) { s_rows[n] = y * W; s_wa[n] = a * inv_count; s_wb[n] = b * inv_count; ++n; }
        }
        s_nrows = n;
    }
    for (int e = tid; e < ncols * 8; e += NTH) {
        const int xi = e >> 3, pw = e & 7;
        s_wx[xi][pw] = pw < P ? wx[pw * W + xlo + xi] : 0.0f;
    }
    __syncthreads();
    const int nrows = s_nrows;
    const int OC = MASKCAT ? 2 * C : C;
    for (int c = tid * 4; c < C; c += NTH * 4) {
        const float* __restrict__ fc = fb + c;
        ga_f2 al[7], ah[7], bl[7], bh[7];
#pragma unroll
        for (int pw = 0; pw < 7; ++pw) al[pw] = ah[pw] = bl[pw] = bh[pw] = ga_f2{0.f, 0.f};
        for (int xi = 0; xi < ncols; xi += 2) {
            const bool pair = xi + 1 < ncols;            // wave-uniform; the odd last column is loaded twice, weight 0
            const int x0 = (xlo + xi) * C, dx1 = pair ? C : 0;
            ga_f2 ta0l = {0.f, 0.f}, ta0h = {0.f, 0.f}, ta1l = {0.f, 0.f}, ta1h = {0.f, 0.f};
            ga_f2 tb0l = {0.f, 0.f}, tb0h = {0.f, 0.f}, tb1l = {0.f, 0.f}, tb1h = {0.f, 0.f};
            int r = 0;
            for (; r + 4 <= nrows; r += 4) {
                float4 v0[4], v1[4];
                float wa[4], wb[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float* src = fc + (size_t)s_rows[r + j] * C + x0;
                    v0[j] = *reinterpret_cast<const float4*>(src);
                    v1[j] = *reinterpret_cast<const float4*>(src + dx1);
                    wa[j] = s_wa[r + j];
                    wb[j] = s_wb[r + j];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    ta0l = ga_fma(wa[j], ga_lo(v0[j]), ta0l); ta0h = ga_fma(wa[j], ga_hi(v0[j]), ta0h);
                    ta1l = ga_fma(wa[j], ga_lo(v1[j]), ta1l); ta1h = ga_fma(wa[j], ga_hi(v1[j]), ta1h);
                    tb0l = ga_fma(wb[j], ga_lo(v0[j]), tb0l); tb0h = ga_fma(wb[j], ga_hi(v0[j]), tb0h);
                    tb1l = ga_fma(wb[j], ga_lo(v1[j]), tb1l); tb1h = ga_fma(wb[j], ga_hi(v1[j]), tb1h);
                }
            }
            for (; r < nrows; ++r) {
                const float* src = fc + (size_t)s_rows[r] * C + x0;
                const float4 v0 = *reinterpret_cast<const float4*>(src);
                const float4 v1 = *reinterpret_cast<const float4*>(src + dx1);
                const float wa = s_wa[r], wb = s_wb[r];
                ta0l = ga_fma(wa, ga_lo(v0), ta0l); ta0h = ga_fma(wa, ga_hi(v0), ta0h);
                ta1l = ga_fma(wa, ga_lo(v1), ta1l); ta1h = ga_fma(wa, ga_hi(v1), ta1h);
                tb0l = ga_fma(wb, ga_lo(v0), tb0l); tb0h = ga_fma(wb, ga_hi(v0), tb0h);
                tb1l = ga_fma(wb, ga_lo(v1), tb1l); tb1h = ga_fma(wb, ga_hi(v1), tb1h);
            }
            const int xj = pair ? xi + 1 : xi;
            const float4 wa0 = *reinterpret_cast<const float4*>(&s_wx[xi][0]), wb0 = *reinterpret_cast<const float4*>(&s_wx[xi][4]);
            const float4 wa1 = *reinterpret_cast<const float4*>(&s_wx[xj][0]), wb1 = *reinterpret_cast<const float4*>(&s_wx[xj][4]);
            const float m1 = pair ? 1.0f : 0.0f;
            const float w0[7] = {wa0.x, wa0.y, wa0.z, wa0.w, wb0.x, wb0.y, wb0.z};
            const float w1[7] = {m1 * wa1.x, m1 * wa1.y, m1 * wa1.z, m1 * wa1.w, m1 * wb1.x, m1 * wb1.y, m1 * wb1.z};
#pragma unroll
            for (int pw = 0; pw < 7; ++pw) {
                al[pw] = ga_fma(w0[pw], ta0l, al[pw]); ah[pw] = ga_fma(w0[pw], ta0h, ah[pw]);
                al[pw] = ga_fma(w1[pw], ta1l, al[pw]); ah[pw] = ga_fma(w1[pw], ta1h, ah[pw]);
                bl[pw] = ga_fma(w0[pw], tb0l, bl[pw]); bh[pw] = ga_fma(w0[pw], tb0h, bh[pw]);
                bl[pw] = ga_fma(w1[pw], tb1l, bl[pw]); bh[pw] = ga_fma(w1[pw], tb1h, bh[pw]);
            }
        }
        typedef float ga_f4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int pw = 0; pw < 7; ++pw) {
            if (pw >= P) break;
            float* dst = out + (((size_t)k * P + ph0) * P + pw) * OC + c;
            __builtin_nontemporal_store(ga_f4{al[pw].x, al[pw].y, ah[pw].x, ah[pw].y}, reinterpret_cast<ga_f4*>(dst));
            if (MASKCAT) {
                const float m = masks[((size_t)k * P + ph0) * P + pw];
                __builtin_nontemporal_store(ga_f4{m * al[pw].x, m * al[pw].y, m * ah[pw].x, m * ah[pw].y}, reinterpret_cast<ga_f4*>(dst + C));
            }
            if (two) {
                float* dst2 = dst + (size_t)P * OC;
                __builtin_nontemporal_store(ga_f4{bl[pw].x, bl[pw].y, bh[pw].x, bh[pw].y}, reinterpret_cast<ga_f4*>(dst2));
                if (MASKCAT) {
                    const float m = masks[((size_t)k * P + ph0 + 1) * P + pw];
                    __builtin_nontemporal_store(ga_f4{m * bl[pw].x, m * bl[pw].y, m * bh[pw].x, m * bh[pw].y}, reinterpret_cast<ga_f4*>(dst2 + C));
                }
            }
        }
    }
}

template <int CH, bool MASKCAT, bool PRE>
__global__ __launch_bounds__(TILE_THREADS) void roi_align_bwd_tile_kernel(const float* __restrict__ grad_out,
                                                                          const float* __restrict__ rois,
                                                                          const float* __restrict__ masks,
                                                                          float* __restrict__ grad_in, int B, int C,
                                                                          int H, int W, int K, int P, float scale,
                                                                          int sampling_ratio, int aligned,
                                                                          int use_atomic, const float* __restrict__ pre) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int CG = CH / 4;
    constexpr int NT = TILE_THREADS;
    const int HW = H * W;
    const int tab = PRE ? (P + 1) * (H + W) : P * (H + W);     // PRE: + packed bin ranges per row / column
    const int recw = roi_rec_words(P, H, W);
    float* tile = lds;                        // [HW][CH]
    float* gbuf = tile + (size_t)HW * CH;     // 2 x [P*P][CH]   (already divided by count), double-buffered
    float* wtab = gbuf + 2 * P * P * CH;      // 2 x ([P][H] + [P][W]), double-buffered
    const int tid = threadIdx.x;
    const int c0 = blockIdx.x * CH;
    const int OC = MASKCAT ? 2 * C : C;

    for (int b = 0; b < B; ++b) {
        for (int i = tid; i < HW * CG; i += NT) reinterpret_cast<float4*>(tile)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        int buf = 0;
        for (int k = blockIdx.y; k < K; k += gridDim.y) {
            RoiGeom g;
            const float* rec = pre + (size_t)k * recw;
            const int* box = reinterpret_cast<const int*>(rec + (P + 1) * (H + W));
            if (PRE) {
                g.count = reinterpret_cast<const float*>(box)[4];
                g.b = box[5];
            } else {
                g = roi_geom(rois + 5 * (size_t)k, scale, P, sampling_ratio, aligned);
            }
            if (g.b != b) continue;                               // block-uniform
            float* gb_ = gbuf + buf * P * P * CH;
            float* wy = wtab + buf * tab;
            float* wx = wy + P * H;
            if (PRE) {      // tables + bin ranges of this ROI were computed once for all channel chunks
                for (int e = tid; e < tab; e += NT) wy[e] = rec[e];
            }
            // weight tables: one entry per lane, loop over the bin's samples
            for (int e = tid; e < (PRE ? 0 : tab); e += NT) {
                const bool isy = e < P * H;
                const int e2 = isy ? e : e - P * H;
                const int size = isy ? H : W;
                const int pb = e2 / size, pos = e2 % size;
                const int gn = isy ? g.gh : g.gw;
                const float start = isy ? g.y1 : g.x1, bin = isy ? g.bh : g.bw;
                float acc = 0.0f;
                for (int is = 0; is < gn; ++is) {
                    const float v = start + pb * bin + (is + 0.5f) * bin / (float)gn;
                    const Tap t = make_tap(v, size);
                    if (!t.valid) continue;
                    if (t.lo == pos) acc += t.h;
                    if (t.hi == pos) acc += t.l;
                }
                (isy ? wy : wx)[e2] = acc;
            }
            // gradient of this ROI for our channel chunk, pre-divided by count
            for (int e = tid; e < P * P * CG; e += NT) {
                const int bin = e / CG, cg = e % CG;
                const float* src = grad_out + ((size_t)k * P * P + bin) * OC + c0 + cg * 4;
                float4 v = *reinterpret_cast<const float4*>(src);
                if (MASKCAT) {
                    const float m = masks[(size_t)k * P * P + bin];
                    const float4 v2 = *reinterpret_cast<const float4*>(src + C);
                    v = make_float4(v.x + m * v2.x, v.y + m * v2.y, v.z + m * v2.z, v.w + m * v2.w);
                }
                reinterpret_cast<float4*>(gb_)[e] = make_float4(v.x / g.count, v.y / g.count, v.z / g.count, v.w / g.count);
            }
            // one barrier per ROI: tables/gradient of ROI k are double-buffered, and the tile
            // read-modify-writes of ROI k-1 (issued before this barrier by every lane) are complete.
            __syncthreads();
            // conservative bounding box of the pixels the ROI's samples can touch
            int ylo, yhi, xlo, xhi;
            const int* yr = reinterpret_cast<const int*>(wx + P * W);
            const int* xr = yr + H;
            if (PRE) {
                ylo = box[0]; yhi = box[1]; xlo = box[2]; xhi = box[3];
            } else {
                const float ylast = g.y1 + (float)P * g.bh, xlast = g.x1 + (float)P * g.bw;
                ylo = max(0, (int)floorf(fminf(g.y1, ylast)) - 1); yhi = min(H - 1, (int)floorf(fmaxf(g.y1, ylast)) + 2);
                xlo = max(0, (int)floorf(fminf(g.x1, xlast)) - 1); xhi = min(W - 1, (int)floorf(fmaxf(g.x1, xlast)) + 2);
            }
            if (yhi >= ylo && xhi >= xlo) {
                const int rw = xhi - xlo + 1, items = (yhi - ylo + 1) * rw * CG;
                for (int it = tid; it < items; it += NT) {
                    const int cg = it % CG, pix = it / CG;
                    const int y = ylo + pix / rw, x = xlo + pix % rw;
                    int phl, phh, pwl, pwh;
                    if (PRE) {
                        const int ry = yr[y], rx = xr[x];
                        if ((ry | rx) & 0x10000) continue;
                        phl = ry & 0xff; phh = (ry >> 8) & 0xff; pwl = rx & 0xff; pwh = (rx >> 8) & 0xff;
                    } else {
                        bin_range(g.y1, g.bh, P, y, phl, phh);
                        bin_range(g.x1, g.bw, P, x, pwl, pwh);
                    }
                    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                    for (int ph = phl; ph <= phh; ++ph) {
                        const float a = wy[ph * H + y];
                        if (!PRE && a == 0.0f) continue;
                        float4 row = make_float4(0.f, 0.f, 0.f, 0.f);
                        for (int pw = pwl; pw <= pwh; ++pw) {
                            const float bw_ = wx[pw * W + x];
                            const float4 gv = reinterpret_cast<const float4*>(gb_)[(ph * P + pw) * CG + cg];
                            row.x += bw_ * gv.x; row.y += bw_ * gv.y; row.z += bw_ * gv.z; row.w += bw_ * gv.w;
                        }
                        acc.x += a * row.x; acc.y += a * row.y; acc.z += a * row.z; acc.w += a * row.w;
                    }
                    float4* t4 = reinterpret_cast<float4*>(tile) + (size_t)(y * W + x) * CG + cg;
                    float4 cur = *t4;
                    cur.x += acc.x; cur.y += acc.y; cur.z += acc.z; cur.w += acc.w;
                    *t4 = cur;
                }
            }
            buf ^= 1;
        }
        __syncthreads();
        float* gb = grad_in + (size_t)b * HW * C;
        for (int i = tid; i < HW * CG; i += NT) {
            const int pix = i / CG, cg = i % CG;
            const float4 v = reinterpret_cast<const float4*>(tile)[i];
            float* dst = gb + (size_t)pix * C + c0 + cg * 4;
            if (use_atomic) {
                atomicAdd(dst + 0, v.x); atomicAdd(dst + 1, v.y); atomicAdd(dst + 2, v.z); atomicAdd(dst + 3, v.w);
            } else {
                *reinterpret_cast<float4*>(dst) = v;
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// Backward, gather form (default when the tables exist): the adjoint of roi_align_fwd_agg_kernel.
//
//   grad_in[y,x,c] = sum over ROIs k and bins (ph,pw) of  WY_k[ph][y] WX_k[pw][x] / count_k * g_k[ph,pw,c]
// A workgroup owns a GH x GW block of feature pixels x 1024 channels (lanes along C, 16 B per lane, GH*GW float4
// accumulators in registers) for one group of 256 ROIs: no LDS tile, no atomics inside the ROI loop; the groups'
// partial sums meet in grad_in through one atomicAdd per element (ceil(K/256) per element in total; plain stores
// when K <= 256).  Each lane inspects ONE ROI's bin ranges for the block's rows / columns (one packed table word
// per row / column); the (gradient offset, mask, GH*GW weights) entries of the bins that touch the block are laid
// out in LDS by a block-wide prefix sum (deterministic order); then all lanes stream the entry list, 4 entries
// (8 x 16 B loads with the mask-cat prologue fused) in flight per lane - one gradient load feeds GH*GW FMAs, so
// neighbouring pixels share the loads of the bins they share (a bin spans ~3.6 x 4.1 pixels at the benchmark's
// ROI sizes: a 2 x 2 block reads each gradient vector ~5.5 times instead of ~15).
#ifndef CIM_ROI_GE
#define CIM_ROI_GE 1024            // entries per LDS window
#endif
constexpr int GA_MAXE = CIM_ROI_GE;
#ifndef CIM_ROI_GS
#define CIM_ROI_GS 128             // ROIs per workgroup (<= 256: one per lane in the inspection phase)
#endif
#ifndef CIM_ROI_GU
#define CIM_ROI_GU 4               // entries in flight per lane in the streaming phase (4 or 8)
#endif
constexpr int GA_GS = CIM_ROI_GS;
#ifndef CIM_ROI_GOCC
#define CIM_ROI_GOCC 4          // waves per SIMD the register allocator must leave room for (<= 128 VGPRs)
#endif
#ifndef CIM_ROI_GEXP
#define CIM_ROI_GEXP 0          // ablations: 1 = no streaming phase, 2 = no flush
#endif

#ifndef CIM_ROI_GNT
#define CIM_ROI_GNT 0              // 1 = nontemporal gradient loads: measured slower (0.293 vs 0.263 ms - the re-reads of neighbouring blocks want L2)
#endif
__device__ __forceinline__ float4 ga_ntload(const float* p) {
    typedef float ga_f4 __attribute__((ext_vector_type(4)));
    const ga_f4 v = __builtin_nontemporal_load(reinterpret_cast<const ga_f4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
template <int GH, int GW, bool MASKCAT>
__global__ __launch_bounds__(256, CIM_ROI_GOCC) void roi_align_bwd_gather_kernel(const float* __restrict__ grad_out,
                                                                   const float* __restrict__ masks,
                                                                   float* __restrict__ grad_in, int C, int H, int W,
                                                                   int K, int P, int B, int use_atomic,
                                                                   const float* __restrict__ rec_all) {
    constexpr int NPX = GH * GW;
    __shared__ int e_off[GA_MAXE];
    __shared__ float e_m[GA_MAXE];
    __shared__ __attribute__((aligned(16))) float e_w[GA_MAXE * NPX];
    __shared__ int s_wave[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_x = (W + GW - 1) / GW;
    // XCD-aware tile order (speed only): workgroup i runs on XCD i % 8; give each XCD a contiguous run of pixel
    // tiles so neighbouring tiles - which share most of their bins' gradient vectors - re-read them from ONE L2
    int tile = blockIdx.x, by = blockIdx.y;
#ifndef CIM_ROI_NO_XCD
    if (gridDim.z == 1) {      // (with channel chunks in grid.z the slices start on different XCDs: keep the plain order)
        // workgroups are dealt to the XCDs round robin in dispatch order (x fastest, then y): remap the LINEAR index so
        // that XCD c works on a contiguous run of (group, tile) pairs
        const int nt = gridDim.x * gridDim.y, lin = blockIdx.x + gridDim.x * blockIdx.y;
        const int q = nt >> 3, r = nt & 7, xcd = lin & 7, i = lin >> 3;
        const int j = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
        tile = j % gridDim.x;
        by = j / gridDim.x;
    }
#endif
    const int y0 = (tile / tiles_x) * GH, x0 = (tile % tiles_x) * GW;
    const int b = by % B, kgroup = by / B;
    const int recw = roi_rec_words(P, H, W);
    const int OC = MASKCAT ? 2 * C : C, PP = P * P;

    ga_f2 accl[NPX], acch[NPX];
#pragma unroll
    for (int p = 0; p < NPX; ++p) accl[p] = acch[p] = ga_f2{0.f, 0.f};

    // ---- this lane's ROI: bins touching the block's rows / columns
    const int k = kgroup * GA_GS + tid;
    int phl = 0, phh = -1, pwl = 0, pwh = -1;
    const float* rec = rec_all + (size_t)min(k, K - 1) * recw;
    const int* yr = reinterpret_cast<const int*>(rec + P * (H + W));
    const int* xr = yr + H;
    float inv_count = 0.0f;
    if (tid < GA_GS && k < K && yr[H + W + 5] == b) {
        int lo = P, hi = -1;
#pragma unroll
        for (int i = 0; i < GH; ++i) {
            const int r = yr[min(y0 + i, H - 1)];
            if (y0 + i < H && !(r & 0x10000)) { lo = min(lo, r & 0xff); hi = max(hi, (r >> 8) & 0xff); }
        }
        phl = lo; phh = hi;
        lo = P; hi = -1;
#pragma unroll
        for (int j = 0; j < GW; ++j) {
            const int r = xr[min(x0 + j, W - 1)];
            if (x0 + j < W && !(r & 0x10000)) { lo = min(lo, r & 0xff); hi = max(hi, (r >> 8) & 0xff); }
        }
        pwl = lo; pwh = hi;
        inv_count = 1.0f / reinterpret_cast<const float*>(yr)[H + W + 4];
    }
    const int nph = max(phh - phl + 1, 0), npw = max(pwh - pwl + 1, 0);
    const int n_mine = (nph > 0 && npw > 0) ? nph * npw : 0;
    // ---- block-wide exclusive prefix sum of n_mine (deterministic entry order): wave scan + 4 wave totals
    int incl = n_mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int wbase = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const int t = s_wave[w];
        if (w < wave) wbase += t;
        total += t;
    }
    const int base = wbase + incl - n_mine;
    const int c = min(blockIdx.z * 1024 + tid * 4, C - 4);      // lanes past C redo the last quad (never stored)
    const float* __restrict__ gc = grad_out + c;

#if CIM_ROI_GNT
#define GA_LD4(P) ga_ntload(P)
#else
#define GA_LD4(P) (*reinterpret_cast<const float4*>(P))
#endif
#define GA_LOAD(G, I)                                                                                  \
    float4 G = GA_LD4(gc + e_off[I]);                                                                  \
    float4 G##h;                                                                                       \
    if (MASKCAT) G##h = GA_LD4(gc + e_off[I] + C);
#define GA_ACC(G, I)                                                                                   \
    {                                                                                                  \
        ga_f2 gl_ = ga_lo(G), gh_ = ga_hi(G);                                                          \
        if (MASKCAT) {                                                                                 \
            const float m_ = e_m[I];                                                                   \
            gl_ = ga_fma(m_, ga_lo(G##h), gl_);                                                        \
            gh_ = ga_fma(m_, ga_hi(G##h), gh_);                                                        \
        }                                                                                              \
        _Pragma("unroll") for (int p = 0; p < NPX; ++p) {                                              \
            const float w_ = e_w[(I) * NPX + p];                                                       \
            accl[p] = ga_fma(w_, gl_, accl[p]);                                                        \
            acch[p] = ga_fma(w_, gh_, acch[p]);                                                        \
        }                                                                                              \
    }
    for (int w0 = 0; w0 < total; w0 += GA_MAXE) {
        // ---- lay out the entries whose index falls in [w0, w0 + GA_MAXE)
        if (n_mine > 0 && base < w0 + GA_MAXE && base + n_mine > w0) {
            const float* wy = rec;
            const float* wx = rec + P * H;
            int idx = base - w0;
            for (int ph = phl; ph <= phh; ++ph) {
                float wyv[GH];
#pragma unroll
                for (int i = 0; i < GH; ++i) wyv[i] = (y0 + i < H) ? wy[ph * H + y0 + i] * inv_count : 0.0f;
                for (int pw = pwl; pw <= pwh; ++pw, ++idx) {
                    if (idx < 0 || idx >= GA_MAXE) continue;
                    e_off[idx] = (((k * P) + ph) * P + pw) * OC;      // < 2^31: checked by the launcher
                    if (MASKCAT) e_m[idx] = masks[(size_t)k * PP + ph * P + pw];
#pragma unroll
                    for (int j = 0; j < GW; ++j) {
                        const float wxv = (x0 + j < W) ? wx[pw * W + x0 + j] : 0.0f;
#pragma unroll
                        for (int i = 0; i < GH; ++i) e_w[idx * NPX + i * GW + j] = wyv[i] * wxv;
                    }
                }
            }
        }
        __syncthreads();
        // ---- stream the window: lanes along C, 4 entries in flight
#if CIM_ROI_GEXP == 1
        const int n = 0;
#else
        const int n = min(GA_MAXE, total - w0);
#endif
        int i = 0;
#if CIM_ROI_GU == 8
        for (; i + 8 <= n; i += 8) {
            GA_LOAD(g0, i)
            GA_LOAD(g1, i + 1)
            GA_LOAD(g2, i + 2)
            GA_LOAD(g3, i + 3)
            GA_LOAD(g4, i + 4)
            GA_LOAD(g5, i + 5)
            GA_LOAD(g6, i + 6)
            GA_LOAD(g7, i + 7)
            GA_ACC(g0, i)
            GA_ACC(g1, i + 1)
            GA_ACC(g2, i + 2)
            GA_ACC(g3, i + 3)
            GA_ACC(g4, i + 4)
            GA_ACC(g5, i + 5)
            GA_ACC(g6, i + 6)
            GA_ACC(g7, i + 7)
        }
#endif
        for (; i + 4 <= n; i += 4) {
            GA_LOAD(g0, i)
            GA_LOAD(g1, i + 1)
            GA_LOAD(g2, i + 2)
            GA_LOAD(g3, i + 3)
            GA_ACC(g0, i)
            GA_ACC(g1, i + 1)
            GA_ACC(g2, i + 2)
            GA_ACC(g3, i + 3)
        }
        for (; i < n; ++i) {
            GA_LOAD(g0, i)
            GA_ACC(g0, i)
        }
        __syncthreads();
    }
#undef GA_LOAD
#undef GA_ACC
    const int cs = blockIdx.z * 1024 + tid * 4;
#if CIM_ROI_GEXP == 2
    if (cs < C && acc[0].x == 123.456f) {
#else
    if (cs < C && (total > 0 || !use_atomic)) {
#endif
#pragma unroll
        for (int i = 0; i < GH; ++i)
#pragma unroll
            for (int j = 0; j < GW; ++j)
                if (y0 + i < H && x0 + j < W) {
                    float* dst = grad_in + (((size_t)b * H + y0 + i) * W + x0 + j) * C + cs;
                    const float4 v = make_float4(accl[i * GW + j].x, accl[i * GW + j].y, acch[i * GW + j].x, acch[i * GW + j].y);
                    if (use_atomic) {
                        atomicAdd(dst + 0, v.x); atomicAdd(dst + 1, v.y); atomicAdd(dst + 2, v.z); atomicAdd(dst + 3, v.w);
                    } else {
                        *reinterpret_cast<float4*>(dst) = v;
                    }
                }
    }
}

// ---------------------------------------------------------------------------------------------
// Backward, region form (default): the gather above with the sharing made explicit.
//
// What bounds the gather kernel is that every gradient vector is re-read by each of the ~3.7 pixel blocks its bin
// touches, from workgroups on different XCDs: 1.5 GB through the fabric for 401 MB of gradients.  Here a workgroup of
// 16 waves owns a REGION of 12 x 16 feature pixels (4 x 4 sub-blocks of 3 x 4, one per wave, 12 float4 accumulators per
// lane) x a 256-channel slice (lanes along C, 16 B per lane) for one group of ROIs.  The gradient slices of the bins that
// touch the region are brought in ONCE per workgroup - all 1024 lanes stream a window of 32 entries (64 KB with the
// mask-cat halves, combined on the way: g_lo + m g_hi) into LDS, double buffered, the next window's loads in flight
// while the waves consume the current one - and every wave reads from LDS the entries whose bin touches its sub-block
// (a ballot over the window's row / column masks picks them).  Re-read factor of a 12 x 16 region at the benchmark's bin
// size (3.6 x 4.1 px): (15.6 / 12)(20.1 / 16) = 1.6 instead of 3.7; registers per lane as in the 3 x 4 gather.
// Entry record in LDS: gradient offset, mask, separable weights wy[4 sub-rows][4] (3 used) and wx[4 sub-columns][4], touch mask.
// Regions are dispatched centre first (they carry the most entries); groups meet in grad_in through atomicAdd.
// ROIs per workgroup (template parameter GS): rg_group_size() below (32: 0.66, 96: 0.26, 160: 0.38, 256: 0.56 ms at 1000 ROIs).
// Entry order inside a group: interleaved over the ROIs (see the entry map below); sub-blocks spread over the SIMDs.
#ifndef CIM_ROI_RG_DIRECT
#define CIM_ROI_RG_DIRECT 0         // 1 = no LDS staging: every wave loads its entries itself (L1 / L2 serve the re-reads)
#endif
#ifndef CIM_ROI_RG_EXP
#define CIM_ROI_RG_EXP 0            // ablations: 1 = no consume phase, 2 = no gradient loads, 3 = records only
#endif
constexpr int RG_SBH = 3, RG_SBW = 4, RG_WR = 4, RG_WC = 4;
constexpr int RG_RH = RG_SBH * RG_WR, RG_RW = RG_SBW * RG_WC;          // 12 x 16 pixels
constexpr int RG_NT = 64 * RG_WR * RG_WC;                               // 1024 threads
constexpr int RG_WIN = 32;                                              // entries per staging window
constexpr int RG_MAXE = 256;                                            // entries per super-window (records in LDS)
// ROIs per workgroup: 64, or 128 when 64 would make more than ~3.5 workgroups per CU (measured, ms at 64 / 128:
// 1000 ROIs on 33 x 43: 0.188 / 0.284, 800 on 27 x 36: 0.159 / 0.253, 1200 on 41 x 54: 0.288 / 0.261, 2000 on 33 x 43: 0.353 / 0.321)
static inline int rg_group_size(int K, int B, int C, int H, int W) {
    const char* e = getenv("CIM_ROI_RG_GS");                // 64 / 128: sweep switch (tools/bench_roi_bwd.py)
    if (e && (atoi(e) == 64 || atoi(e) == 128)) return atoi(e);
    const long long wgs = (long long)B * ((K + 63) / 64) * ((H + 11) / 12) * ((W + 15) / 16) * ((C + 255) / 256);
    return wgs > 900 ? 128 : 64;
}
constexpr int RG_REC = 36;                                              // words per entry record: off, m, touch, pad, wy[16], wx[16]
constexpr int RG_MAXREG = 256;
struct RegionOrder { unsigned char o[RG_MAXREG]; };
static inline size_t rg_lds_bytes(int GS, int P) {     // staging windows + entry records + entry map
    return sizeof(float) * (2 * RG_WIN * 256 + RG_MAXE * RG_REC) + sizeof(unsigned short) * ((size_t)GS * P * P + 8);
}

template <bool MASKCAT, int RG_GS>
__global__ __launch_bounds__(RG_NT) void roi_align_bwd_region_kernel(const float* __restrict__ grad_out,
                                                                     const float* __restrict__ masks,
                                                                     float* __restrict__ grad_in, int C, int H, int W, int K,
                                                                     int P, int B, int use_atomic,
                                                                     const float* __restrict__ rec_all,
                                                                     const RegionOrder region_order, int n_regions,
                                                                     int regions_x, int n_slices, float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float rg_smem[];
    float (*stage)[RG_WIN][256] = reinterpret_cast<float (*)[RG_WIN][256]>(rg_smem);   // [2][32][256]: 64 KB
    float* erec = rg_smem + 2 * RG_WIN * 256;                                         // [MAXE][REC]: 36 KB
    unsigned short* emap = reinterpret_cast<unsigned short*>(erec + RG_MAXE * RG_REC);   // [GS * P * P]: entry -> ROI | ph << 7 | pw << 11
    __shared__ int s_wave[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // work order: region (centre first) slowest, then ROI group, then channel slice
    int lin = blockIdx.x;
    const int slice = lin % n_slices;
    lin /= n_slices;
    const int groups_total = gridDim.x / (n_slices * n_regions);                      // B * groups
    const int by = lin % groups_total;
    const int region = region_order.o[lin / groups_total];
    const int b = by % B, kgroup = by / B;
    const int y0 = (region / regions_x) * RG_RH, x0 = (region % regions_x) * RG_RW;
    // wave w runs on SIMD w % 4: the sub-blocks of one SIMD are spread out (row t, column (s + 2 t) % 4: every 2 x 2
    // neighbourhood of sub-blocks sits on four different SIMDs) - a bin hits ADJACENT sub-blocks together
    const int wr = wave / RG_WC, wc = (wave + 2 * wr) % RG_WC;
    const int sy0 = y0 + wr * RG_SBH, sx0 = x0 + wc * RG_SBW;
    const int recw = roi_rec_words(P, H, W);
    const int OC = MASKCAT ? 2 * C : C, PP = P * P;

    ga_f2 accl[12], acch[12];
#pragma unroll
    for (int p = 0; p < 12; ++p) accl[p] = acch[p] = ga_f2{0.f, 0.f};

    // ---- inspection: thread t < GS looks at ROI kgroup * GS + t and leaves a descriptor in LDS
    __shared__ int d_base[RG_GS + 1];          // first entry index of the ROI (exclusive prefix sum), [GS] = total
    __shared__ int d_bins[RG_GS];              // phl | pwl << 8 | npw << 16
    __shared__ float d_inv[RG_GS];             // 1 / count
    __shared__ int d_cnt[RG_GS];               // entries of the ROI
    {
        const int k = kgroup * RG_GS + tid;
        int phl = 0, phh = -1, pwl = 0, pwh = -1;
        float inv_count = 0.0f;
        if (tid < RG_GS && k < K) {
            const float* rec = rec_all + (size_t)k * recw;
            const int* yr = reinterpret_cast<const int*>(rec + P * (H + W));
            const int* xr = yr + H;
            if (yr[H + W + 5] == b) {
                int lo = P, hi = -1;
#pragma unroll
                for (int i = 0; i < RG_RH; ++i) {
                    const int r = yr[min(y0 + i, H - 1)];
                    if (y0 + i < H && !(r & 0x10000)) { lo = min(lo, r & 0xff); hi = max(hi, (r >> 8) & 0xff); }
                }
                phl = lo; phh = hi;
                lo = P; hi = -1;
#pragma unroll
                for (int j = 0; j < RG_RW; ++j) {
                    const int r = xr[min(x0 + j, W - 1)];
                    if (x0 + j < W && !(r & 0x10000)) { lo = min(lo, r & 0xff); hi = max(hi, (r >> 8) & 0xff); }
                }
                pwl = lo; pwh = hi;
                inv_count = 1.0f / reinterpret_cast<const float*>(yr)[H + W + 4];
            }
        }
        const int nph = max(phh - phl + 1, 0), npw = max(pwh - pwl + 1, 0);
        const int n_mine = (nph > 0 && npw > 0) ? nph * npw : 0;
        int incl = n_mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o);
            if (lane >= o) incl += v;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        int wbase = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w)
            if (w < wave) wbase += s_wave[w];
        if (tid < RG_GS) {
            d_base[tid] = wbase + incl - n_mine;
            d_bins[tid] = phl | (pwl << 8) | (max(npw, 1) << 16);
            d_inv[tid] = inv_count;
            d_cnt[tid] = n_mine;
        }
        if (tid == RG_GS - 1) d_base[RG_GS] = wbase + incl;
        __syncthreads();
        // ---- entry order: INTERLEAVED over the group's ROIs (level l = the l-th touching bin of every ROI that has one):
        // consecutive entries come from different ROIs, i.e. from all over the region, so every window spreads over all
        // waves.  In ROI-major order a window holds the bins of one or two ROIs and costs the time of the 2-4 waves they
        // hit (summed over the windows of cfg2 the busiest wave saw 32 k entries against 9 k on average).
        // Level l starts at sum_r min(n_r, l); wave w writes the levels l = w (mod 16), rank inside a level by ballot.
        {
            constexpr int RPL = RG_GS / 64;                                  // ROIs per lane
            int n[RPL], bins[RPL];
#pragma unroll
            for (int h = 0; h < RPL; ++h) { n[h] = d_cnt[lane + 64 * h]; bins[h] = d_bins[lane + 64 * h]; }
            const unsigned long long below = (1ull << lane) - 1ull;
            for (int l = wave; l < PP; l += 16) {
                int mins = 0, before = 0;
                unsigned long long have[RPL];
                bool any = false;
#pragma unroll
                for (int h = 0; h < RPL; ++h) { have[h] = __ballot(n[h] > l); mins += min(n[h], l); any |= have[h] != 0; }
                if (!any) break;                                             // uniform: levels are nested
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) mins += __shfl_xor(mins, o);
#pragma unroll
                for (int h = 0; h < RPL; ++h) {
                    if (n[h] > l) {
                        const int npw = bins[h] >> 16;
                        const int dph = (int)(((float)l + 0.5f) / (float)npw);      // l / npw (exact for these sizes)
                        const int ph = (bins[h] & 0xff) + dph, pw = ((bins[h] >> 8) & 0xff) + l - dph * npw;
                        emap[mins + before + __popcll(have[h] & below)] = (unsigned short)((lane + 64 * h) | (ph << 7) | (pw << 11));
                    }
                    before += __popcll(have[h]);
                }
            }
        }
        __syncthreads();
    }
    const int total = (CIM_ROI_RG_EXP == 6) ? 0 : d_base[RG_GS];

    // loader role of this thread inside a staging window: entry le = tid / 32, two float4 of the slice's 64
    const int le = tid >> 5, lq = (tid & 31) * 2;
    const int cbase = slice * 256;
    // consumer role: lane's 4 channels of the slice
    const int my_touch = (7 << (RG_SBH * wr)) | ((15 << (RG_SBW * wc)) << 16);

    for (int w0 = 0; w0 < total; w0 += RG_MAXE) {
        __syncthreads();                                   // previous super-window fully consumed
        // ---- entry records of [w0, w0 + MAXE): 4 threads per entry, each 3 rows of wy and 4 columns of wx
        {
            const int e = tid >> 2, part = tid & 3;
            const int ge = w0 + e;
            int rm = 0, cm = 0;
            if (ge < total) {
                const int code = emap[ge];
                const int lo = code & 127, ph = (code >> 7) & 15, pw = code >> 11;
                const int k = kgroup * RG_GS + lo;
                const float* rec = rec_all + (size_t)k * recw;
                const float* wy = rec + ph * H;
                const float* wx = rec + P * H + pw * W;
                const float inv_count = d_inv[lo];
                float* r = erec + e * RG_REC;
#pragma unroll
                for (int i = 0; i < RG_SBH; ++i) {
                    const int y = y0 + part * RG_SBH + i;
                    const float v = (y < H) ? wy[y] * inv_count : 0.0f;
                    r[4 + part * 4 + i] = v;
                    if (v != 0.0f) rm |= 1 << (part * RG_SBH + i);
                }
#pragma unroll
                for (int j = 0; j < RG_SBW; ++j) {
                    const int x = x0 + part * RG_SBW + j;
                    const float v = (x < W) ? wx[x] : 0.0f;
                    r[20 + part * 4 + j] = v;
                    if (v != 0.0f) cm |= 1 << (part * RG_SBW + j);
                }
                if (part == 0) {
                    reinterpret_cast<int*>(r)[0] = (((k * P) + ph) * P + pw) * OC;      // < 2^31: checked by the launcher
                    r[1] = MASKCAT ? masks[(size_t)k * PP + ph * P + pw] : 0.0f;
                }
            }
            rm |= __shfl_xor(rm, 1); rm |= __shfl_xor(rm, 2);
            cm |= __shfl_xor(cm, 1); cm |= __shfl_xor(cm, 2);
            if (ge < total && part == 0) reinterpret_cast<int*>(erec + e * RG_REC)[2] = rm | (cm << 16);
        }
        __syncthreads();
        const int n = min(RG_MAXE, total - w0);
        const int nwin = (CIM_ROI_RG_EXP == 3 || CIM_ROI_RG_EXP == 5) ? 0 : (n + RG_WIN - 1) / RG_WIN;
        // ---- software pipeline over the staging windows: the next window's loads are in flight while this one is consumed
        // (two windows in flight - a second register set - measured no faster: 0.213 vs 0.207 ms)
        struct Regs { float4 r0, r1, h0, h1; float m; };
        auto gload = [&](Regs& R, int win) {
            const int e = win * RG_WIN + le;
            R.r0 = R.r1 = R.h0 = R.h1 = make_float4(0.f, 0.f, 0.f, 0.f);
            R.m = 0.0f;
            if (e < n && CIM_ROI_RG_EXP != 2) {
                const float* rr = erec + e * RG_REC;
                const int off = reinterpret_cast<const int*>(rr)[0];
                const int c0 = min(cbase + lq * 4, C - 4), c1 = min(cbase + lq * 4 + 4, C - 4);
                R.r0 = *reinterpret_cast<const float4*>(grad_out + off + c0);
                R.r1 = *reinterpret_cast<const float4*>(grad_out + off + c1);
                if (MASKCAT) {
                    R.h0 = *reinterpret_cast<const float4*>(grad_out + off + C + c0);
                    R.h1 = *reinterpret_cast<const float4*>(grad_out + off + C + c1);
                    R.m = rr[1];
                }
            }
        };
        auto put = [&](const Regs& R, int buf) {
            float4 a = R.r0, c = R.r1;
            if (MASKCAT) {
                a = make_float4(fmaf(R.m, R.h0.x, a.x), fmaf(R.m, R.h0.y, a.y), fmaf(R.m, R.h0.z, a.z), fmaf(R.m, R.h0.w, a.w));
                c = make_float4(fmaf(R.m, R.h1.x, c.x), fmaf(R.m, R.h1.y, c.y), fmaf(R.m, R.h1.z, c.z), fmaf(R.m, R.h1.w, c.w));
            }
            float* sb = &stage[buf][le][lq * 4];
            *reinterpret_cast<float4*>(sb) = a;
            *reinterpret_cast<float4*>(sb + 4) = c;
        };
        auto consume = [&](int buf, int win) {
            // entries of the window whose bin touches this wave's sub-block
            const int e0 = win * RG_WIN;
            int t = 0;
            if (lane < RG_WIN && e0 + lane < n) t = reinterpret_cast<const int*>(erec + (e0 + lane) * RG_REC)[2];
            const bool hit = ((t & my_touch & 0xffff) != 0) && (((t & my_touch) >> 16) != 0);
            unsigned long long todo = __ballot(hit);
            if (CIM_ROI_RG_EXP == 1) todo = 0;
            while (todo) {
                const int j = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                const float* rr = erec + (e0 + j) * RG_REC;
                const float4 g = *reinterpret_cast<const float4*>(&stage[buf][j][lane * 4]);
                const float4 wyq = *reinterpret_cast<const float4*>(rr + 4 + wr * 4);
                const float4 wxq = *reinterpret_cast<const float4*>(rr + 20 + wc * 4);
                const ga_f2 gl = ga_lo(g), gh = ga_hi(g);
                const float wys[3] = {wyq.x, wyq.y, wyq.z};
                const float wxs[4] = {wxq.x, wxq.y, wxq.z, wxq.w};
                // (tried: skipping rows with a zero weight by a scalar branch, 0.227 vs 0.213 ms; reading the next entry's
                // LDS operands one iteration ahead, 0.224 vs 0.207 ms - the phase is paced by the busiest wave of a window)
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const ga_f2 tl = ga_f2{wys[i], wys[i]} * gl, th = ga_f2{wys[i], wys[i]} * gh;
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        accl[i * 4 + jj] = ga_fma(wxs[jj], tl, accl[i * 4 + jj]);
                        acch[i * 4 + jj] = ga_fma(wxs[jj], th, acch[i * 4 + jj]);
                    }
                }
            }
        };
#if CIM_ROI_RG_DIRECT
        // DIRECT variant: no LDS staging and no per-window barrier - every wave loads the gradient slices of the entries
        // that touch its sub-block itself, four entries (8 x 16 B loads) in flight; the ~2.5 waves of the workgroup that
        // share an entry re-read it from this CU's L1 / this XCD's L2 (all consumers of a region sit on one CU), so the
        // fabric still sees each slice once per region.
        (void)gload; (void)put; (void)consume; (void)nwin;
        const float* __restrict__ gc = grad_out + min(cbase + lane * 4, C - 4);
#define RD_LOAD(G, J)                                                                                  \
        const float* rr##G = erec + (J) * RG_REC;                                                      \
        const int off##G = reinterpret_cast<const int*>(rr##G)[0];                                     \
        float4 G = *reinterpret_cast<const float4*>(gc + off##G);                                      \
        float4 G##h;                                                                                   \
        if (MASKCAT) G##h = *reinterpret_cast<const float4*>(gc + off##G + C);
#define RD_ACC(G)                                                                                      \
        {                                                                                              \
            ga_f2 gl_ = ga_lo(G), gh_ = ga_hi(G);                                                      \
            if (MASKCAT) {                                                                             \
                const float m_ = rr##G[1];                                                             \
                gl_ = ga_fma(m_, ga_lo(G##h), gl_);                                                    \
                gh_ = ga_fma(m_, ga_hi(G##h), gh_);                                                    \
            }                                                                                          \
            const float4 wyq = *reinterpret_cast<const float4*>(rr##G + 4 + wr * 4);                   \
            const float4 wxq = *reinterpret_cast<const float4*>(rr##G + 20 + wc * 4);                  \
            const float wys[3] = {wyq.x, wyq.y, wyq.z};                                                \
            const float wxs[4] = {wxq.x, wxq.y, wxq.z, wxq.w};                                         \
            _Pragma("unroll") for (int i = 0; i < 3; ++i) {                                            \
                const ga_f2 tl = ga_f2{wys[i], wys[i]} * gl_, th = ga_f2{wys[i], wys[i]} * gh_;        \
                _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) {                                     \
                    accl[i * 4 + jj] = ga_fma(wxs[jj], tl, accl[i * 4 + jj]);                          \
                    acch[i * 4 + jj] = ga_fma(wxs[jj], th, acch[i * 4 + jj]);                          \
                }                                                                                      \
            }                                                                                          \
        }
        for (int e0 = 0; e0 < n; e0 += 64) {
            int t = 0;
            if (e0 + lane < n) t = reinterpret_cast<const int*>(erec + (e0 + lane) * RG_REC)[2];
            const bool hit = ((t & my_touch & 0xffff) != 0) && (((t & my_touch) >> 16) != 0);
            unsigned long long todo = __ballot(hit);
            while (todo) {
                const int cnt = __popcll(todo);
                if (cnt >= 4) {
                    const int j0 = __ffsll((long long)todo) - 1; todo &= todo - 1;
                    const int j1 = __ffsll((long long)todo) - 1; todo &= todo - 1;
                    const int j2 = __ffsll((long long)todo) - 1; todo &= todo - 1;
                    const int j3 = __ffsll((long long)todo) - 1; todo &= todo - 1;
                    RD_LOAD(g0, e0 + j0)
                    RD_LOAD(g1, e0 + j1)
                    RD_LOAD(g2, e0 + j2)
                    RD_LOAD(g3, e0 + j3)
                    RD_ACC(g0)
                    RD_ACC(g1)
                    RD_ACC(g2)
                    RD_ACC(g3)
                } else {
                    const int j0 = __ffsll((long long)todo) - 1; todo &= todo - 1;
                    RD_LOAD(g0, e0 + j0)
                    RD_ACC(g0)
                }
            }
        }
#undef RD_LOAD
#undef RD_ACC
#else
        Regs A;
        gload(A, 0);
        for (int win = 0; win < nwin; ++win) {
            put(A, win & 1);
            __syncthreads();                               // (also: every wave is done with the buffer the NEXT put overwrites)
            if (win + 1 < nwin) gload(A, win + 1);         // in flight while this window is consumed
            consume(win & 1, win);
        }
#endif
    }
    // ---- flush this wave's 3 x 4 pixels
    const int cs = slice * 256 + lane * 4;
    if (cs < C && (total > 0 || !use_atomic || partial) && ((CIM_ROI_RG_EXP != 4 && CIM_ROI_RG_EXP != 5 && CIM_ROI_RG_EXP != 6) || accl[0].x == 123.456f)) {
#pragma unroll
        for (int i = 0; i < RG_SBH; ++i)
#pragma unroll
            for (int j = 0; j < RG_SBW; ++j)
                if (sy0 + i < H && sx0 + j < W && sy0 + i < y0 + RG_RH) {
                    const size_t o = (((size_t)b * H + sy0 + i) * W + sx0 + j) * C + cs;
                    float* dst = grad_in + o;
                    const float4 v = make_float4(accl[i * 4 + j].x, accl[i * 4 + j].y, acch[i * 4 + j].x, acch[i * 4 + j].y);
                    if (partial) {      // the groups' partial maps are summed by roi_partial_reduce_kernel (no atomics)
                        __builtin_nontemporal_store(v.x, partial + (size_t)kgroup * B * H * W * C + o);
                        __builtin_nontemporal_store(v.y, partial + (size_t)kgroup * B * H * W * C + o + 1);
                        __builtin_nontemporal_store(v.z, partial + (size_t)kgroup * B * H * W * C + o + 2);
                        __builtin_nontemporal_store(v.w, partial + (size_t)kgroup * B * H * W * C + o + 3);
                    } else if (use_atomic) {
                        atomicAdd(dst + 0, v.x); atomicAdd(dst + 1, v.y); atomicAdd(dst + 2, v.z); atomicAdd(dst + 3, v.w);
                    } else {
                        *reinterpret_cast<float4*>(dst) = v;
                    }
                }
    }
}

// ---------------------------------------------------------------------------------------------
// Backward, pipelined region form (opt-in: CIM_ROI_BWD_PIPE=1, P == 7): the region form above with the loading and the
// accumulating decoupled - an experiment that measured NO faster (cfg2: 0.197 vs 0.195 ms; 800 ROIs 0.156 vs 0.165; 1200
// ROIs 0.313 vs 0.307), kept for its instrumentation (CIM_ROI_PL_TRACE + tools/trace_roi_bwd.py) and for what it showed:
//   * the 16 waves of the workgroup are specialised: 4 PRODUCER waves stream the gradient slices (wave p takes batch p of
//     every round of 16 entries: entry map look-up, both mask-cat halves of the 4 entries' 256-channel slices - 8 x 16 B
//     per lane, the NEXT round's loads issued before this round's are waited for - combined g_lo + m g_hi into a ring of
//     4 rounds in LDS with a two-word descriptor, published through a byte counter in LDS); 12 CONSUMER waves own 4 x 4
//     pixel sub-blocks (16 float4 accumulators per lane), poll the producers' counters (one word), pick the entries of the
//     round that touch their sub-block by a ballot over the descriptors, accumulate from LDS with the next hit's operands
//     in flight, and post their own counter, which the producers read before they overwrite a ring slot.  No workgroup
//     barrier inside the stream.  The per-ROI separable weights are staged in LDS once per chunk of 64 ROIs (196 floats
//     per ROI) instead of travelling in per-entry records.
//   * per-workgroup time stamps: set-up 9 us (the first version 13-19 us: the prologue is INSTRUCTION bound - 16 waves
//     share 4 SIMDs, index arithmetic with constant divisions cost ~1400 VALU slots per wave - not latency bound), then a
//     constant 0.105 us per entry whatever the workgroup's size or the load of the chip, 2 us flush; producers alone
//     0.058 us / entry, consumers alone 0.086.  The heaviest workgroup (1467 entries, centre region) alone takes
//     144 of the kernel's 162 us; the CUs are busy 131 us on average.
//   * what bounds both this and the region kernel is VALU issue: the FMAs are ~80 cycles per entry per SIMD when spread
//     perfectly (34 us for the launch), the rest is per-round / per-window bookkeeping on 12-16 waves; HBM is not the limit
//     (a CU streams 19 GB/s here against its 28 GB/s share).  Interleaving the entry order and spreading the sub-blocks over
//     the SIMDs came out of this and moved into the region kernel (-6 %).
#ifndef CIM_ROI_PL_EXP
#define CIM_ROI_PL_EXP 0            // ablations: 1 = consumers skip the accumulation, 2 = producers skip the gradient loads, 3 = chunk set-up only
#endif
#ifndef CIM_ROI_PL_TRACE
#define CIM_ROI_PL_TRACE 0          // 1 = (tools/trace_roi_bwd.py) per-workgroup time stamps behind the partial maps
#endif
#if CIM_ROI_PL_TRACE
#define PL_TRACE_PTR (pl_trace + (size_t)blockIdx.x * 16)
#define PL_STAMP(I) if (lane == 0 && (wave == 0 || wave == 15)) pl_trace[(size_t)blockIdx.x * 16 + (wave ? 8 : 0) + (I)] = wall_clock64()
#else
#define PL_TRACE_PTR nullptr
#define PL_STAMP(I)
#endif
constexpr int PL_NPROD = 4, PL_NCONS = 12;
constexpr int PL_BT = 4;                                     // entries per producer batch (16 lanes each)
constexpr int PL_ROUND = PL_NPROD * PL_BT;                   // entries per round
constexpr int PL_RING = 4;                                   // rounds in the LDS ring
constexpr int PL_SLOTS = PL_RING * PL_ROUND;                 // 64 entries x 1 KB
constexpr int PL_GS = 64;                                    // ROIs per chunk
constexpr int PL_WR = 3, PL_WC = 4;                          // consumer sub-blocks: 3 x 4 of 4 x 4 pixels
static_assert(PL_WR * 4 == RG_RH && PL_WC * 4 == RG_RW && PL_NPROD + PL_NCONS == 16, "region geometry");
static inline int pl_chunks(int K, int B, int C, int H, int W) { return rg_group_size(K, B, C, H, W) / 64; }
template <int P> constexpr int pl_tab_words() { return P * (RG_RH + RG_RW); }
template <int P> constexpr size_t pl_lds_bytes() {
    return sizeof(float) * (PL_SLOTS * 256 + PL_GS * pl_tab_words<P>() + PL_GS * P * P) + sizeof(int) * (PL_GS * 2 * P + 2 * PL_SLOTS) +
           sizeof(unsigned short) * (PL_GS * P * P + 8);
}

// chunk prologue, executed by all 16 waves (two workgroup barriers): returns the chunk's entry count.
// ONE global round trip: every thread first issues all its loads - its share of the ROIs' packed bin ranges (16 lanes
// per ROI: 12 rows + 16 columns of the region), of the region's table segments (49 unaligned 16-byte segments per ROI:
// 7 x 3 of wy, 7 x 4 of wx) and of the masks - then the ranges are reduced with shuffles, the tables go to LDS, and
// after the first barrier the entry map and the touch masks are derived from LDS alone.  (The first version went
// through three dependent global round trips of ~2 us each and scalar 4-byte gathers: 13 us per workgroup.)
// Entry order: INTERLEAVED over the chunk's ROIs (level l = the l-th touching bin of every ROI that has one, ROIs in
// order): consecutive entries come from different ROIs, i.e. from all over the region, so every round spreads over all
// consumer waves / SIMDs.  ROI-major order kept the same 2-4 sub-blocks busy for a whole ROI (up to 49 entries in a
// row, most of the 64-entry ring) while the other consumer waves starved.
typedef float pl_f4u __attribute__((ext_vector_type(4), aligned(4)));
template <bool MASKCAT, int P>
__device__ __forceinline__ int pl_chunk_setup(float* __restrict__ tab, float* __restrict__ mk, int* __restrict__ rcm,
                                              unsigned short* __restrict__ emap, int* d_cnt, int* d_bins, int* f_ready, int* f_done,
                                              const float* __restrict__ rec_all, const float* __restrict__ masks, int kb,
                                              int K, int H, int W, int b, int y0, int x0, int tid, int lane, int wave,
                                              unsigned long long* tr = nullptr) {
    constexpr int TW = pl_tab_words<P>(), PP = P * P;
    constexpr int SEG_Y = RG_RH / 4, SEG_X = RG_RW / 4, SEGS = P * (SEG_Y + SEG_X);       // 3, 4, 49 segments per ROI
    constexpr int RPW = PL_GS / 16;                                                        // ROIs per wave
    constexpr int NMK = (PL_GS * PP + RG_NT - 1) / RG_NT;
    static_assert(SEGS <= 64 && PL_GS % 16 == 0, "one lane per table segment");
    const int recw = roi_rec_words(P, H, W);
    // The prologue is INSTRUCTION bound (16 waves share 4 SIMDs; the first version spent ~1400 VALU slots per wave on
    // index arithmetic: 11 us), so everything per lane is computed once and each load costs an add.
    // ---- tables: wave w stages ROIs 4 w ... 4 w + 3, lane = segment (49 of 64 lanes); all loads first
    const bool seg_on = lane < SEGS;
    const int sg = seg_on ? lane : 0;
    const bool is_y = sg < P * SEG_Y;
    const int s2 = sg - P * SEG_Y;
    const int src_off = is_y ? (sg / SEG_Y) * H + y0 + 4 * (sg % SEG_Y) : P * H + (s2 / SEG_X) * W + x0 + 4 * (s2 % SEG_X);
    const int dst_off = is_y ? (sg / SEG_Y) * RG_RH + 4 * (sg % SEG_Y) : P * RG_RH + (s2 / SEG_X) * RG_RW + 4 * (s2 % SEG_X);
    const int pos = is_y ? y0 + 4 * (sg % SEG_Y) : x0 + 4 * (s2 % SEG_X), lim = is_y ? H : W;
    pl_f4u seg[RPW];
    float seg_cnt[RPW], mkv[NMK];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const float* rec2 = rec_all + (size_t)min(kb + wave * RPW + i, K - 1) * recw;
        seg[i] = *reinterpret_cast<const pl_f4u*>(rec2 + src_off);     // may run past the row's end: masked below, inside the record
        seg_cnt[i] = rec2[P * (H + W) + H + W + 4];
    }
    if (MASKCAT) {
#pragma unroll
        for (int i = 0; i < NMK; ++i) mkv[i] = masks[min((size_t)kb * PP + tid + i * RG_NT, (size_t)K * PP - 1)];
    }
    if (wave == 15) {
        // ---- (a producer wave: registers to spare) lane t inspects ROI kb + t: the bins that touch the region, from the
        // packed per-row / per-column ranges (28 clamped loads, in flight with the segment loads)
        const float* rec = rec_all + (size_t)min(kb + lane, K - 1) * recw;
        const int* yr = reinterpret_cast<const int*>(rec + P * (H + W));
        const int* xr = yr + H;
        int wy_[RG_RH], wx_[RG_RW];
#pragma unroll
        for (int i = 0; i < RG_RH; ++i) wy_[i] = yr[min(y0 + i, H - 1)];
#pragma unroll
        for (int jj = 0; jj < RG_RW; ++jj) wx_[jj] = xr[min(x0 + jj, W - 1)];
        const int batch = yr[H + W + 5];
        int ylo = P, yhi = -1, xlo = P, xhi = -1;
#pragma unroll
        for (int i = 0; i < RG_RH; ++i)
            if (y0 + i < H && !(wy_[i] & 0x10000)) { ylo = min(ylo, wy_[i] & 0xff); yhi = max(yhi, (wy_[i] >> 8) & 0xff); }
#pragma unroll
        for (int jj = 0; jj < RG_RW; ++jj)
            if (x0 + jj < W && !(wx_[jj] & 0x10000)) { xlo = min(xlo, wx_[jj] & 0xff); xhi = max(xhi, (wx_[jj] >> 8) & 0xff); }
        const bool in = (kb + lane < K) && batch == b;
        const int nph = max(yhi - ylo + 1, 0), npw = max(xhi - xlo + 1, 0);
        d_cnt[lane] = (in && nph > 0 && npw > 0) ? nph * npw : 0;
        d_bins[lane] = ylo | (xlo << 8) | (max(npw, 1) << 16);
        if (lane == 0) f_ready[0] = 0;
        if (lane < 16) f_done[lane] = 0;
    }
    if (seg_on) {
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const float sc = is_y ? 1.0f / seg_cnt[i] : 1.0f;          // 1 / count folded into wy
            float4 v;
            v.x = (pos + 0 < lim) ? seg[i].x * sc : 0.0f;
            v.y = (pos + 1 < lim) ? seg[i].y * sc : 0.0f;
            v.z = (pos + 2 < lim) ? seg[i].z * sc : 0.0f;
            v.w = (pos + 3 < lim) ? seg[i].w * sc : 0.0f;
            *reinterpret_cast<float4*>(tab + (wave * RPW + i) * TW + dst_off) = v;
        }
    }
    if (MASKCAT) {
#pragma unroll
        for (int i = 0; i < NMK; ++i)
            if (tid + i * RG_NT < PL_GS * PP) mk[tid + i * RG_NT] = mkv[i];
    }
    __syncthreads();
    if (CIM_ROI_PL_TRACE && tr && tid == 0) tr[4] = wall_clock64();
    // ---- entry map: level l of the interleaved order holds the l-th bin of every ROI with more than l bins; it starts
    // at sum_r min(n_r, l).  Wave w writes the levels l = w (mod 16): lanes = ROIs, rank inside the level by a ballot.
    const int n = d_cnt[lane], bins = d_bins[lane];
    auto wave_sum = [&](int v) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        return v;
    };
    const int total = wave_sum(n);
    {
        const int npw = bins >> 16;
        const float inv_npw = 1.0f / (float)npw;
        const unsigned long long below = (1ull << lane) - 1ull;
        for (int l = wave; l < PP; l += 16) {
            const unsigned long long have = __ballot(n > l);
            if (have == 0) break;                                      // uniform: levels are nested
            const int base = wave_sum(min(n, l));
            if (n > l) {
                const int dph = (int)(((float)l + 0.5f) * inv_npw);   // l / npw (exact: npw <= P)
                const int ph = (bins & 0xff) + dph, pw = ((bins >> 8) & 0xff) + l - dph * npw;
                emap[base + __popcll(have & below)] = (unsigned short)(lane | (ph << 6) | (pw << 9));
            }
        }
    }
    // ---- per (ROI, bin row) / (ROI, bin column) touch masks over the region's rows / columns
    if (tid < PL_GS * 2 * P) {
        const int r3 = tid / (2 * P), q = tid % (2 * P);
        int m = 0;
        if (q < P) {
#pragma unroll
            for (int i = 0; i < RG_RH; ++i) m |= (tab[r3 * TW + q * RG_RH + i] != 0.0f) << i;
        } else {
#pragma unroll
            for (int jj = 0; jj < RG_RW; ++jj) m |= (tab[r3 * TW + P * RG_RH + (q - P) * RG_RW + jj] != 0.0f) << jj;
        }
        rcm[tid] = m;
    }
    __syncthreads();
    if (CIM_ROI_PL_TRACE && tr && tid == 0) tr[5] = wall_clock64();
    return total;
}

template <bool MASKCAT, int P>
__global__ __launch_bounds__(RG_NT) void roi_align_bwd_pipe_kernel(const float* __restrict__ grad_out,
                                                                   const float* __restrict__ masks,
                                                                   float* __restrict__ grad_in, int C, int H, int W, int K,
                                                                   int B, int chunks, const float* __restrict__ rec_all,
                                                                   const RegionOrder region_order, int n_regions,
                                                                   int regions_x, int n_slices, float* __restrict__ partial) {
    constexpr int TW = pl_tab_words<P>(), PP = P * P;
    extern __shared__ __attribute__((aligned(16))) float pl_smem[];
    float* ring = pl_smem;                                            // [SLOTS][256]
    float* tab = ring + PL_SLOTS * 256;                               // [GS][P * 12 | P * 16]
    float* mk = tab + PL_GS * TW;                                     // [GS][P * P]
    int* rcm = reinterpret_cast<int*>(mk + PL_GS * PP);               // [GS][P row masks | P column masks]
    int* desc = rcm + PL_GS * 2 * P;                                  // [SLOTS][touch, weight offsets]
    unsigned short* emap = reinterpret_cast<unsigned short*>(desc + 2 * PL_SLOTS);   // [GS * P * P]
    __shared__ int d_cnt[PL_GS];
    __shared__ int d_bins[PL_GS];
    __shared__ int f_ready[1];                                        // byte p: rounds published by producer p (mod 256)
    __shared__ int f_done[16];                                        // rounds consumed by consumer c
    // LDS-qualified volatile views (a generic volatile pointer makes the polls flat loads, which count on vmcnt too).
    // The four producers' counters share ONE word (byte stores): a consumer's poll is a single ds_read_b32 - with a word per
    // producer the polls of 12 spinning consumers alone kept the LDS pipeline busy.
    typedef volatile int __attribute__((address_space(3))) pl_flag;
    typedef volatile unsigned char __attribute__((address_space(3))) pl_flag8;
    pl_flag* const ready = (pl_flag*)f_ready;
    pl_flag8* const ready8 = (pl_flag8*)f_ready;
    pl_flag* const done = (pl_flag*)f_done;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int lin = blockIdx.x;
    const int slice = lin % n_slices;
    lin /= n_slices;
    const int groups_total = gridDim.x / (n_slices * n_regions);      // B * groups
    const int by = lin % groups_total;
    const int region = region_order.o[lin / groups_total];
    const int b = by % B, kgroup = by / B;
    const int y0 = (region / regions_x) * RG_RH, x0 = (region % regions_x) * RG_RW;
    const int OC = MASKCAT ? 2 * C : C;
    const int cbase = slice * 256;
#if CIM_ROI_PL_TRACE
    unsigned long long* pl_trace = reinterpret_cast<unsigned long long*>(partial + (size_t)(gridDim.x / (n_slices * n_regions * B)) * B * H * W * C);
    PL_STAMP(0);
    if (tid == 0) pl_trace[(size_t)blockIdx.x * 16 + 7] = ((unsigned long long)__builtin_amdgcn_s_getreg(0xF804) << 32) | __builtin_amdgcn_s_getreg(0xF814);
#endif

#define PL_CHUNK_PROLOGUE()                                                                                                \
        const int kb = (kgroup * chunks + ch) * PL_GS;                                                                   \
        if (kb >= K) break;                                            /* uniform */                                      \
        if (ch > 0) __syncthreads();                                   /* every wave is done with the previous chunk */   \
        const int total = pl_chunk_setup<MASKCAT, P>(tab, mk, rcm, emap, d_cnt, d_bins, f_ready, f_done, rec_all, masks, \
                                                     kb, K, H, W, b, y0, x0, tid, lane, wave, PL_TRACE_PTR);              \
        PL_STAMP(1 + 3 * ch);                                                                                            \
        if (lane == 0 && wave == 0 && CIM_ROI_PL_TRACE) reinterpret_cast<unsigned long long*>(partial + (size_t)(gridDim.x / (n_slices * n_regions * B)) * B * H * W * C)[(size_t)blockIdx.x * 16 + 6] = total; \
        if (total == 0) continue;                                      /* uniform */                                      \
        const int nrounds = (CIM_ROI_PL_EXP == 3) ? 0 : (total + PL_ROUND - 1) / PL_ROUND;

    if (wave >= PL_NCONS) {
        // ================= producer waves (no accumulators live here) =================
        const int p = wave - PL_NCONS;
        const int e = lane >> 4;
        const float* __restrict__ gc = grad_out + min(cbase + lane * 4, C - 4);
        for (int ch = 0; ch < chunks; ++ch) {
            PL_CHUNK_PROLOGUE()
            struct Batch { float4 lo[PL_BT], hi[PL_BT]; float m; int touch, woff; };
            auto prepare = [&](Batch& Bt, int round) {
                const int ge = round * PL_ROUND + p * PL_BT + e;
                int off = 0;
                Bt.m = 0.0f; Bt.touch = 0; Bt.woff = 0;
                if (ge < total) {
                    const int code = emap[ge];
                    const int lo = code & 63, ph = (code >> 6) & 7, pw = code >> 9;
                    off = (((kb + lo) * P + ph) * P + pw) * OC;        // < 2^31: checked by the launcher
                    if (MASKCAT) Bt.m = mk[lo * PP + ph * P + pw];
                    Bt.touch = rcm[lo * 2 * P + ph] | (rcm[lo * 2 * P + P + pw] << 16);
                    Bt.woff = (lo * TW + ph * RG_RH) | ((lo * TW + P * RG_RH + pw * RG_RW) << 16);
                }
#pragma unroll
                for (int i = 0; i < PL_BT; ++i) {                      // unconditional: entries past the end read offset 0
                    const int off_i = __builtin_amdgcn_readlane(off, i * 16);
                    if (CIM_ROI_PL_EXP == 2) { Bt.lo[i] = Bt.hi[i] = make_float4(0.f, 0.f, 0.f, (float)off_i); continue; }
                    Bt.lo[i] = *reinterpret_cast<const float4*>(gc + off_i);
                    if (MASKCAT) Bt.hi[i] = *reinterpret_cast<const float4*>(gc + off_i + C);
                }
            };
            auto finish = [&](Batch& Bt, int round) {
                if (round >= PL_RING) {                                // the slots' previous round consumed by every consumer?
                    const int need = round - PL_RING + 1;
                    while (true) {
                        const int d = (lane < PL_NCONS) ? done[lane] : need;
                        if (__ballot(d < need) == 0) break;
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
                asm volatile("" ::: "memory");
                const int sbase = (round % PL_RING) * PL_ROUND + p * PL_BT;
#pragma unroll
                for (int i = 0; i < PL_BT; ++i) {
                    float4 a = Bt.lo[i];
                    if (MASKCAT) {
                        const float m = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, Bt.m), i * 16));
                        a = make_float4(fmaf(m, Bt.hi[i].x, a.x), fmaf(m, Bt.hi[i].y, a.y), fmaf(m, Bt.hi[i].z, a.z), fmaf(m, Bt.hi[i].w, a.w));
                    }
                    *reinterpret_cast<float4*>(ring + (sbase + i) * 256 + lane * 4) = a;
                }
                if ((lane & 15) == 0) {
                    desc[2 * (sbase + e)] = Bt.touch;
                    desc[2 * (sbase + e) + 1] = Bt.woff;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) ready8[p] = (unsigned char)(round + 1);
            };
            Batch A, Bb;
            prepare(A, 0);
            for (int r = 0; r < nrounds; r += 2) {
                prepare(Bb, r + 1);                                    // in flight while round r is finished
                finish(A, r);
                prepare(A, r + 2);
                if (r + 1 < nrounds) finish(Bb, r + 1);
            }
            PL_STAMP(2 + 3 * ch);
        }
        return;
    }

    // ================= consumer waves =================
    // wave w runs on SIMD w % 4: the sub-blocks of one SIMD are spread out ((row t, column (s + 2 t) % 4): every 2 x 2
    // neighbourhood of sub-blocks sits on four different SIMDs), because a bin hits ADJACENT sub-blocks together - with
    // column = w % 4 a whole column of sub-blocks shared one SIMD's VALU while the other three idled
    const int wr = wave / PL_WC, wc = (wave + 2 * wr) % PL_WC;
    const int my_touch = (15 << (4 * wr)) | ((15 << (4 * wc)) << 16);
    ga_f2 accl[16], acch[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) accl[q] = acch[q] = ga_f2{0.f, 0.f};
    for (int ch = 0; ch < chunks; ++ch) {
        PL_CHUNK_PROLOGUE()
        // Steady state of a round: nothing waits.  `known` (rounds every producer has published) is refreshed by a poll
        // issued one round earlier, the descriptors of round r + 1 are read while round r is accumulated, and inside a
        // round the LDS operands of the next hit are in flight while the current hit's 40 packed FMAs run.
        struct Ops { float4 g, wy, wx; };
        auto load_ops = [&](Ops& o, int slot0, int j, int wo) {
            const int woj = __builtin_amdgcn_readlane(wo, j);
            o.g = *reinterpret_cast<const float4*>(ring + (slot0 + j) * 256 + lane * 4);
            o.wy = *reinterpret_cast<const float4*>(tab + (woj & 0xffff) + wr * 4);
            o.wx = *reinterpret_cast<const float4*>(tab + (woj >> 16) + wc * 4);
        };
        auto accumulate = [&](const Ops& o) {
            const ga_f2 gl = ga_lo(o.g), gh = ga_hi(o.g);
            const float wys[4] = {o.wy.x, o.wy.y, o.wy.z, o.wy.w};
            const float wxs[4] = {o.wx.x, o.wx.y, o.wx.z, o.wx.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const ga_f2 tl = ga_f2{wys[i], wys[i]} * gl, th = ga_f2{wys[i], wys[i]} * gh;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    accl[i * 4 + jj] = ga_fma(wxs[jj], tl, accl[i * 4 + jj]);
                    acch[i * 4 + jj] = ga_fma(wxs[jj], th, acch[i * 4 + jj]);
                }
            }
        };
        auto published = [&](unsigned r4) {                            // min over the four producers' bytes (< 256 rounds per chunk)
            return (int)min(min(r4 & 0xff, (r4 >> 8) & 0xff), min((r4 >> 16) & 0xff, r4 >> 24));
        };
        auto read_desc = [&](int round, int& t, int& wo) {
            const int slot0 = (round % PL_RING) * PL_ROUND;
            const int n_here = min(PL_ROUND, total - round * PL_ROUND);
            t = 0; wo = 0;
            if (lane < n_here) {
                t = desc[2 * (slot0 + lane)];
                wo = desc[2 * (slot0 + lane) + 1];
            }
        };
        int known = 0, t_nx = 0, wo_nx = 0;
        bool have_nx = false;
        for (int round = 0; round < nrounds; ++round) {
            int t, wo;
            if (have_nx) {
                t = t_nx; wo = wo_nx;
            } else {
                while (known <= round) {
                    known = published((unsigned)ready[0]);
                    if (known <= round) __builtin_amdgcn_s_sleep(2);
                }
                asm volatile("" ::: "memory");
                read_desc(round, t, wo);
            }
            const unsigned poll = (unsigned)ready[0];                  // consumed at the end of the round
            have_nx = (round + 1 < nrounds) && (known > round + 1);
            if (have_nx) read_desc(round + 1, t_nx, wo_nx);
            const int slot0 = (round % PL_RING) * PL_ROUND;
            const bool hit = ((t & my_touch & 0xffff) != 0) && (((t & my_touch) >> 16) != 0);
            unsigned long long todo = __ballot(hit);
            if (CIM_ROI_PL_EXP == 1) todo = 0;
            if (todo) {
                // two operand sets, loads unconditional (past the last hit the current one is read again): straight-line
                // code, so that the next set's ds_reads stay ABOVE the current set's FMAs and their uses below
                Ops A, Bo;
                int j = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                load_ops(A, slot0, j, wo);
                while (true) {
                    const bool more_b = todo != 0;
                    j = more_b ? __ffsll((long long)todo) - 1 : j;
                    todo &= todo - 1;
                    load_ops(Bo, slot0, j, wo);
                    __builtin_amdgcn_sched_barrier(0);
                    accumulate(A);
                    __builtin_amdgcn_sched_barrier(0);
                    if (!more_b) break;
                    const bool more_a = todo != 0;
                    j = more_a ? __ffsll((long long)todo) - 1 : j;
                    todo &= todo - 1;
                    load_ops(A, slot0, j, wo);
                    __builtin_amdgcn_sched_barrier(0);
                    accumulate(Bo);
                    __builtin_amdgcn_sched_barrier(0);
                    if (!more_a) break;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // this round's LDS reads have returned
            if (lane == 0) done[wave] = round + 1;
            known = max(known, published(poll));
        }
        PL_STAMP(2 + 3 * ch);
    }
#undef PL_CHUNK_PROLOGUE
    // ---- flush this wave's 4 x 4 pixels (zeros included: the partial maps are not cleared)
    const int cs = slice * 256 + lane * 4;
    const int sy0 = y0 + wr * 4, sx0 = x0 + wc * 4;
    if (cs < C) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (sy0 + i < H && sx0 + j < W) {
                    const size_t o = (((size_t)b * H + sy0 + i) * W + sx0 + j) * C + cs;
                    const float4 v = make_float4(accl[i * 4 + j].x, accl[i * 4 + j].y, acch[i * 4 + j].x, acch[i * 4 + j].y);
                    if (partial) {
                        float* dst = partial + (size_t)kgroup * B * H * W * C + o;
                        __builtin_nontemporal_store(v.x, dst);
                        __builtin_nontemporal_store(v.y, dst + 1);
                        __builtin_nontemporal_store(v.z, dst + 2);
                        __builtin_nontemporal_store(v.w, dst + 3);
                    } else {
                        *reinterpret_cast<float4*>(grad_in + o) = v;
                    }
                }
    }
    PL_STAMP(3);
}

// ---------------------------------------------------------------------------------------------
// Backward, pixel-owner form (default when the per-launch tables exist and the tile fits).
// Same ownership idea as above - a workgroup owns 16 channels of the whole map as an LDS tile - but
//   * a LANE owns one pixel of the ROI's bounding box with all 16 channels (16 accumulators), so the
//     index arithmetic, range unpacking and weight fetches are paid once per 16 channels, not per 4;
//   * the tile uses a 20-dword pixel stride: the read-modify-write ds_read/write_b128 of 16
//     neighbouring pixels hit 16 distinct bank quads; the gradient reads are wave-wide broadcasts;
//   * PX_DEPTH ROIs' gradient blocks and tables are in flight in registers (fetched while earlier ROIs
//     are accumulated): at 64 B granularity one ROI in flight per CU reads only ~1.1 TB/s;
//   * when at most 2 x 2 bins touch every pixel of the wave (bins >= ~2 px, the common case) all
//     weights and then all four gradient blocks are fetched back-to-back: 3 LDS round trips per pixel.
// grid = (C/16, RG); block = 512.
#ifndef CIM_ROI_EXP
#define CIM_ROI_EXP 0      // ablation switches for tools/bench_roi.py
#endif
constexpr int PX_THREADS = 512;
constexpr int PX_STRIDE = 20;      // dwords per tile pixel: 16 channels + 4 pad
constexpr int PX_DEPTH = 3;        // ROIs in flight (registers) per workgroup
constexpr int PX_MAXTAB = 2;       // staged table words per lane (ceil(((P+1)(H+W)+6)/512) must not exceed this)

static size_t bwd_px_lds(int H, int W, int P) {
    return sizeof(float) * ((size_t)H * W * PX_STRIDE + 2 * PX_DEPTH * ((size_t)P * P * 16 + (size_t)((P + 1) * (H + W) + 8)));
}

template <bool MASKCAT>
__global__ __launch_bounds__(PX_THREADS) void roi_align_bwd_px16_kernel(const float* __restrict__ grad_out,
                                                                        const float* __restrict__ masks,
                                                                        float* __restrict__ grad_in, int B, int C, int H,
                                                                        int W, int K, int P, int use_atomic,
                                                                        const float* __restrict__ pre) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int HW = H * W, PP = P * P;
    const int tabw = (P + 1) * (H + W);            // wy | wx | yr | xr
    const int stw = tabw + 6;                      // + box[4], count, batch
    const int recw = roi_rec_words(P, H, W);
    float* tile = lds;                             // [HW][PX_STRIDE]
    float* gbuf = tile + (size_t)HW * PX_STRIDE;   // 2 sets x PX_DEPTH slots x [PP][16]
    float* tbuf = gbuf + 2 * PX_DEPTH * PP * 16;   // 2 sets x PX_DEPTH slots x (tabw + 8)
    const int tstride = tabw + 8;
    const int tid = threadIdx.x;
    const int c0 = blockIdx.x * 16;
    const int OC = MASKCAT ? 2 * C : C;
    const int gbin = tid >> 2, gcg = tid & 3;      // staging role: gradient float4 (bin, channel quad)

    for (int b = 0; b < B; ++b) {
        for (int i = tid; i < HW * (PX_STRIDE / 4); i += PX_THREADS)
            reinterpret_cast<float4*>(tile)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        constexpr int D = PX_DEPTH;
        float4 sg[D], sg2[D];
        float sm[D], scount[D], stab[D][PX_MAXTAB];
        // branch-free (clamped indices, every lane loads): the D fetches must be ONE basic block, or hipcc
        // puts an s_waitcnt vmcnt(0) in front of each and serialises them ahead of the barrier
        const int gbin_c = min(gbin, PP - 1);
        auto fetch = [&](int k, int d) {
            k = min(k, K - 1);
            const float* rec = pre + (size_t)k * recw;
            scount[d] = rec[tabw + 4];
            const float* src = grad_out + ((size_t)k * PP + gbin_c) * OC + c0 + gcg * 4;
            sg[d] = *reinterpret_cast<const float4*>(src);
            if (MASKCAT) {
                sg2[d] = *reinterpret_cast<const float4*>(src + C);
                sm[d] = masks[(size_t)k * PP + gbin_c];
            }
#pragma unroll
            for (int j = 0; j < PX_MAXTAB; ++j) stab[d][j] = rec[min(tid + j * PX_THREADS, stw - 1)];
        };
        auto stage = [&](int slot, int d) {
            if (gbin < PP) {
                float4 v = sg[d];
                if (MASKCAT)
                    v = make_float4(v.x + sm[d] * sg2[d].x, v.y + sm[d] * sg2[d].y, v.z + sm[d] * sg2[d].z,
                                    v.w + sm[d] * sg2[d].w);
                const float c = scount[d];
                reinterpret_cast<float4*>(gbuf + slot * PP * 16)[tid] = make_float4(v.x / c, v.y / c, v.z / c, v.w / c);
            }
#pragma unroll
            for (int j = 0; j < PX_MAXTAB; ++j) {
                const int e = tid + j * PX_THREADS;
                if (e < stw) tbuf[slot * tstride + e] = stab[d][j];
            }
        };
        // Batches of D ROIs: stage the batch (fetched during the previous batch) into one of two slot sets,
        // issue the next batch's fetches, ONE barrier, accumulate the D ROIs.  Inside a batch the waves never
        // synchronise: wave w owns the tile rows y = w (mod 8), so no two waves touch the same pixel.
        const int kstep = gridDim.y;
        const int wave = tid >> 6, lane = tid & 63;
        int set = 0;
#pragma unroll
        for (int d = 0; d < D; ++d) fetch(blockIdx.y + d * kstep, d);
        for (int k0 = blockIdx.y; k0 < ((CIM_ROI_EXP == 5 || CIM_ROI_EXP == 6) ? 0 : K); k0 += D * kstep) {
#pragma unroll
            for (int d = 0; d < D; ++d)
                if (k0 + d * kstep < K) stage(set * D + d, d);
#pragma unroll
            for (int d = 0; d < D; ++d)
                if (CIM_ROI_EXP != 3) fetch(k0 + (D + d) * kstep, d);
            __syncthreads();
            for (int d = 0; d < D; ++d) {
                if (k0 + d * kstep >= K) break;
                const int slot = set * D + d;
                const float* wy = tbuf + slot * tstride;
                const float* wx = wy + P * H;
                const int* yr = reinterpret_cast<const int*>(wx + P * W);
                const int* xr = yr + H;
                const int* box = xr + W;
                const float* gb_ = gbuf + slot * PP * 16;
                if (box[5] != b) continue;             // block-uniform
                const int ylo = box[0], yhi = box[1], xlo = box[2], xhi = box[3];
                if (yhi < ylo || xhi < xlo) continue;
                const int ys = ylo + ((wave - ylo) & 7);                 // first row of this wave inside the box
                if (ys > yhi) continue;                                  // wave-uniform
                const int rw = xhi - xlo + 1, items = ((yhi - ys) / 8 + 1) * rw;
                const float inv_rw = 1.0f / (float)rw;
                for (int it0 = 0; it0 < (CIM_ROI_EXP == 1 ? 0 : items); it0 += 64) {
                    const int it = it0 + lane;
                    const bool live = it < items;
                    const int itc = live ? it : 0;
                    const int yy = (int)(((float)itc + 0.5f) * inv_rw);   // exact: the fraction is >= 0.5/rw off an integer
                    const int y = ys + 8 * yy, x = xlo + (itc - yy * rw);
                    const int ry = yr[y], rx = xr[x];
                    const bool on = live && !((ry | rx) & 0x10000);
                    const int phl = ry & 0xff, phh = (ry >> 8) & 0xff, pwl = rx & 0xff, pwh = (rx >> 8) & 0xff;
                    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
                    const bool small = !on || (phh - phl <= 1 && pwh - pwl <= 1);
                    if (__all(small)) {
                        const int p0 = on ? phl : 0, p1 = on ? phh : 0, q0 = on ? pwl : 0, q1 = on ? pwh : 0;
                        const float wy0 = on ? wy[p0 * H + y] : 0.0f, wy1 = (on && p1 != p0) ? wy[p1 * H + y] : 0.0f;
                        const float wx0 = wx[q0 * W + x], wx1 = (q1 != q0) ? wx[q1 * W + x] : 0.0f;
                        const float4* g00 = reinterpret_cast<const float4*>(gb_ + (p0 * P + q0) * 16);
                        const float4* g01 = reinterpret_cast<const float4*>(gb_ + (p0 * P + q1) * 16);
                        const float4* g10 = reinterpret_cast<const float4*>(gb_ + (p1 * P + q0) * 16);
                        const float4* g11 = reinterpret_cast<const float4*>(gb_ + (p1 * P + q1) * 16);
                        const float w00 = wy0 * wx0, w01 = wy0 * wx1, w10 = wy1 * wx0, w11 = wy1 * wx1;
#define PX_ACC(A, J)                                                                                          \
    {                                                                                                         \
        const float4 u = g00[J], v = g01[J], s_ = g10[J], t_ = g11[J];                                        \
        A.x = w00 * u.x + w01 * v.x + w10 * s_.x + w11 * t_.x;                                                \
        A.y = w00 * u.y + w01 * v.y + w10 * s_.y + w11 * t_.y;                                                \
        A.z = w00 * u.z + w01 * v.z + w10 * s_.z + w11 * t_.z;                                                \
        A.w = w00 * u.w + w01 * v.w + w10 * s_.w + w11 * t_.w;                                                \
    }
                        PX_ACC(a0, 0) PX_ACC(a1, 1) PX_ACC(a2, 2) PX_ACC(a3, 3)
#undef PX_ACC
                    } else if (on && CIM_ROI_EXP != 4) {
                        for (int ph = phl; ph <= phh; ++ph) {
                            const float a = wy[ph * H + y];
                            for (int pw = pwl; pw <= pwh; ++pw) {
                                const float w = a * wx[pw * W + x];
                                const float4* gv = reinterpret_cast<const float4*>(gb_ + (ph * P + pw) * 16);
                                const float4 g0 = gv[0], g1 = gv[1], g2 = gv[2], g3 = gv[3];
                                a0.x += w * g0.x; a0.y += w * g0.y; a0.z += w * g0.z; a0.w += w * g0.w;
                                a1.x += w * g1.x; a1.y += w * g1.y; a1.z += w * g1.z; a1.w += w * g1.w;
                                a2.x += w * g2.x; a2.y += w * g2.y; a2.z += w * g2.z; a2.w += w * g2.w;
                                a3.x += w * g3.x; a3.y += w * g3.y; a3.z += w * g3.z; a3.w += w * g3.w;
                            }
                        }
                    }
                    if (on) {
                        float4* t4 = reinterpret_cast<float4*>(tile + (size_t)(y * W + x) * PX_STRIDE);
                        float4 c0_ = t4[0], c1_ = t4[1], c2_ = t4[2], c3_ = t4[3];
                        c0_.x += a0.x; c0_.y += a0.y; c0_.z += a0.z; c0_.w += a0.w;
                        c1_.x += a1.x; c1_.y += a1.y; c1_.z += a1.z; c1_.w += a1.w;
                        c2_.x += a2.x; c2_.y += a2.y; c2_.z += a2.z; c2_.w += a2.w;
                        c3_.x += a3.x; c3_.y += a3.y; c3_.z += a3.z; c3_.w += a3.w;
                        t4[0] = c0_; t4[1] = c1_; t4[2] = c2_; t4[3] = c3_;
                    }
                }
            }
            set ^= 1;
        }
        __syncthreads();
        float* gb = grad_in + (size_t)b * HW * C;
        for (int i = tid; i < ((CIM_ROI_EXP == 2 || CIM_ROI_EXP == 6) ? 0 : HW * 4); i += PX_THREADS) {
            const int pix = i >> 2, cg = i & 3;
            const float4 v = *reinterpret_cast<const float4*>(tile + (size_t)pix * PX_STRIDE + cg * 4);
            float* dst = gb + (size_t)pix * C + c0 + cg * 4;
            if (use_atomic) {
                atomicAdd(dst + 0, v.x); atomicAdd(dst + 1, v.y); atomicAdd(dst + 2, v.z); atomicAdd(dst + 3, v.w);
            } else {
                *reinterpret_cast<float4*>(dst) = v;
            }
        }
        __syncthreads();
    }
}

// LDS bytes of the tile kernel for a channel chunk of CH
static size_t bwd_tile_lds(int CH, int H, int W, int P) {
    return sizeof(float) * ((size_t)H * W * CH + 2 * (size_t)P * P * CH + 2 * (size_t)(P + 1) * (H + W));
}

template <int CH, bool MASKCAT>
static int launch_bwd_tile(const float* go, const float* rois, const float* masks, float* gin, int B, int C, int H,
                           int W, int K, int P, float scale, int sr, int aligned, float* ws, hipStream_t st) {
    const size_t lds = bwd_tile_lds(CH, H, W, P);
    auto kern = ws ? roi_align_bwd_tile_kernel<CH, MASKCAT, true> : roi_align_bwd_tile_kernel<CH, MASKCAT, false>;
    if (ws) hipLaunchKernelGGL(roi_tables_kernel, dim3(K), dim3(256), sizeof(float) * roi_rec_words(P, H, W), st, rois, ws, K, P, H, W, scale, sr, aligned);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    const int chunks = C / CH;
    int rg = (256 + chunks - 1) / chunks;          // >= one workgroup per CU (1 resident per CU at ~96 KB LDS)
    if (rg > K) rg = K;
    if (rg < 1) rg = 1;
    if (rg > 1) {
        hipError_t e = hipMemsetAsync(gin, 0, sizeof(float) * (size_t)B * H * W * C, st);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kern, dim3(chunks, rg), dim3(TILE_THREADS), lds, st, go, rois, masks, gin, B, C, H, W, K, P, scale,
                       sr, aligned, rg > 1 ? 1 : 0, ws);
    return 0;
}

template <bool MASKCAT>
int launch_fwd(const float* feat, const float* rois, const float* masks, float* out, int B, int C, int H, int W, int K,
               int P, float scale, int sr, int aligned, hipStream_t st, float* ws = nullptr) {
    if (K == 0) return 0;
    dim3 grid(K, P), block(256);
    // aggregated-weight kernel: needs the table workspace, 16-byte channel rows and bins of <= AG_MAXE pixels
    // (rows, columns per bin <= size/P + 3; maps up to ~35 x 49 at P = 7)
    if (ws != nullptr && C % 4 == 0 && P <= FW_MAXP && (long long)H * W * C < (1ll << 30) && H <= 64 &&
        getenv("CIM_ROI_FWD_EXACT") == nullptr) {
        hipLaunchKernelGGL(roi_tables_kernel, dim3(K), dim3(256), sizeof(float) * roi_rec_words(P, H, W), st, rois, ws, K, P, H, W, scale, sr, aligned);
        if (CIM_ROI_FROW && P <= 7 && W <= RS_MAXD && getenv("CIM_ROI_FWD_LIST") == nullptr) {
            const int nth = C >= 1024 ? 256 : ((C / 4 + 63) / 64) * 64;       // narrow maps (VGG: 512 channels): no idle waves
            if (CIM_ROI_FROW2)
                hipLaunchKernelGGL((roi_align_fwd_rowsum2_kernel<MASKCAT>), dim3(K, (P + 1) / 2), dim3(nth), 0, st, feat, masks, out, C, H, W, P, ws);
            else
                hipLaunchKernelGGL((roi_align_fwd_rowsum_kernel<MASKCAT>), dim3(K, P), dim3(nth), 0, st, feat, masks, out, C, H, W, P, ws);
            return 0;
        }
        const int fz = (C >= 512 * CIM_ROI_FZ) ? CIM_ROI_FZ : 1;
        hipLaunchKernelGGL((roi_align_fwd_agg_kernel<MASKCAT>), dim3(K, P, fz), dim3(fz > 1 ? 128 : 256), 0, st, feat, masks, out,
                           C, H, W, P, ws, rois, scale, sr, aligned);
        return 0;
    }
    if (C % 4 == 0 && P <= FW_MAXP && (long long)H * W * C < (1ll << 30) && getenv("CIM_ROI_FWD_DIRECT") == nullptr)
        hipLaunchKernelGGL((roi_align_fwd_tab_kernel<MASKCAT>), grid, block, 0, st, feat, rois, masks, out, C, H, W, P,
                           scale, sr, aligned);
    else if (C % 4 == 0)
        hipLaunchKernelGGL((roi_align_fwd_kernel<4, MASKCAT>), grid, block, 0, st, feat, rois, masks, out, C, H, W, P,
                           scale, sr, aligned);
    else
        hipLaunchKernelGGL((roi_align_fwd_kernel<1, MASKCAT>), grid, block, 0, st, feat, rois, masks, out, C, H, W, P,
                           scale, sr, aligned);
    return 0;
}

template <bool MASKCAT>
static int launch_bwd_px16(const float* go, const float* rois, const float* masks, float* gin, int B, int C, int H, int W,
                           int K, int P, float scale, int sr, int aligned, float* ws, hipStream_t st) {
    const size_t lds = bwd_px_lds(H, W, P);
    auto kern = roi_align_bwd_px16_kernel<MASKCAT>;
    hipLaunchKernelGGL(roi_tables_kernel, dim3(K), dim3(256), sizeof(float) * roi_rec_words(P, H, W), st, rois, ws, K, P, H, W, scale, sr, aligned);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    const int chunks = C / 16;
    int rg = (256 + chunks - 1) / chunks;
    if (rg > K) rg = K;
    if (rg < 1) rg = 1;
    if (rg > 1) {
        e = hipMemsetAsync(gin, 0, sizeof(float) * (size_t)B * H * W * C, st);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kern, dim3(chunks, rg), dim3(PX_THREADS), lds, st, go, masks, gin, B, C, H, W, K, P, rg > 1 ? 1 : 0,
                       ws);
    return 0;
}

#ifndef CIM_ROI_GPAD
#define CIM_ROI_GPAD 0
#endif
// 3 x 4 pixel blocks, 128 ROIs per group (2 workgroups per CU at 56 KB of LDS): 0.258 ms at cfg2.  Measured alternatives
// (tools/bench_roi.py): 2x4/256 0.310, 2x4/128 capped to 2 per CU 0.258, 4x4/128 0.282, 2x8/128 0.290, 3x3/128 0.272,
// 3x5/128 0.278, 3x6/128 0.407, 4x8/64 0.462; 3x4 with 64 / 96 / 144 / 168 / 200 / 256 ROIs: 0.358 / 0.291 / 0.257 / 0.264 /
// 0.293 / 0.281; 512- or 256-entry windows (4-5 workgroups per CU): 0.40-0.42; one workgroup per CU: 0.273.
#ifndef CIM_ROI_GH
#define CIM_ROI_GH 3
#endif
#ifndef CIM_ROI_GW
#define CIM_ROI_GW 4
#endif
template <bool MASKCAT>
static int launch_bwd_gather(const float* go, const float* rois, const float* masks, float* gin, int B, int C, int H, int W,
                             int K, int P, float scale, int sr, int aligned, float* ws, hipStream_t st, int tables_ready) {
    constexpr int GH = CIM_ROI_GH, GW = CIM_ROI_GW;
    if (!tables_ready)
        hipLaunchKernelGGL(roi_tables_kernel, dim3(K), dim3(256), sizeof(float) * roi_rec_words(P, H, W), st, rois, ws, K, P, H, W, scale, sr, aligned);
    const int tiles = ((H + GH - 1) / GH) * ((W + GW - 1) / GW);
    const int groups = (K + GA_GS - 1) / GA_GS;
    if (groups > 1) {
        hipError_t e = hipMemsetAsync(gin, 0, sizeof(float) * (size_t)B * H * W * C, st);
        if (e != hipSuccess) return (int)e;
    }
    // CIM_ROI_GPAD: unused dynamic LDS, only there to cap the workgroups per CU (fewer resident workgroups re-read less:
    // the neighbouring tiles that share gradient vectors then run closer together in time)
    auto kern = roi_align_bwd_gather_kernel<GH, GW, MASKCAT>;
    if (CIM_ROI_GPAD > 0) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, CIM_ROI_GPAD);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kern, dim3(tiles, B * groups, (C + 1023) / 1024), dim3(256), CIM_ROI_GPAD,
                       st, go, masks, gin, C, H, W, K, P, B, groups > 1 ? 1 : 0, ws);
    return 0;
}

// Region order, centre first (the centre regions carry the most entries: longest jobs first), computed on the host
// and passed by value.
static RegionOrder region_order(int ry, int rx) {
    RegionOrder ro;
    const int n = ry * rx;
    int idx[RG_MAXREG];
    float d[RG_MAXREG];
    for (int r = 0; r < n; ++r) {
        const float dy = (r / rx) + 0.5f - 0.5f * ry, dx = (r % rx) + 0.5f - 0.5f * rx;
        idx[r] = r;
        d[r] = dy * dy * (float)(RG_RH * RG_RH) + dx * dx * (float)(RG_RW * RG_RW);
    }
    for (int i = 0; i < n; ++i) {
        int best = i;
        for (int j = i + 1; j < n; ++j)
            if (d[idx[j]] < d[idx[best]]) best = j;
        const int t = idx[i]; idx[i] = idx[best]; idx[best] = t;
        ro.o[i] = (unsigned char)idx[i];
    }
    return ro;
}

// grad_in = sum over the ROI groups' partial maps: n4 float4 per map, groups maps `stride4` float4 apart.
__global__ __launch_bounds__(256) void roi_partial_reduce_kernel(const float4* __restrict__ partial, float4* __restrict__ out,
                                                                 size_t n4, int groups) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    typedef float rp_f4 __attribute__((ext_vector_type(4)));
    const rp_f4* p = reinterpret_cast<const rp_f4*>(partial) + i;
    rp_f4 acc = __builtin_nontemporal_load(p);
    for (int g = 1; g < groups; ++g) acc += __builtin_nontemporal_load(p + (size_t)g * n4);
    out[i] = make_float4(acc.x, acc.y, acc.z, acc.w);
}

template <bool MASKCAT>
static int launch_bwd_region(const float* go, const float* rois, const float* masks, float* gin, int B, int C, int H, int W,
                             int K, int P, float scale, int sr, int aligned, float* ws, hipStream_t st, int tables_ready,
                             float* scratch) {
    if (!tables_ready)
        hipLaunchKernelGGL(roi_tables_kernel, dim3(K), dim3(256), sizeof(float) * roi_rec_words(P, H, W), st, rois, ws, K, P, H, W, scale, sr, aligned);
    const int ry = (H + RG_RH - 1) / RG_RH, rx = (W + RG_RW - 1) / RG_RW, n_regions = ry * rx;
    const RegionOrder order = region_order(ry, rx);
    const int GS = rg_group_size(K, B, C, H, W);
    const int groups = (K + GS - 1) / GS, n_slices = (C + 255) / 256;
    float* partial = (groups > 1) ? scratch : nullptr;      // without scratch the groups meet through atomicAdd (slow)
    if (groups > 1 && !partial) {
        hipError_t e = hipMemsetAsync(gin, 0, sizeof(float) * (size_t)B * H * W * C, st);
        if (e != hipSuccess) return (int)e;
    }
    auto kern = GS == 128 ? roi_align_bwd_region_kernel<MASKCAT, 128> : roi_align_bwd_region_kernel<MASKCAT, 64>;
    const size_t lds = rg_lds_bytes(GS, P);
    hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (ea != hipSuccess) return (int)ea;
    hipLaunchKernelGGL(kern, dim3((unsigned)(n_slices * B * groups * n_regions)), dim3(RG_NT), lds, st, go, masks, gin, C, H, W, K, P, B,
                       groups > 1 ? 1 : 0, ws, order, n_regions, rx, n_slices, partial);
    if (partial) {
        const size_t n4 = (size_t)B * H * W * C / 4;
        hipLaunchKernelGGL(roi_partial_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st,
                           reinterpret_cast<const float4*>(partial), reinterpret_cast<float4*>(gin), n4, groups);
    }
    return 0;
}

// pipelined region form: same grid, scratch and reduce as launch_bwd_region
template <bool MASKCAT>
static int launch_bwd_pipe(const float* go, const float* rois, const float* masks, float* gin, int B, int C, int H, int W,
                           int K, float scale, int sr, int aligned, float* ws, hipStream_t st, int tables_ready, float* scratch) {
    constexpr int P = 7;
    if (!tables_ready)
        hipLaunchKernelGGL(roi_tables_kernel, dim3(K), dim3(256), sizeof(float) * roi_rec_words(P, H, W), st, rois, ws, K, P, H, W, scale, sr, aligned);
    const int ry = (H + RG_RH - 1) / RG_RH, rx = (W + RG_RW - 1) / RG_RW, n_regions = ry * rx;
    const RegionOrder order = region_order(ry, rx);
    const int chunks = pl_chunks(K, B, C, H, W);
    const int groups = (K + PL_GS * chunks - 1) / (PL_GS * chunks), n_slices = (C + 255) / 256;
    float* partial = (groups > 1) ? scratch : nullptr;
    auto kern = roi_align_bwd_pipe_kernel<MASKCAT, P>;
    const size_t lds = pl_lds_bytes<P>();
    hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (ea != hipSuccess) return (int)ea;
    hipLaunchKernelGGL(kern, dim3((unsigned)(n_slices * B * groups * n_regions)), dim3(RG_NT), lds, st, go, masks, gin, C, H, W, K, B,
                       chunks, ws, order, n_regions, rx, n_slices, partial);
    if (partial) {
        const size_t n4 = (size_t)B * H * W * C / 4;
        hipLaunchKernelGGL(roi_partial_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st,
                           reinterpret_cast<const float4*>(partial), reinterpret_cast<float4*>(gin), n4, groups);
    }
    return 0;
}

template <bool MASKCAT>
int launch_bwd(const float* go, const float* rois, const float* masks, float* gin, int B, int C, int H, int W, int K,
               int P, float scale, int sr, int aligned, float* ws, hipStream_t st, int tables_ready = 0, float* scratch = nullptr) {
    const size_t budget = 150 * 1024;
    // gather form: 256 lanes x 4 channels per workgroup (grid.z chunks of 1024 channels), 8-bit bin indices in the
    // packed ranges, 32-bit element offsets
    if (K > 0 && ws != nullptr && C % 4 == 0 && P <= 16 && H < 256 && W < 256 && (long long)B * ((K + GA_GS - 1) / GA_GS) <= 65535 &&
        (long long)K * P * P * (MASKCAT ? 2 : 1) * C < (1ll << 31) && getenv("CIM_ROI_BWD_TILE") == nullptr &&
        getenv("CIM_ROI_BWD_PX16") == nullptr) {
        // region form needs the partial-map scratch (or a single ROI group); without it the gather form's fewer atomics win
        if (getenv("CIM_ROI_BWD_GATHER") == nullptr && (scratch != nullptr || K <= rg_group_size(K, B, C, H, W)) && rg_lds_bytes(rg_group_size(K, B, C, H, W), P) <= 158 * 1024 && ((H + RG_RH - 1) / RG_RH) * ((W + RG_RW - 1) / RG_RW) <= RG_MAXREG &&
            (long long)B * ((K + 63) / 64) * ((C + 255) / 256) * RG_MAXREG < (1ll << 31)) {
            // pipelined form (opt-in, CIM_ROI_BWD_PIPE=1: measured no faster, see its header): P = 7, partial maps (or one group)
            if (P == 7 && (scratch != nullptr || K <= PL_GS * pl_chunks(K, B, C, H, W)) && getenv("CIM_ROI_BWD_PIPE") != nullptr)
                return launch_bwd_pipe<MASKCAT>(go, rois, masks, gin, B, C, H, W, K, scale, sr, aligned, ws, st, tables_ready, scratch);
            return launch_bwd_region<MASKCAT>(go, rois, masks, gin, B, C, H, W, K, P, scale, sr, aligned, ws, st, tables_ready, scratch);
        }
        return launch_bwd_gather<MASKCAT>(go, rois, masks, gin, B, C, H, W, K, P, scale, sr, aligned, ws, st, tables_ready);
    }
    if (K > 0 && ws != nullptr && C % 16 == 0 && P * P * 4 <= PX_THREADS && H < 256 && W < 256 &&
        (P + 1) * (H + W) + 6 <= PX_MAXTAB * PX_THREADS && bwd_px_lds(H, W, P) <= 160 * 1024 - 512 &&
        getenv("CIM_ROI_BWD_TILE") == nullptr)
        return launch_bwd_px16<MASKCAT>(go, rois, masks, gin, B, C, H, W, K, P, scale, sr, aligned, ws, st);
    if (K > 0 && P <= 16 && C % 4 == 0) {
        if (C % 16 == 0 && bwd_tile_lds(16, H, W, P) <= budget)
            return launch_bwd_tile<16, MASKCAT>(go, rois, masks, gin, B, C, H, W, K, P, scale, sr, aligned, ws, st);
        if (C % 8 == 0 && bwd_tile_lds(8, H, W, P) <= budget)
            return launch_bwd_tile<8, MASKCAT>(go, rois, masks, gin, B, C, H, W, K, P, scale, sr, aligned, ws, st);
        if (bwd_tile_lds(4, H, W, P) <= budget)
            return launch_bwd_tile<4, MASKCAT>(go, rois, masks, gin, B, C, H, W, K, P, scale, sr, aligned, ws, st);
    }
    // generic fallback (odd channel counts, maps too large for an LDS tile): global atomics
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

extern "C" int cim_roi_align_fwd_ws(const float* feat, const float* rois, float* out, int B, int C, int H, int W, int K,
                                    int P, float spatial_scale, int sampling_ratio, int aligned, float* workspace,
                                    void* stream) {
    ROI_ARGS_OK();
    CIM_CHECK_ARG(feat && (out || K == 0));
    int rc = launch_fwd<false>(feat, rois, nullptr, out, B, C, H, W, K, P, spatial_scale, sampling_ratio, aligned,
                               cim::as_stream(stream), workspace);
    if (rc) return rc;
    CIM_CHECK_LAUNCH();
    return 0;
}

extern "C" int cim_roi_align_maskcat_fwd_ws(const float* feat, const float* rois, const float* masks, float* cat, int B,
                                            int C, int H, int W, int K, int P, float spatial_scale, int sampling_ratio,
                                            int aligned, float* workspace, void* stream) {
    ROI_ARGS_OK();
    CIM_CHECK_ARG(feat && ((cat && masks) || K == 0));
    int rc = launch_fwd<true>(feat, rois, masks, cat, B, C, H, W, K, P, spatial_scale, sampling_ratio, aligned,
                              cim::as_stream(stream), workspace);
    if (rc) return rc;
    CIM_CHECK_LAUNCH();
    return 0;
}

extern "C" long long cim_roi_align_bwd_workspace(int K, int P, int H, int W) {
    return (long long)sizeof(float) * (long long)K * roi_rec_words(P, H, W);
}

extern "C" int cim_roi_align_bwd(const float* grad_out, const float* rois, float* grad_in, int B, int C, int H, int W,
                                 int K, int P, float spatial_scale, int sampling_ratio, int aligned, float* workspace,
                                 void* stream) {
    ROI_ARGS_OK();
    CIM_CHECK_ARG(grad_in && (grad_out || K == 0));
    int rc = launch_bwd<false>(grad_out, rois, nullptr, grad_in, B, C, H, W, K, P, spatial_scale, sampling_ratio,
                               aligned, workspace, cim::as_stream(stream));
    if (rc) return rc;
    CIM_CHECK_LAUNCH();
    return 0;
}

extern "C" long long cim_roi_align_bwd_scratch(int K, int B, int C, int H, int W) {
    const int GS = rg_group_size(K, B, C, H, W);
    const long long groups = (K + GS - 1) / GS;
    return groups > 1 ? (long long)sizeof(float) * groups * B * H * W * C : 0;
}

extern "C" int cim_roi_align_bwd_ws(const float* grad_out, const float* rois, float* grad_in, int B, int C, int H, int W,
                                    int K, int P, float spatial_scale, int sampling_ratio, int aligned, float* workspace,
                                    int tables_ready, float* scratch, void* stream) {
    ROI_ARGS_OK();
    CIM_CHECK_ARG(grad_in && (grad_out || K == 0) && (workspace || !tables_ready));
    int rc = launch_bwd<false>(grad_out, rois, nullptr, grad_in, B, C, H, W, K, P, spatial_scale, sampling_ratio,
                               aligned, workspace, cim::as_stream(stream), tables_ready, scratch);
    if (rc) return rc;
    CIM_CHECK_LAUNCH();
    return 0;
}

extern "C" int cim_roi_align_maskcat_bwd_ws(const float* grad_cat, const float* rois, const float* masks, float* grad_in,
                                            int B, int C, int H, int W, int K, int P, float spatial_scale,
                                            int sampling_ratio, int aligned, float* workspace, int tables_ready,
                                            float* scratch, void* stream) {
    ROI_ARGS_OK();
    CIM_CHECK_ARG(grad_in && ((grad_cat && masks) || K == 0) && (workspace || !tables_ready));
    int rc = launch_bwd<true>(grad_cat, rois, masks, grad_in, B, C, H, W, K, P, spatial_scale, sampling_ratio, aligned,
                              workspace, cim::as_stream(stream), tables_ready, scratch);
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
                                         int sampling_ratio, int aligned, float* workspace, void* stream) {
    ROI_ARGS_OK();
    CIM_CHECK_ARG(grad_in && ((grad_cat && masks) || K == 0));
    int rc = launch_bwd<true>(grad_cat, rois, masks, grad_in, B, C, H, W, K, P, spatial_scale, sampling_ratio, aligned,
                              workspace, cim::as_stream(stream));
    if (rc) return rc;
    CIM_CHECK_LAUNCH();
    return 0;
}
