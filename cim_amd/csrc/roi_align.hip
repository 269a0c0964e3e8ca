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
// Per-ROI tables (shared by the table-driven forward and the region backward).
//
// ROIAlign is linear and separable: with per-ROI weight tables
//   WY[ph][y] = sum over the bin-row's y-samples of the bilinear weight that sample puts on row y
//   WX[pw][x] = likewise for columns
// (a sample is dropped when either coordinate is out of range, which factorises too),
//   out[ph,pw,c] = (1/count) sum_y sum_x WY[ph][y] WX[pw][x] feat[y,x,c]
//   grad_in[y,x,c] += (1/count) * sum_ph WY[ph][y] * sum_pw WX[pw][x] * g[ph,pw,c].
// (Rounds 1-3 carried five more backward kernels built on them - an LDS-resident whole-map tile, a 16-channel pixel-owner form,
// a 3 x 4 pixel-block gather, and a producer / consumer variant of the region form - and two more forwards; all measured slower
// than the kernels below and removed in round 4: DESIGN.md section 4.2 keeps their numbers.)

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
constexpr int FW_MAXP = 8;      // bins per axis the table-driven forwards take
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

// Forward, row-sum form of the same separable sum (default when P <= 7 and the map is at most 128 x 128):
//   out[ph,pw,c] = sum_x WX[pw][x] * ( sum_y WY[ph][y] / count * feat[y,x,c] ).
// A workgroup still owns one (roi, bin row), but walks the COLUMNS of the ROI once: the inner sum t(x) over the bin row's
// rows is formed once per column and fed to every bin whose WX[pw][x] is non-zero, so the pixels that neighbouring bins
// share (one or two columns per bin boundary) are loaded once - rows x (columns of the ROI) loads instead of
// rows x (sum of the bins' column counts), ~23 % fewer at the benchmark's ROI sizes.  Two columns are in flight per
// iteration (up to 8 loads per lane).  The 7 column weights of a column sit in LDS as one padded row (zero outside the
// bin's range), so the bin update is 7 unconditional packed FMAs.
constexpr int RS_MAXD = 128;     // rows / columns of the map (round 6: was 64 - the reference's largest training scale gives 57 x 75 maps)
// Round 4, measured and dropped: workgroup = one ROI x one 256-channel slice (four waves = the four bin-row pairs) with the slices
// pinned to XCDs, so that an XCD's L2 only sees its 1.45 MB of the 5.8 MB map (the calibrated FETCH_SIZE of this launch is 232 MB
// for that map: 38x).  0.152 vs 0.140 ms at cfg2, 0.287 vs 0.279 at 2000 ROIs (tools/bench_roi.py, same box): the re-fetches are
// served by the Infinity Cache and were not what bounds the kernel - the L1 -> L2 request rate is (section 4.2 of DESIGN.md).
// Two bin rows per workgroup: the rows of the map that bin rows ph and ph + 1 share (one or two of ~3.6)
// are loaded once as well; grid = (K, ceil(P / 2)).
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

// ---------------------------------------------------------------------------------------------
// Forward, fused with the MaskFuse prologue AND the Winograd input transform of MaskFuse's convolution (round 5):
//   feat --ROIAlign--> box [7][7][C] --(box, box * mask)--> cat [7][7][2C] --B^T d B per tile--> V' [121][Rs][2C] pair image
// for /root/reference/lib/modeling/resnet50.py:121-135 (roi_xform -> mask multiply -> concat -> mask_branch's 3 x 3 convolution in
// the mixed 4 + 3 Winograd tiling of csrc/winograd.hip).  `cat` never exists: the two-kernel path wrote its 401 MB (cfg2), read
// them back in wino7_input_pair_kernel (0.288 ms) and wrote the 991 MB image; here the image is the only output.
// Workgroup = (ROI r, 256-channel slice), four waves.  Phase A: wave w computes the bin rows 2w, 2w + 1 of the ROI for the lane's
// 4 channels with the row-sum arithmetic of roi_align_fwd_rowsum2_kernel (the same operations in the same order: the values are
// the bits that kernel writes) into an LDS patch [49][256].  Phase B: wave t transforms tile type t (36 / 30 / 30 / 25 positions)
// of the patch - once as it is, once multiplied by the ROI's 7 x 7 mask - with the arithmetic of w7_input_tile and stores the
// split (h, l) fp16 chunks (w7_store_pair's protocol: lane pairs exchange halves, 16 bytes per lane).  Rows r >= K (the pad up
// to a multiple of 32 rows the GEMM wants) are zero-filled.  The image is bit-identical to the two-kernel path's.
#include "wino43_mats.h"
namespace w7f {
constexpr int NP[2] = {6, 5}, IN0[2] = {-1, 3}, QOFF[4] = {0, 36, 66, 96};
typedef unsigned u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned swap1(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false);      // quad_perm [1,0,3,2]
}
__device__ __forceinline__ void store_pair(float* p, const float4& v, float s) {         // (= w7_store_pair of winograd.hip)
    unsigned h0, l0, h1, l1;
    cim::pair_split2(v.x * s, v.y * s, h0, l0);
    cim::pair_split2(v.z * s, v.w * s, h1, l1);
    const bool odd = (threadIdx.x & 1) != 0;
    const unsigned r0 = swap1(odd ? h0 : l0), r1 = swap1(odd ? h1 : l1);
    const u4 o = odd ? u4{r0, r1, l0, l1} : u4{h0, h1, r0, r1};
    __builtin_nontemporal_store(o, reinterpret_cast<u4*>(p));
}
__device__ __forceinline__ void fma4(float4& a, float s, const float4& v) {
    a.x = fmaf(s, v.x, a.x); a.y = fmaf(s, v.y, a.y); a.z = fmaf(s, v.z, a.z); a.w = fmaf(s, v.w, a.w);
}
// one tile type of the patch: d = patch (x mask when MASKED) -> B^T d B -> pair image positions Q0 .. Q0 + NA NB
template <int KA, int KB, bool MASKED>
__device__ __forceinline__ void tile(const float4* __restrict__ patch, const float* __restrict__ mask49, float* __restrict__ V,
                                     size_t MC, size_t off, const float* __restrict__ scale, int lane) {
    constexpr int NA = NP[KA], NB = NP[KB], P = 7, Q0 = QOFF[KA * 2 + KB];
    float4 d[NA][NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int iy = IN0[KA] + i;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int ix = IN0[KB] + j;
            if ((unsigned)iy < (unsigned)P && (unsigned)ix < (unsigned)P) {
                d[i][j] = patch[(iy * P + ix) * 64 + lane];
                if (MASKED) { const float m = mask49[iy * P + ix]; d[i][j] = make_float4(m * d[i][j].x, m * d[i][j].y, m * d[i][j].z, m * d[i][j].w); }
            } else {
                d[i][j] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        float4 trow[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            trow[j] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int k = 0; k < NA; ++k)
                if (W7_BT[KA][i][k] != 0.0f) fma4(trow[j], W7_BT[KA][i][k], d[k][j]);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int k = 0; k < NB; ++k)
                if (W7_BT[KB][j][k] != 0.0f) fma4(v, W7_BT[KB][j][k], trow[k]);
            store_pair(V + (size_t)(Q0 + i * NB + j) * MC + off, v, scale[Q0 + i * NB + j]);
        }
    }
}
}  // namespace w7f

__global__ __launch_bounds__(256) void roi_align_wino7_pair_kernel(const float* __restrict__ feat, const float* __restrict__ masks,
                                                                   float* __restrict__ V, int C, int H, int W, int K, int Rs,
                                                                   const float* __restrict__ rec_all,
                                                                   const float* __restrict__ scale) {
    constexpr int P = 7;
    __shared__ __attribute__((aligned(16))) float4 s_patch[49 * 64];                    // [bin][lane]: 49 KB
    __shared__ __attribute__((aligned(16))) float s_wx[RS_MAXD][8];
    __shared__ float s_wa[4][RS_MAXD], s_wb[4][RS_MAXD];
    __shared__ int s_rows[4][RS_MAXD];
    __shared__ float s_mask[49];
    const int r = blockIdx.x, slice = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = slice * 256 + lane * 4;
    const size_t C2 = 2 * (size_t)C, MC = (size_t)Rs * C2;
    if (r >= K) {                                                                        // pad rows: zero
        if (c < C)
            for (int q = wave; q < 121; q += 4) {
                float* dst = V + (size_t)q * MC + (size_t)r * C2 + c;
                __builtin_nontemporal_store(w7f::u4{0, 0, 0, 0}, reinterpret_cast<w7f::u4*>(dst));
                __builtin_nontemporal_store(w7f::u4{0, 0, 0, 0}, reinterpret_cast<w7f::u4*>(dst + C));
            }
        return;
    }
    const int k = r;
    const float* rec = rec_all + (size_t)k * roi_rec_words(P, H, W);
    const float* wx = rec + P * H;
    const int* box = reinterpret_cast<const int*>(rec + (P + 1) * (H + W));
    const int ylo = box[0], yhi = box[1], xlo = box[2], xhi = box[3];
    const float inv_count = 1.0f / reinterpret_cast<const float*>(box)[4];
    const float* __restrict__ fb = feat + (size_t)box[5] * H * W * C;
    const int ncols = max(xhi - xlo + 1, 0);
    // ---- phase A: bin rows ph0, ph0 + 1 of this wave (rowsum2's arithmetic)
    const int ph0 = wave * 2;
    const bool two = ph0 + 1 < P;
    const float* wy0 = rec + ph0 * H;
    const float* wy1 = wy0 + H;
    int nrows;
    {
        // the rows of the map with a non-zero weight in bin row ph0 or ph0 + 1, ascending (a wave-wide compaction of the loop
        // rowsum2's thread 0 runs: same list, same order)
        nrows = 0;
        for (int y0 = ylo; y0 <= yhi; y0 += 64) {             // (an ROI spans up to RS_MAXD = 128 rows: two passes of the wave)
            const int y = y0 + lane;
            float a = 0.0f, b = 0.0f;
            if (y <= yhi) { a = wy0[y]; b = two ? wy1[y] : 0.0f; }
            const bool on = y <= yhi && (a != 0.0f || b != 0.0f);
            const unsigned long long m = __ballot(on);
            if (on) {
                const int n = nrows + __popcll(m & ((1ull << lane) - 1ull));
                s_rows[wave][n] = y * W; s_wa[wave][n] = a * inv_count; s_wb[wave][n] = b * inv_count;
            }
            nrows += __popcll(m);
        }
    }
    for (int e = tid; e < ncols * 8; e += 256) {
        const int xi = e >> 3, pw = e & 7;
        s_wx[xi][pw] = pw < P ? wx[pw * W + xlo + xi] : 0.0f;
    }
    if (tid < 49) s_mask[tid] = masks[(size_t)k * 49 + tid];
    __syncthreads();
    if (c < C) {
        const int* __restrict__ rows = s_rows[wave];
        const float* __restrict__ rwa = s_wa[wave];
        const float* __restrict__ rwb = s_wb[wave];
        const float* __restrict__ fc = fb + c;
        ga_f2 al[7], ah[7], bl[7], bh[7];
#pragma unroll
        for (int pw = 0; pw < 7; ++pw) al[pw] = ah[pw] = bl[pw] = bh[pw] = ga_f2{0.f, 0.f};
        for (int xi = 0; xi < ncols; xi += 2) {
            const bool pair = xi + 1 < ncols;            // wave-uniform; the odd last column is loaded twice, weight 0
            const int x0 = (xlo + xi) * C, dx1 = pair ? C : 0;
            ga_f2 ta0l = {0.f, 0.f}, ta0h = {0.f, 0.f}, ta1l = {0.f, 0.f}, ta1h = {0.f, 0.f};
            ga_f2 tb0l = {0.f, 0.f}, tb0h = {0.f, 0.f}, tb1l = {0.f, 0.f}, tb1h = {0.f, 0.f};
            int rr = 0;
            for (; rr + 4 <= nrows; rr += 4) {
                float4 v0[4], v1[4];
                float wa[4], wb[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float* src = fc + (size_t)rows[rr + j] * C + x0;
                    v0[j] = *reinterpret_cast<const float4*>(src);
                    v1[j] = *reinterpret_cast<const float4*>(src + dx1);
                    wa[j] = rwa[rr + j];
                    wb[j] = rwb[rr + j];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    ta0l = ga_fma(wa[j], ga_lo(v0[j]), ta0l); ta0h = ga_fma(wa[j], ga_hi(v0[j]), ta0h);
                    ta1l = ga_fma(wa[j], ga_lo(v1[j]), ta1l); ta1h = ga_fma(wa[j], ga_hi(v1[j]), ta1h);
                    tb0l = ga_fma(wb[j], ga_lo(v0[j]), tb0l); tb0h = ga_fma(wb[j], ga_hi(v0[j]), tb0h);
                    tb1l = ga_fma(wb[j], ga_lo(v1[j]), tb1l); tb1h = ga_fma(wb[j], ga_hi(v1[j]), tb1h);
                }
            }
            for (; rr < nrows; ++rr) {
                const float* src = fc + (size_t)rows[rr] * C + x0;
                const float4 v0 = *reinterpret_cast<const float4*>(src);
                const float4 v1 = *reinterpret_cast<const float4*>(src + dx1);
                const float wa = rwa[rr], wb = rwb[rr];
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
#pragma unroll
        for (int pw = 0; pw < 7; ++pw) {
            s_patch[(ph0 * P + pw) * 64 + lane] = make_float4(al[pw].x, al[pw].y, ah[pw].x, ah[pw].y);
            if (two) s_patch[((ph0 + 1) * P + pw) * 64 + lane] = make_float4(bl[pw].x, bl[pw].y, bh[pw].x, bh[pw].y);
        }
    }
    __syncthreads();
    // ---- phase B: tile type `wave` of the patch -> pair image, both halves of the channel concat
    if (c >= C) return;
    const size_t off = (size_t)r * C2 + c;
    switch (wave) {
        case 0: w7f::tile<0, 0, false>(s_patch, s_mask, V, MC, off, scale, lane); w7f::tile<0, 0, true>(s_patch, s_mask, V, MC, off + C, scale, lane); break;
        case 1: w7f::tile<0, 1, false>(s_patch, s_mask, V, MC, off, scale, lane); w7f::tile<0, 1, true>(s_patch, s_mask, V, MC, off + C, scale, lane); break;
        case 2: w7f::tile<1, 0, false>(s_patch, s_mask, V, MC, off, scale, lane); w7f::tile<1, 0, true>(s_patch, s_mask, V, MC, off + C, scale, lane); break;
        default: w7f::tile<1, 1, false>(s_patch, s_mask, V, MC, off, scale, lane); w7f::tile<1, 1, true>(s_patch, s_mask, V, MC, off + C, scale, lane); break;
    }
}

// ---------------------------------------------------------------------------------------------
// Backward, region form (default).
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
#ifndef CIM_ROI_RG_WIN
#define CIM_ROI_RG_WIN 32
#endif
#ifndef CIM_ROI_RG_MAXE
#define CIM_ROI_RG_MAXE 256
#endif
constexpr int RG_WIN = CIM_ROI_RG_WIN;                                  // entries per staging window (32; 16 = 64 lanes x one float4 per entry, 76 KB of LDS: two
                                                                        // workgroups per CU instead of one - measured equal, 0.19 ms at cfg2, round 4)
constexpr int RG_MAXE = CIM_ROI_RG_MAXE;                                // entries per super-window (records in LDS)
static_assert(RG_WIN == 32 || RG_WIN == 16, "staging window: 32 or 16 entries");
// ROIs per workgroup: 64, or 128 when 64 would make more than ~3.5 workgroups per CU (measured, ms at 64 / 128:
// 1000 ROIs on 33 x 43: 0.188 / 0.284, 800 on 27 x 36: 0.159 / 0.253, 1200 on 41 x 54: 0.288 / 0.261, 2000 on 33 x 43: 0.353 / 0.321)
static inline int rg_group_size(int K, int B, int C, int H, int W) {
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
    const int le = RG_WIN == 32 ? tid >> 5 : tid >> 6, lq = RG_WIN == 32 ? (tid & 31) * 2 : (tid & 63);
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
                if (RG_WIN == 32) R.r1 = *reinterpret_cast<const float4*>(grad_out + off + c1);
                if (MASKCAT) {
                    R.h0 = *reinterpret_cast<const float4*>(grad_out + off + C + c0);
                    if (RG_WIN == 32) R.h1 = *reinterpret_cast<const float4*>(grad_out + off + C + c1);
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
            if (RG_WIN == 32) *reinterpret_cast<float4*>(sb + 4) = c;
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

template <bool MASKCAT>
int launch_fwd(const float* feat, const float* rois, const float* masks, float* out, int B, int C, int H, int W, int K,
               int P, float scale, int sr, int aligned, hipStream_t st, float* ws = nullptr) {
    if (K == 0) return 0;
    dim3 grid(K, P), block(256);
    // table-driven kernels: need the table workspace, 16-byte channel rows and bin rows of at most 64 map rows (maps of <= 64 rows,
    // or <= 128 rows with P >= 4: H / P + 2 rows per bin).  Row-sum kernel (two bin rows per workgroup) for maps up to 128 columns
    // and P <= 7; the flat entry-list kernel otherwise.
    if (ws != nullptr && C % 4 == 0 && P <= FW_MAXP && (long long)H * W * C < (1ll << 30) && (H <= 64 || (P >= 4 && H <= RS_MAXD))) {
        hipLaunchKernelGGL(roi_tables_kernel, dim3(K), dim3(256), sizeof(float) * roi_rec_words(P, H, W), st, rois, ws, K, P, H, W, scale, sr, aligned);
        if (P <= 7 && W <= RS_MAXD) {
            const int nth = C >= 1024 ? 256 : ((C / 4 + 63) / 64) * 64;       // narrow maps: no idle waves
            hipLaunchKernelGGL((roi_align_fwd_rowsum2_kernel<MASKCAT>), dim3(K, (P + 1) / 2), dim3(nth), 0, st, feat, masks, out, C, H, W, P, ws);
            return 0;
        }
        const int fz = (C >= 512 * CIM_ROI_FZ) ? CIM_ROI_FZ : 1;
        hipLaunchKernelGGL((roi_align_fwd_agg_kernel<MASKCAT>), dim3(K, P, fz), dim3(fz > 1 ? 128 : 256), 0, st, feat, masks, out,
                           C, H, W, P, ws, rois, scale, sr, aligned);
        return 0;
    }
    // sample-order kernel (the oracle's operation order: bit-identical; any C, P, map size)
    if (C % 4 == 0)
        hipLaunchKernelGGL((roi_align_fwd_kernel<4, MASKCAT>), grid, block, 0, st, feat, rois, masks, out, C, H, W, P,
                           scale, sr, aligned);
    else
        hipLaunchKernelGGL((roi_align_fwd_kernel<1, MASKCAT>), grid, block, 0, st, feat, rois, masks, out, C, H, W, P,
                           scale, sr, aligned);
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

template <bool MASKCAT>
int launch_bwd(const float* go, const float* rois, const float* masks, float* gin, int B, int C, int H, int W, int K,
               int P, float scale, int sr, int aligned, float* ws, hipStream_t st, int tables_ready = 0, float* scratch = nullptr) {
    // region form: 16-byte channel rows, the table workspace, 8-bit bin indices in the packed ranges, 32-bit element offsets.
    // (With several ROI groups and no partial-map scratch the groups meet through atomicAdd: slow, but every entry point works.)
    if (K > 0 && ws != nullptr && C % 4 == 0 && P <= 16 && H < 256 && W < 256 &&
        (long long)K * P * P * (MASKCAT ? 2 : 1) * C < (1ll << 31) &&
        rg_lds_bytes(rg_group_size(K, B, C, H, W), P) <= 158 * 1024 && ((H + RG_RH - 1) / RG_RH) * ((W + RG_RW - 1) / RG_RW) <= RG_MAXREG &&
        (long long)B * ((K + 63) / 64) * ((C + 255) / 256) * RG_MAXREG < (1ll << 31))
        return launch_bwd_region<MASKCAT>(go, rois, masks, gin, B, C, H, W, K, P, scale, sr, aligned, ws, st, tables_ready, scratch);
    // generic form (odd channel counts, no workspace, maps beyond the region form's limits): one thread per output element,
    // global atomics - the reference's own formulation
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

extern "C" int cim_roi_align_wino7_pair_fwd(const float* feat, const float* rois, const float* masks, void* V, const float* scale,
                                           int B, int C, int H, int W, int K, int Rs, int P, float spatial_scale,
                                           int sampling_ratio, int aligned, float* workspace, void* stream) {
    CIM_CHECK_ARG(feat && rois && masks && V && scale && workspace);
    CIM_CHECK_ARG(B > 0 && C > 0 && C % 8 == 0 && H > 0 && W > 0 && K > 0 && Rs >= K && P == 7);
    CIM_CHECK_ARG(H <= RS_MAXD && W <= RS_MAXD && (long long)H * W * C < (1ll << 30));
    hipStream_t st = cim::as_stream(stream);
    hipLaunchKernelGGL(roi_tables_kernel, dim3(K), dim3(256), sizeof(float) * roi_rec_words(P, H, W), st, rois, workspace, K, P, H, W,
                       spatial_scale, sampling_ratio, aligned);
    hipLaunchKernelGGL(roi_align_wino7_pair_kernel, dim3((unsigned)Rs, (unsigned)((C + 255) / 256)), dim3(256), 0, st, feat, masks,
                       (float*)V, C, H, W, K, Rs, workspace, scale);
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
