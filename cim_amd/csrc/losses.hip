// The four CIM training losses and their gradients w.r.t. the head scores, fused (gfx950).
//
// Replaces cls_iou_loss + loss_weight_bag_loss (x REFINE_TIMES), mil_bag_loss and PCL_loss of
// /root/reference/lib/modeling/heads.py:10-166 (formulas: SURVEY.md App. E) - ~400 tiny ATen
// launches per step in the reference formulation - by ONE launch of REFINE_TIMES + 2 workgroups
// ("jobs"): job i < R = refinement layer i, job R = mil_bag_loss, job R+1 = PCL_loss.  The inputs
// are a few tens of thousand floats, so each job is one 1024-lane workgroup: row passes with lanes
// over proposals, column passes with one wave per class column (first-index arg-max by shuffles).
// Alongside each loss the job writes its gradient "components" (d loss_k / d score tensor) for a
// unit upstream gradient; autograd combines them with the actual upstream scalars.
//   fp32, log via logf; every score is clamped to [1e-6, 1-1e-6] before a log exactly like the
//   reference, with the clamp's pass-through gradient (inclusive bounds).
#include "common.h"
#include "../../include/cim_hip.h"
#include <limits.h>

namespace {

constexpr int NT = 1024;
constexpr float LO = 1e-6f, HI = 1.0f - 1e-6f;
#ifndef CIM_LOSS_CLOCKS
#define CIM_LOSS_CLOCKS 0            // 1: phase stamps (100 MHz wall clock) of the loss launch's jobs -> cim_debug_loss_clocks (tools/loss_clocks.py)
#endif
#if CIM_LOSS_CLOCKS
__device__ unsigned long long g_loss_clk[8][16];                         // [job][stamp]
#define LCLK(JOB, I) do { if (threadIdx.x == 0 && (JOB) < 8) g_loss_clk[JOB][I] = wall_clock64(); } while (0)
#else
#define LCLK(JOB, I) do { } while (0)
#endif

__device__ __forceinline__ float clampf(float x) { return fminf(fmaxf(x, LO), HI); }
__device__ __forceinline__ float inrange(float x) { return (x >= LO && x <= HI) ? 1.0f : 0.0f; }

__device__ float block_sum(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float s = 0.0f;
    for (int w = 0; w < NT / 64; ++w) s += red[w];
    return s;
}

// four block sums behind ONE pair of barriers (buf: 64 floats of LDS nobody else uses until the next barrier)
__device__ void block_sum4(float& v0, float& v1, float& v2, float& v3, float* buf) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        v0 += __shfl_xor(v0, o); v1 += __shfl_xor(v1, o); v2 += __shfl_xor(v2, o); v3 += __shfl_xor(v3, o);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) { buf[wave * 4 + 0] = v0; buf[wave * 4 + 1] = v1; buf[wave * 4 + 2] = v2; buf[wave * 4 + 3] = v3; }
    __syncthreads();
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    for (int w = 0; w < NT / 64; ++w) { s0 += buf[w * 4 + 0]; s1 += buf[w * 4 + 1]; s2 += buf[w * 4 + 2]; s3 += buf[w * 4 + 3]; }
    v0 = s0; v1 = s1; v2 = s2; v3 = s3;
}

// (value, index) max over the 64 lanes; ties -> lower index (first maximum)
__device__ __forceinline__ void wave_argmax(float& v, int& i) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o);
        const int oi = __shfl_xor(i, o);
        if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
    }
}

// lanes per row of the column passes: the smallest power of two >= C1 (1024: one wave per column instead)
__device__ __forceinline__ int loss_group(int C1) {
    return C1 <= 32 ? 32 : C1 <= 64 ? 64 : C1 <= 128 ? 128 : C1 <= 256 ? 256 : 1024;
}

// loss_weight_bag_loss (heads.py:43-74) of ONE class column from its two arg-maxima: the column's term of the bag loss (returned) and
// its gradient components (added in place: one thread per column, no two columns share an element)
__device__ __forceinline__ float refine_column_finish(const cim_loss_args& a, int li, int c, float fv, int fi, float uv, int ui,
                                                      const int* hot, const float* rc, const float* ri, const float* w, float ws,
                                                      float* g_rc_bag, float* g_ri_bag, int LD) {
    const int C1 = a.C1;
    const float L = (c == 0) ? 1.0f : a.labels[c - 1];
    const float raw = fv * L + uv * (1.0f - L);
    const float agg = clampf(raw);
    const bool seen = (L == 1.0f);
    const int idx = seen ? fi : ui;
    const float om = seen ? ws * w[idx] : 1.0f;
    const float term = -(L * logf(agg) + (1.0f - L) * logf(1.0f - agg)) * om / (float)C1;
    const float dagg = -(L / agg - (1.0f - L) / (1.0f - agg)) * om / (float)C1 * inrange(raw);
    const float du_f = dagg * L * ((hot[fi] == c) ? 1.0f : 0.0f);
    const float du_u = dagg * (1.0f - L);
    if (du_f != 0.0f) {
        const float xr = rc[(size_t)fi * LD + c], xi = ri[(size_t)fi * LD + c];
        g_rc_bag[(size_t)fi * C1 + c] += du_f * clampf(xi) * inrange(xr);
        g_ri_bag[(size_t)fi * C1 + c] += du_f * clampf(xr) * inrange(xi);
    }
    if (du_u != 0.0f) {
        const float xr = rc[(size_t)ui * LD + c], xi = ri[(size_t)ui * LD + c];
        g_rc_bag[(size_t)ui * C1 + c] += du_u * clampf(xi) * inrange(xr);
        g_ri_bag[(size_t)ui * C1 + c] += du_u * clampf(xr) * inrange(xi);
    }
    return term;
}

__device__ void refine_job(const cim_loss_args& a, int li, float* red, int* hot, float* grp_scratch) {
    const int N = a.N, C1 = a.C1, LD = a.ld > 0 ? a.ld : a.C1, tid = threadIdx.x;
    (void)LD;
    const float* rc = a.rc[li];
    const float* ri = a.ri[li];
    const float* Y = a.pseudo_labels[li];
    const uint16_t* t16 = a.pseudo_iou_f16[li];
    const float* w = a.loss_weights[li];
    const float ws = a.weight_scale[li];           // lmda of model_builder.py:172,194
    float* g_rc_cls = a.grad + (size_t)(3 + 4 * li + 0) * N * C1;
    float* g_rc_bag = a.grad + (size_t)(3 + 4 * li + 1) * N * C1;
    float* g_ri_iou = a.grad + (size_t)(3 + 4 * li + 2) * N * C1;
    float* g_ri_bag = a.grad + (size_t)(3 + 4 * li + 3) * N * C1;
    for (int i = tid; i < N * C1; i += NT) { g_rc_cls[i] = 0.f; g_rc_bag[i] = 0.f; g_ri_iou[i] = 0.f; g_ri_bag[i] = 0.f; }
    float* out = a.part + 4 * li;                  // [bag, pcl, cls, iou]
    LCLK(li, 1);
    if (!a.layer_valid[li]) {                      // CIM_layer returned None: the layer contributes nothing
        if (tid < 4) out[tid] = 0.0f;
        return;
    }
    // pass 1 over rows: labelled class of each row, cls / iou numerators and counts
    float s_cls = 0.f, s_iou = 0.f, n_lab = 0.f, n_fg = 0.f;
    for (int n = tid; n < N; n += NT) {
        // (the first non-zero column, WITHOUT an early exit: a `break` makes every load wait for the branch on the one before it -
        // up to C1 dependent L2 round trips per row; walking down from the last column keeps the loads independent)
        int h = -1;
        for (int c = C1 - 1; c >= 0; --c)
            if (Y[(size_t)n * C1 + c] != 0.0f) h = c;
        hot[n] = h;
        if (h < 0) continue;
        const float wn = ws * w[n];
        s_cls += -logf(clampf(rc[(size_t)n * LD + h])) * wn;
        n_lab += 1.0f;
        if (h >= 1) {
            const float d = clampf(ri[(size_t)n * LD + h]) - cim::h2f(t16[n]);
            const float ad = fabsf(d);
            s_iou += (ad < 1.0f ? 0.5f * d * d : ad - 0.5f) * wn;
            n_fg += 1.0f;
        }
    }
    LCLK(li, 2);
    block_sum4(s_cls, s_iou, n_lab, n_fg, grp_scratch);          // (one pair of barriers instead of four: same sums, same order)
    const float cls_loss = n_lab > 0.f ? s_cls / n_lab : 0.f;                  // heads.py:104-114
    const float iou_loss = n_fg > 0.f ? s_iou / n_fg : 0.f;                    // heads.py:116-136
    LCLK(li, 3);
    // pass 2 over rows: gradient components of cls_loss / iou_loss
    for (int n = tid; n < N; n += NT) {
        const int h = hot[n];
        if (h < 0) continue;
        const float wn = ws * w[n];
        const float x = rc[(size_t)n * LD + h];
        g_rc_cls[(size_t)n * C1 + h] = -wn / (clampf(x) * n_lab) * inrange(x);
        if (h >= 1) {
            const float xi = ri[(size_t)n * LD + h];
            const float d = clampf(xi) - cim::h2f(t16[n]);
            const float dd = fabsf(d) < 1.0f ? d : (d > 0.f ? 1.0f : -1.0f);
            g_ri_iou[(size_t)n * C1 + h] = dd * wn / n_fg * inrange(xi);
        }
    }
    __syncthreads();
    // column pass (loss_weight_bag_loss, heads.py:43-74): one wave per class column
    LCLK(li, 4);
    // column pass (round 6: lanes along the COLUMNS of a row).  The first form gave a wave one column and its lanes the rows: a lane's
    // loads were LD floats apart (64 cache lines per instruction) - 25 us of the job's 47 (tools/loss_clocks.py).  Now a group of G
    // lanes (G = power of two >= C1) reads one row's C1 consecutive scores, NT / G rows per pass; every lane keeps the running
    // (maximum, first index) of ITS column over its rows, the groups meet in LDS in group order with ties to the lower row - the same
    // arg-max as before (first maximum), whichever way the rows are dealt.
    const int G = loss_group(C1), NG = NT / G, grp = tid / G, c = tid % G;
    float bag = 0.0f;
    if (G <= 256) {
        float fv = -INFINITY, uv = -INFINITY;
        int fi = INT_MAX, ui = INT_MAX;
        if (c < C1)
            for (int n0 = grp; n0 < N; n0 += 8 * NG) {      // eight rows' loads in flight (one round trip per iteration otherwise)
                float xr[8], xi[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int n = min(n0 + j * NG, N - 1);
                    xr[j] = rc[(size_t)n * LD + c];
                    xi[j] = ri[(size_t)n * LD + c];
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int n = n0 + j * NG;
                    if (n < N) {
                        const float u = clampf(xr[j]) * clampf(xi[j]);
                        const float f = (hot[n] == c) ? u : 0.0f;                // ind * predict * tmp_pseudo_label
                        if (f > fv) { fv = f; fi = n; }
                        if (u > uv) { uv = u; ui = n; }
                    }
                }
            }
        float* gv = grp_scratch;                                                 // [4][NG][C1]: fv, fi, uv, ui
        int* gi = reinterpret_cast<int*>(grp_scratch);
        if (c < C1) {
            gv[(0 * NG + grp) * C1 + c] = fv; gi[(1 * NG + grp) * C1 + c] = fi;
            gv[(2 * NG + grp) * C1 + c] = uv; gi[(3 * NG + grp) * C1 + c] = ui;
        }
        __syncthreads();
        if (tid < C1) {
            const int cc = tid;
            fv = gv[(0 * NG) * C1 + cc]; fi = gi[(1 * NG) * C1 + cc]; uv = gv[(2 * NG) * C1 + cc]; ui = gi[(3 * NG) * C1 + cc];
            for (int k = 1; k < NG; ++k) {
                const float ofv = gv[(0 * NG + k) * C1 + cc], ouv = gv[(2 * NG + k) * C1 + cc];
                const int ofi = gi[(1 * NG + k) * C1 + cc], oui = gi[(3 * NG + k) * C1 + cc];
                if (ofv > fv || (ofv == fv && ofi < fi)) { fv = ofv; fi = ofi; }
                if (ouv > uv || (ouv == uv && oui < ui)) { uv = ouv; ui = oui; }
            }
            bag = refine_column_finish(a, li, cc, fv, fi, uv, ui, hot, rc, ri, w, ws, g_rc_bag, g_ri_bag, LD);
        }
    } else {                                      // (more than 256 classes: one wave per column, lanes over the rows)
        const int wave = tid >> 6, lane = tid & 63;
        for (int cc = wave; cc < C1; cc += NT / 64) {
            float fv = -INFINITY, uv = -INFINITY;
            int fi = INT_MAX, ui = INT_MAX;
            for (int n = lane; n < N; n += 64) {
                const float u = clampf(rc[(size_t)n * LD + cc]) * clampf(ri[(size_t)n * LD + cc]);
                const float f = (hot[n] == cc) ? u : 0.0f;
                if (f > fv) { fv = f; fi = n; }
                if (u > uv) { uv = u; ui = n; }
            }
            wave_argmax(fv, fi);
            wave_argmax(uv, ui);
            if (lane == 0) bag += refine_column_finish(a, li, cc, fv, fi, uv, ui, hot, rc, ri, w, ws, g_rc_bag, g_ri_bag, LD);
        }
    }
    LCLK(li, 5);
    bag = block_sum(bag, red);
    if (tid == 0) { out[0] = bag; out[1] = 0.f; out[2] = cls_loss; out[3] = iou_loss; }
}

__device__ void mil_job(const cim_loss_args& a, float* red, float* grp_scratch) {
    const int N = a.N, C1 = a.C1, LD = a.ld > 0 ? a.ld : a.C1, tid = threadIdx.x;
    float* g_pc = a.grad;                                   // component 0: d bag / d predict_cls
    float* g_pd = a.grad + (size_t)2 * N * C1;              // component 2: d bag / d predict_det
    float bag = 0.0f;
    const int G = loss_group(C1), NG = NT / G, grp = tid / G, c = tid % G;
    if (G <= 256) {
        // heads.py:149-166 with lanes along the columns of a row (see refine_job's column pass: this job took 61 us with a wave per
        // column - the longest of the launch): column dots as NG partial sums per column, added in group order; then the gradient
        // components, rows coalesced
        float s = 0.0f;
        if (c < C1)
            for (int n0 = grp; n0 < N; n0 += 8 * NG) {
                float x[8], y[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int n = min(n0 + j * NG, N - 1);
                    x[j] = a.pc[(size_t)n * LD + c];
                    y[j] = a.pd[(size_t)n * LD + c];
                }
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (n0 + j * NG < N) s += x[j] * y[j];
            }
        float* part = grp_scratch;                          // [NG][C1]
        float* dsv = grp_scratch + NG * C1;                 // [C1]
        if (c < C1) part[grp * C1 + c] = s;
        __syncthreads();
        if (tid < C1) {
            float t = part[tid];
            for (int k = 1; k < NG; ++k) t += part[k * C1 + tid];
            const float L = (tid == 0) ? 1.0f : a.labels[tid - 1];
            const float p = clampf(t);
            bag = -(L * logf(p) + (1.0f - L) * logf(1.0f - p)) / (float)C1;
            dsv[tid] = -(L / p - (1.0f - L) / (1.0f - p)) / (float)C1 * inrange(t);
        }
        __syncthreads();
        if (c < C1) {
            const float ds = dsv[c];
            for (int n0 = grp; n0 < N; n0 += 8 * NG) {
                float x[8], y[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int n = min(n0 + j * NG, N - 1);
                    x[j] = a.pc[(size_t)n * LD + c];
                    y[j] = a.pd[(size_t)n * LD + c];
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int n = n0 + j * NG;
                    if (n < N) {
                        g_pc[(size_t)n * C1 + c] = ds * y[j];
                        g_pd[(size_t)n * C1 + c] = ds * x[j];
                    }
                }
            }
        }
    } else {
        const int wave = tid >> 6, lane = tid & 63;
        for (int cc = wave; cc < C1; cc += NT / 64) {
            float s = 0.0f;
            for (int n = lane; n < N; n += 64) s += a.pc[(size_t)n * LD + cc] * a.pd[(size_t)n * LD + cc];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            const float L = (cc == 0) ? 1.0f : a.labels[cc - 1];
            const float p = clampf(s);
            if (lane == 0) bag += -(L * logf(p) + (1.0f - L) * logf(1.0f - p)) / (float)C1;
            const float ds = -(L / p - (1.0f - L) / (1.0f - p)) / (float)C1 * inrange(s);
            for (int n = lane; n < N; n += 64) {
                g_pc[(size_t)n * C1 + cc] = ds * a.pd[(size_t)n * LD + cc];
                g_pd[(size_t)n * C1 + cc] = ds * a.pc[(size_t)n * LD + cc];
            }
        }
    }
    bag = block_sum(bag, red);
    float* out = a.part + 4 * a.R;
    if (tid == 0) { out[0] = bag; out[1] = 0.f; out[2] = 0.f; out[3] = 0.f; }
}

__device__ float block_min(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float s = INFINITY;
    for (int w = 0; w < NT / 64; ++w) s = fminf(s, red[w]);
    return s;
}

// The cluster structure of `mat` (heads.py:14-21), in LDS: row_val [N] = the row's non-zero entry (0: none),
// row_col [N] = its column, row_cluster [N] = rank of that id among the distinct ids (ascending = torch.unique order),
// csize [K].  Returns the number of clusters; *bg = rank of the id found in column 0 (-1: none).
__device__ int pcl_plan(const cim_loss_args& a, float* red, float* row_val, int16_t* row_col, int16_t* row_cluster,
                        int* csize, int* bg) {
    const int N = a.N, C1 = a.C1, LD = a.ld > 0 ? a.ld : a.C1, tid = threadIdx.x;
    (void)LD;
    float bgv_lo = INFINITY, bgv_hi = -INFINITY;
    int err = 0;
    for (int n = tid; n < N; n += NT) {
        int cnt = 0, col = -1;
        float val = 0.0f;
        for (int c = 0; c < C1; ++c) {
            const float v = a.mat[(size_t)n * C1 + c];
            if (v != 0.0f) {
                if (cnt == 0) { col = c; val = v; }
                ++cnt;
            }
        }
        if (cnt > 1) err |= 4;
        row_val[n] = val;
        row_col[n] = (int16_t)col;
        row_cluster[n] = -1;
        if (col == 0) { bgv_lo = fminf(bgv_lo, val); bgv_hi = fmaxf(bgv_hi, val); }
    }
    const float bg_lo = block_min(bgv_lo, red);
    const float bg_hi = -block_min(-bgv_hi, red);
    if (bg_lo != INFINITY && bg_lo != bg_hi) err |= 8;      // heads.py:20: assert len(unique ids in column 0) <= 1
    int K = 0;
    *bg = -1;
    float last = -INFINITY;
    for (;;) {
        float m = INFINITY;
        for (int n = tid; n < N; n += NT)
            if (row_col[n] >= 0 && row_val[n] > last) m = fminf(m, row_val[n]);
        m = block_min(m, red);
        if (m == INFINITY) break;
        if (K >= CIM_PCL_MAX_CLUSTERS) { err |= 16; break; }
        float cnt = 0.0f;
        for (int n = tid; n < N; n += NT)
            if (row_col[n] >= 0 && row_val[n] == m) { row_cluster[n] = (int16_t)K; cnt += 1.0f; }
        cnt = block_sum(cnt, red);
        if (tid == 0) csize[K] = (int)cnt;
        if (bg_lo != INFINITY && m == bg_lo) *bg = K;
        last = m;
        ++K;
    }
    if (err && a.status) atomicOr(a.status, err);
    __syncthreads();
    return K;
}

__device__ void pcl_job(const cim_loss_args& a, float* red, unsigned char* scratch, float* grp_scratch) {
    const int N = a.N, C1 = a.C1, LD = a.ld > 0 ? a.ld : a.C1, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    float* g = a.grad + (size_t)1 * N * C1;                 // component 1: d pcl / d predict_cls
    for (int i = tid; i < N * C1; i += NT) g[i] = 0.f;
    float* row_val = reinterpret_cast<float*>(scratch);                       // [N]
    int16_t* row_col = reinterpret_cast<int16_t*>(row_val + N);               // [N]
    int16_t* row_cluster = row_col + N;                                       // [N]
    __shared__ int csize[CIM_PCL_MAX_CLUSTERS];
    int bg_cluster;
    LCLK(a.R + 1, 1);
    const int K = pcl_plan(a, red, row_val, row_col, row_cluster, csize, &bg_cluster);
    LCLK(a.R + 1, 2);
    float den = 1e-6f;                                      // heads.py:22
    for (int k = 0; k < K; ++k) den += (float)csize[k];
    const float scale = 12.0f / den;                        // heads.py:40-41
    float acc = 0.0f;
    const int G = loss_group(C1), NG = NT / G, grp = tid / G, cg = tid % G;
    if (G <= 256) {
        // lanes along the columns of a row (see refine_job's column pass), one pass over the rows per cluster
        float* part = grp_scratch;                                   // [NG][C1]
        int* anyf = reinterpret_cast<int*>(grp_scratch) + NG * C1;   // [NG][C1]
        float* dvv = grp_scratch + 2 * NG * C1;                      // [C1]
        for (int k = 0; k < K; ++k) {
            const float nk = (float)csize[k];
            if (k == bg_cluster) {                          // heads.py:33-38: every member row vs its own pattern
                if (cg < C1) {
                    float sb = 0.0f;
                    for (int n0 = grp; n0 < N; n0 += 8 * NG) {
                        float xs[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) xs[j] = a.pc[(size_t)min(n0 + j * NG, N - 1) * LD + cg];
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const int n = n0 + j * NG;
                            if (n >= N || row_cluster[n] != k) continue;
                            const float x = xs[j], p = clampf(x);
                            const float t = (row_col[n] == cg) ? 1.0f : 0.0f;
                            sb += -(t * logf(p) + (1.0f - t) * logf(1.0f - p));
                            g[(size_t)n * C1 + cg] = scale * (-(t / p - (1.0f - t) / (1.0f - p))) / (float)C1 * inrange(x);
                        }
                    }
                    acc += sb / (float)C1;                  // = n_k * mean over (rows, cols)
                }
            } else {                                        // heads.py:25-31: mean row vector vs column indicator
                float sc = 0.0f;
                int any = 0;
                if (cg < C1) {
                    for (int n0 = grp; n0 < N; n0 += 8 * NG) {
                        float xs[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) xs[j] = a.pc[(size_t)min(n0 + j * NG, N - 1) * LD + cg];
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const int n = n0 + j * NG;
                            if (n >= N || row_cluster[n] != k) continue;
                            sc += xs[j];
                            any |= (row_col[n] == cg);
                        }
                    }
                    part[grp * C1 + cg] = sc;
                    anyf[grp * C1 + cg] = any;
                }
                __syncthreads();
                if (tid < C1) {
                    float t_sum = part[tid];
                    int t_any = anyf[tid];
                    for (int q = 1; q < NG; ++q) { t_sum += part[q * C1 + tid]; t_any |= anyf[q * C1 + tid]; }
                    const float v = t_sum / nk, p = clampf(v), t = t_any ? 1.0f : 0.0f;
                    acc += nk * (-(t * logf(p) + (1.0f - t) * logf(1.0f - p))) / (float)C1;
                    dvv[tid] = scale * (-(t / p - (1.0f - t) / (1.0f - p))) / (float)C1 * inrange(v);   // n_k * (1/n_k)
                }
                __syncthreads();
                if (cg < C1) {
                    const float dv = dvv[cg];
                    for (int n = grp; n < N; n += NG)
                        if (row_cluster[n] == k) g[(size_t)n * C1 + cg] = dv;
                }
                __syncthreads();                            // (part / anyf / dvv are rewritten by the next cluster)
            }
        }
    } else {
        // (cluster, column) pairs over the 16 waves
        for (int job = wave; job < K * C1; job += NT / 64) {
            const int k = job / C1, c = job % C1;
            const float nk = (float)csize[k];
            if (k == bg_cluster) {                              // heads.py:33-38: every member row vs its own pattern
                float s = 0.0f;
                for (int n = lane; n < N; n += 64) {
                    if (row_cluster[n] != k) continue;
                    const float x = a.pc[(size_t)n * LD + c], p = clampf(x);
                    const float t = (row_col[n] == c) ? 1.0f : 0.0f;
                    s += -(t * logf(p) + (1.0f - t) * logf(1.0f - p));
                    g[(size_t)n * C1 + c] = scale * (-(t / p - (1.0f - t) / (1.0f - p))) / (float)C1 * inrange(x);
                }
    #pragma unroll
                for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
                if (lane == 0) acc += s / (float)C1;            // = n_k * mean over (rows, cols)
            } else {                                            // heads.py:25-31: mean row vector vs column indicator
                float s = 0.0f;
                int any = 0;
                for (int n = lane; n < N; n += 64) {
                    if (row_cluster[n] != k) continue;
                    s += a.pc[(size_t)n * LD + c];
                    any |= (row_col[n] == c);
                }
    #pragma unroll
                for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); any |= __shfl_xor(any, o); }
                const float v = s / nk, p = clampf(v), t = any ? 1.0f : 0.0f;
                if (lane == 0) acc += nk * (-(t * logf(p) + (1.0f - t) * logf(1.0f - p))) / (float)C1;
                const float dv = scale * (-(t / p - (1.0f - t) / (1.0f - p))) / (float)C1 * inrange(v);   // n_k * (1/n_k)
                for (int n = lane; n < N; n += 64)
                    if (row_cluster[n] == k) g[(size_t)n * C1 + c] = dv;
            }
        }
    }
    LCLK(a.R + 1, 3);
    acc = block_sum(acc, red);
    float* out = a.part + 4 * (a.R + 1);
    if (tid == 0) { out[0] = 0.f; out[1] = scale * acc; out[2] = 0.f; out[3] = 0.f; }
}

__global__ __launch_bounds__(NT) void losses_kernel(const cim_loss_args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* red = reinterpret_cast<float*>(smem);            // [16]
    int* hot = reinterpret_cast<int*>(smem + 64);           // [N] (refinement jobs) / the PCL job's 8N bytes of row tables
    float* grp_scratch = reinterpret_cast<float*>(smem + 64 + 8 * (size_t)a.N);     // [4][NT / G][C1] words: the column passes' group partials
    const int job = blockIdx.x;
    LCLK(job, 0);
    if (job < a.R) refine_job(a, job, red, hot, grp_scratch);
    else if (job == a.R) mil_job(a, red, grp_scratch);
    else pcl_job(a, red, smem + 64, grp_scratch);
    LCLK(job, 7);
}

}  // namespace

#if CIM_LOSS_CLOCKS
extern "C" int cim_debug_loss_clocks(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_loss_clk), sizeof(g_loss_clk)) == hipSuccess ? 0 : 1;
}
#endif

extern "C" int cim_losses_fwd(const cim_loss_args* args, void* stream) {
    CIM_CHECK_ARG(args != nullptr);
    const cim_loss_args& a = *args;
    CIM_CHECK_ARG(a.N > 0 && a.N <= 15000 && a.C1 > 1 && a.C1 <= 32767 && a.R >= 0 && a.R <= 3 && (a.ld == 0 || a.ld >= a.C1));
    CIM_CHECK_ARG(a.pc && a.pd && a.labels && a.part && a.grad && a.mat && (a.R == 0 || a.layer_valid));
    for (int i = 0; i < a.R; ++i)
        CIM_CHECK_ARG(a.rc[i] && a.ri[i] && a.pseudo_labels[i] && a.pseudo_iou_f16[i] && a.loss_weights[i]);
    const int lg = a.C1 <= 32 ? 32 : a.C1 <= 64 ? 64 : a.C1 <= 128 ? 128 : a.C1 <= 256 ? 256 : 1024;
    const size_t lds = 64 + 8 * (size_t)a.N + (lg <= 256 ? 16 * (size_t)(NT / lg) * a.C1 + 16 : 0) + 256;      // (+ block_sum4's 64 floats)
    if (lds > 64 * 1024)
        CIM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(losses_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(losses_kernel, dim3(a.R + 2), dim3(NT), lds, cim::as_stream(stream), a);
    CIM_CHECK_LAUNCH();
    return 0;
}

// =============================================================================================
// Head activations (a-3): the epilogue of cls_iou_model.forward, /root/reference/lib/modeling/heads.py:199-217.
// Input = the fused linear output  logits [N, (2+2R)*C1]  with column blocks
//   [ classifier | detector | refine_cls[0..R) | refine_iou[0..R) ],
// output scores in the same layout: softmax over classes (classifier, refine_cls), softmax over
// PROPOSALS (detector, dim 0), sigmoid (refine_iou).  Two launches forward (detector column
// statistics, then one wave per proposal row) and two backward.
namespace {

__global__ __launch_bounds__(256) void head_colstat_kernel(const float* __restrict__ x, int N, int C1, int ld, int off,
                                                           float* __restrict__ stat) {
    // block c: max and sum(exp(x - max)) over rows of column off + c
    __shared__ float red[4];
    const int c = blockIdx.x, tid = threadIdx.x;
    float m = -INFINITY;
    for (int n = tid; n < N; n += 256) m = fmaxf(m, x[(size_t)n * ld + off + c]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.0f;
    for (int n = tid; n < N; n += 256) s += expf(x[(size_t)n * ld + off + c] - m);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) {
        stat[c] = m;
        stat[C1 + c] = red[0] + red[1] + red[2] + red[3];
    }
}

// one wave per row; blocks: 0 = classifier, 1 = detector, 2..2+R = refine_cls, then refine_iou
__global__ __launch_bounds__(256) void head_act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int C1,
                                                           int R, const float* __restrict__ stat) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= N) return;
    const int ld = (2 + 2 * R) * C1;
    const float* xr = x + (size_t)row * ld;
    float* yr = y + (size_t)row * ld;
    for (int b = 0; b < 2 + 2 * R; ++b) {
        const int off = b * C1;
        if (b == 1) {                                   // softmax over proposals (dim 0)
            for (int c = lane; c < C1; c += 64) yr[off + c] = expf(xr[off + c] - stat[c]) / stat[C1 + c];
        } else if (b < 2 + R) {                         // softmax over classes
            float m = -INFINITY;
            for (int c = lane; c < C1; c += 64) m = fmaxf(m, xr[off + c]);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
            float s = 0.0f;
            for (int c = lane; c < C1; c += 64) s += expf(xr[off + c] - m);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            for (int c = lane; c < C1; c += 64) yr[off + c] = expf(xr[off + c] - m) / s;
        } else {                                        // sigmoid
            for (int c = lane; c < C1; c += 64) yr[off + c] = 1.0f / (1.0f + expf(-xr[off + c]));
        }
    }
}

__global__ __launch_bounds__(256) void head_coldot_kernel(const float* __restrict__ y, const float* __restrict__ dy, int N,
                                                          int C1, int ld, int off, float* __restrict__ dot) {
    __shared__ float red[4];
    const int c = blockIdx.x, tid = threadIdx.x;
    float s = 0.0f;
    for (int n = tid; n < N; n += 256) s += y[(size_t)n * ld + off + c] * dy[(size_t)n * ld + off + c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) dot[c] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void head_act_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                           float* __restrict__ dx, int N, int C1, int R,
                                                           const float* __restrict__ dot) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= N) return;
    const int ld = (2 + 2 * R) * C1;
    const float* yr = y + (size_t)row * ld;
    const float* gr = dy + (size_t)row * ld;
    float* dr = dx + (size_t)row * ld;
    for (int b = 0; b < 2 + 2 * R; ++b) {
        const int off = b * C1;
        if (b == 1) {
            for (int c = lane; c < C1; c += 64) dr[off + c] = yr[off + c] * (gr[off + c] - dot[c]);
        } else if (b < 2 + R) {
            float s = 0.0f;
            for (int c = lane; c < C1; c += 64) s += yr[off + c] * gr[off + c];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            for (int c = lane; c < C1; c += 64) dr[off + c] = yr[off + c] * (gr[off + c] - s);
        } else {
            for (int c = lane; c < C1; c += 64) dr[off + c] = gr[off + c] * yr[off + c] * (1.0f - yr[off + c]);
        }
    }
}

// d(total loss) / d(score matrix) from the stored components and the four upstream scalars, in the heads' fused layout
// [N][(2 + 2R) C1] = (predict_cls | predict_det | refine_cls[0..R) | refine_iou[0..R)) - what cim_head_act_bwd consumes:
//   pc: g_bag G0 + g_pcl G1;  pd: g_bag G2;  rc_i: g_cls G[3+4i] + g_bag G[4+4i];  ri_i: g_iou G[5+4i] + g_bag G[6+4i]
// (g = (bag, pcl, cls, iou) on the device; replaces ~20 element-wise ATen launches and the 8-way concatenation of their results)
// g: the six gradient pointers of cim_loss_finish's outputs (bag, pcl, cls, iou, 3 iou, total), any of them null
struct LossGrads { const float* p[6]; };
__global__ __launch_bounds__(256) void loss_grad_combine_kernel(const float* __restrict__ G, const LossGrads g,
                                                                float* __restrict__ out, int N, int C1, int R) {
    const size_t plane = (size_t)N * C1;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= plane) return;
    const int n = (int)(i / C1), c = (int)(i - (size_t)n * C1);
    float v[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) v[k] = g.p[k] ? g.p[k][0] : 0.0f;
    const float g_bag = v[0] + v[5], g_pcl = v[1] + v[5], g_cls = v[2] + v[5], g_iou = v[3] + 3.0f * (v[4] + v[5]);
    float* o = out + (size_t)n * (2 + 2 * R) * C1 + c;
    o[0] = g_bag * G[i] + g_pcl * G[plane + i];
    o[C1] = g_bag * G[2 * plane + i];
    for (int r = 0; r < R; ++r) {
        const float* Gr = G + (size_t)(3 + 4 * r) * plane + i;
        o[(2 + r) * C1] = g_cls * Gr[0] + g_bag * Gr[plane];
        o[(2 + R + r) * C1] = g_iou * Gr[2 * plane] + g_bag * Gr[3 * plane];
    }
}

// part [rows][4] (bag, pcl, cls, iou partial sums of the loss launch) -> out[6] = (bag, pcl, cls, iou, 3 iou, total) with
// total = ((bag + pcl) + cls) + 3 iou  (model_builder.py:199 weights the IoU loss by 3; lib/utils/training_stats.py:72-83 adds
// the four up in dictionary order for the backward pass) - one launch instead of a reduction, a scaling and the driver's
// per-loss means and adds
__global__ void loss_finish_kernel(const float* __restrict__ part, int rows, float* __restrict__ out) {
    const int k = threadIdx.x;
    float s = 0.0f;
    if (k < 4)
        for (int r = 0; r < rows; ++r) s += part[r * 4 + k];
    const float bag = __shfl(s, 0), pcl = __shfl(s, 1), cls = __shfl(s, 2), iou = __shfl(s, 3);
    if (k < 4) out[k] = s;
    if (k == 4) out[4] = 3.0f * iou;
    if (k == 5) out[5] = ((bag + pcl) + cls) + 3.0f * iou;
}

}  // namespace

extern "C" int cim_loss_finish(const float* part, int rows, float* out, void* stream) {
    CIM_CHECK_ARG(part && out && rows > 0);
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(64), 0, cim::as_stream(stream), part, rows, out);
    CIM_CHECK_LAUNCH();
    return 0;
}

extern "C" int cim_loss_grad_combine(const float* G, const float* g_bag, const float* g_pcl, const float* g_cls, const float* g_iou,
                                     const float* g_iou3, const float* g_total, float* out, int N, int C1, int R, void* stream) {
    CIM_CHECK_ARG(G && out && N > 0 && C1 > 0 && R >= 0 && R <= 3);
    const LossGrads g{{g_bag, g_pcl, g_cls, g_iou, g_iou3, g_total}};
    const size_t n = (size_t)N * C1;
    hipLaunchKernelGGL(loss_grad_combine_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, cim::as_stream(stream), G, g, out, N, C1, R);
    CIM_CHECK_LAUNCH();
    return 0;
}

extern "C" int cim_head_act_fwd(const float* logits, float* scores, float* colstat, int N, int C1, int R, void* stream) {
    CIM_CHECK_ARG(logits && scores && colstat && N > 0 && C1 > 0 && R >= 0 && R <= 8);
    const int ld = (2 + 2 * R) * C1;
    hipStream_t st = cim::as_stream(stream);
    hipLaunchKernelGGL(head_colstat_kernel, dim3(C1), dim3(256), 0, st, logits, N, C1, ld, C1, colstat);
    hipLaunchKernelGGL(head_act_fwd_kernel, dim3((N + 3) / 4), dim3(256), 0, st, logits, scores, N, C1, R, colstat);
    CIM_CHECK_LAUNCH();
    return 0;
}

extern "C" int cim_head_act_bwd(const float* scores, const float* grad_scores, float* grad_logits, float* coldot, int N,
                                int C1, int R, void* stream) {
    CIM_CHECK_ARG(scores && grad_scores && grad_logits && coldot && N > 0 && C1 > 0 && R >= 0 && R <= 8);
    const int ld = (2 + 2 * R) * C1;
    hipStream_t st = cim::as_stream(stream);
    hipLaunchKernelGGL(head_coldot_kernel, dim3(C1), dim3(256), 0, st, scores, grad_scores, N, C1, ld, C1, coldot);
    hipLaunchKernelGGL(head_act_bwd_kernel, dim3((N + 3) / 4), dim3(256), 0, st, scores, grad_scores, grad_logits, N, C1, R,
                       coldot);
    CIM_CHECK_LAUNCH();
    return 0;
}
