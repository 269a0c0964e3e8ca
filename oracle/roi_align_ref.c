/*
 * ORACLE (test infrastructure, not product code).
 *
 * Plain-C restatement of the avg-pool ROIAlign the reference calls through
 * lib/ops/__init__.py:6 (mmcv.ops.RoIAlign; call site
 * lib/modeling/model_builder.py:229-231; spec: SURVEY.md App. D).
 *
 * PARITY UNPINNED for aligned=1: mmcv-full 1.x is a third-party dependency that is
 * not vendored under /root/reference and is not pinned by any reference test.
 * The restatement follows mmcv's published roi_align kernel
 * (mmcv/ops/csrc/common/cuda/roi_align_cuda_kernel.cuh, v1.x): `aligned` offset of
 * 0.5, no max(.,1) clamp when aligned, count = max(gh*gw, 1).  The bilinear / bin-grid
 * arithmetic is the same as the (dead) in-tree Caffe2 variant,
 * lib/modeling/roi_xfrom/roi_align/src/roi_align_kernel.cu:16-63 (bilinear),
 * :65-121 (forward), :150-193 (bilinear gradient), :195-270 (backward), which is what
 * the aligned=0 mode reproduces (max(.,1) clamp, count = gh*gw).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Layout: feat NCHW fp32 [B,C,H,W]; rois [K,5] = (batch, x1, y1, x2, y2) in input-image
 * pixels; out [K,C,P,P].  One thread, sample-by-sample accumulation in the same order
 * as the reference kernel (iy outer, ix inner).
 */
#include <math.h>
#include <stddef.h>
#include <string.h>

static float bilinear(const float *d, int H, int W, float y, float x) {
    /* roi_align_kernel.cu:16-63 */
    if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) return 0.0f;
    if (y <= 0) y = 0;
    if (x <= 0) x = 0;
    int yl = (int)y, xl = (int)x, yh, xh;
    if (yl >= H - 1) { yh = yl = H - 1; y = (float)yl; } else yh = yl + 1;
    if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else xh = xl + 1;
    float ly = y - yl, lx = x - xl, hy = 1.0f - ly, hx = 1.0f - lx;
    float v1 = d[yl * W + xl], v2 = d[yl * W + xh], v3 = d[yh * W + xl], v4 = d[yh * W + xh];
    float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
    return w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
}

static void roi_geometry(const float *roi, float scale, int P, int sampling_ratio, int aligned,
                         float *x1, float *y1, float *bw, float *bh, int *gw, int *gh, float *count) {
    float off = aligned ? 0.5f : 0.0f;
    *x1 = roi[1] * scale - off;
    *y1 = roi[2] * scale - off;
    float x2 = roi[3] * scale - off, y2 = roi[4] * scale - off;
    float rw = x2 - *x1, rh = y2 - *y1;
    if (!aligned) { rw = fmaxf(rw, 1.0f); rh = fmaxf(rh, 1.0f); }
    *bh = rh / (float)P;
    *bw = rw / (float)P;
    *gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)P);
    *gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)P);
    if (aligned) {
        int c = *gh * *gw;
        *count = (float)(c > 1 ? c : 1);
    } else {
        *count = (float)(*gh * *gw);
    }
}

int oracle_roi_align_fwd(const float *feat, const float *rois, float *out,
                         int B, int C, int H, int W, int K, int P,
                         float scale, int sampling_ratio, int aligned) {
    (void)B;
    for (int n = 0; n < K; ++n) {
        const float *roi = rois + 5 * n;
        int b = (int)roi[0];
        float x1, y1, bw, bh, count; int gw, gh;
        roi_geometry(roi, scale, P, sampling_ratio, aligned, &x1, &y1, &bw, &bh, &gw, &gh, &count);
        for (int c = 0; c < C; ++c) {
            const float *d = feat + ((size_t)b * C + c) * H * W;
            for (int ph = 0; ph < P; ++ph)
                for (int pw = 0; pw < P; ++pw) {
                    float acc = 0.0f;
                    for (int iy = 0; iy < gh; ++iy) {
                        float y = y1 + ph * bh + (iy + 0.5f) * bh / (float)gh;
                        for (int ix = 0; ix < gw; ++ix) {
                            float x = x1 + pw * bw + (ix + 0.5f) * bw / (float)gw;
                            acc += bilinear(d, H, W, y, x);
                        }
                    }
                    out[(((size_t)n * C + c) * P + ph) * P + pw] = acc / count;
                }
        }
    }
    return 0;
}

/* grad_in must be zero-initialised by the caller (accumulates, like the atomics in
 * roi_align_kernel.cu:255-262). Accumulation is in double so the oracle is a stable
 * target for a GPU kernel whose fp32 atomic order is not deterministic. */
int oracle_roi_align_bwd(const float *grad_out, const float *rois, double *grad_in,
                         int B, int C, int H, int W, int K, int P,
                         float scale, int sampling_ratio, int aligned) {
    (void)B;
    for (int n = 0; n < K; ++n) {
        const float *roi = rois + 5 * n;
        int b = (int)roi[0];
        float x1, y1, bw, bh, count; int gw, gh;
        roi_geometry(roi, scale, P, sampling_ratio, aligned, &x1, &y1, &bw, &bh, &gw, &gh, &count);
        for (int c = 0; c < C; ++c) {
            double *g = grad_in + ((size_t)b * C + c) * H * W;
            for (int ph = 0; ph < P; ++ph)
                for (int pw = 0; pw < P; ++pw) {
                    float go = grad_out[(((size_t)n * C + c) * P + ph) * P + pw];
                    for (int iy = 0; iy < gh; ++iy) {
                        float y = y1 + ph * bh + (iy + 0.5f) * bh / (float)gh;
                        for (int ix = 0; ix < gw; ++ix) {
                            float x = x1 + pw * bw + (ix + 0.5f) * bw / (float)gw;
                            /* roi_align_kernel.cu:150-193 */
                            if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) continue;
                            float yy = y <= 0 ? 0 : y, xx = x <= 0 ? 0 : x;
                            int yl = (int)yy, xl = (int)xx, yh, xh;
                            if (yl >= H - 1) { yh = yl = H - 1; yy = (float)yl; } else yh = yl + 1;
                            if (xl >= W - 1) { xh = xl = W - 1; xx = (float)xl; } else xh = xl + 1;
                            float ly = yy - yl, lx = xx - xl, hy = 1.0f - ly, hx = 1.0f - lx;
                            g[yl * W + xl] += (double)(go * (hy * hx) / count);
                            g[yl * W + xh] += (double)(go * (hy * lx) / count);
                            g[yh * W + xl] += (double)(go * (ly * hx) / count);
                            g[yh * W + xh] += (double)(go * (ly * lx) / count);
                        }
                    }
                }
        }
    }
    return 0;
}
