/*
 * cim_hip.h - C ABI of libcim_hip.so, the MI355X (gfx950) kernel library behind the
 * per-image training step of ZechengLi19/CIM.
 *
 * Conventions (every entry point):
 *   - returns 0 on success, a positive hipError_t on a HIP failure, -1 on bad arguments;
 *     cim_last_error() returns a thread-local description of the last failure;
 *   - every pointer is a caller-owned DEVICE pointer to a dense array of the stated dtype
 *     (the Python host passes torch.Tensor.data_ptr()); nothing is allocated or freed (no hipMalloc, no events, no streams:
 *     device scratch, fork / join events and side streams are the caller's and come in as arguments), no host synchronisation
 *     happens, no state is kept between calls (no process-wide or per-thread switches, counters or pools): the library is
 *     re-entrant - host threads may call it concurrently on different streams (tests/test_gpu_reentrant.py);
 *     the only per-thread datum is the message behind cim_last_error();
 *   - the last argument is the hipStream_t to launch on, passed as void*
 *     (torch.cuda.current_stream().cuda_stream);
 *   - "f16" arrays hold IEEE binary16 bit patterns (uint16_t).  Python-float thresholds are
 *     rounded to binary16 inside the library before comparing against f16 maps, which is
 *     what the reference's fp16-tensor-vs-Python-scalar compares do.
 *
 * Each entry cites the reference interface it replaces (paths under /root/reference).
 */
#ifndef CIM_HIP_H
#define CIM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char* cim_last_error(void);
int cim_abi_version(void);

/* ------------------------------------------------------------------ ROIAlign (a-1, a-2 prologue)
 * Replaces mmcv.ops.RoIAlign(output_size=P, spatial_scale, sampling_ratio, 'avg', aligned)
 * as imported by lib/ops/__init__.py:6 and called at lib/modeling/model_builder.py:229-231
 * (forward) and by autograd (backward).  Arithmetic: SURVEY.md App. D.
 *
 * Layout is channels-last: feat [B,H,W,C] f32, rois [K,5] f32 = (batch, x1, y1, x2, y2) in
 * input-image pixels, out [K,P,P,C] f32.  (The Python wrapper exposes these as logical NCHW
 * tensors in torch.channels_last memory format, so the reference's NCHW API is unchanged.)
 */
int cim_roi_align_fwd(const float* feat, const float* rois, float* out,
                      int B, int C, int H, int W, int K, int P,
                      float spatial_scale, int sampling_ratio, int aligned, void* stream);

/* grad_in [B,H,W,C] is fully overwritten (zero-filled on `stream`, then accumulated).
 * workspace: optional device scratch of cim_roi_align_bwd_workspace(K,P,H,W) bytes; when given, the
 * per-ROI interpolation tables are built once per launch instead of once per (ROI, channel chunk). */
long long cim_roi_align_bwd_workspace(int K, int P, int H, int W);
int cim_roi_align_bwd(const float* grad_out, const float* rois, float* grad_in,
                      int B, int C, int H, int W, int K, int P,
                      float spatial_scale, int sampling_ratio, int aligned, float* workspace, void* stream);

/* Forward with the backward's table workspace (cim_roi_align_bwd_workspace bytes): the per-ROI separable weight
 * tables are built once and every bin reads each pixel it touches ONCE with the aggregated weight
 * WY[ph][y]*WX[pw][x]/count - (gh+1)(gw+1) instead of 4*gh*gw loads per bin.  Same result up to the
 * reassociation of the sum (a few ulp; cim_roi_align_fwd keeps the reference's sample order bit for bit).
 * workspace == NULL behaves like cim_roi_align_fwd / cim_roi_align_maskcat_fwd. */
int cim_roi_align_fwd_ws(const float* feat, const float* rois, float* out,
                         int B, int C, int H, int W, int K, int P,
                         float spatial_scale, int sampling_ratio, int aligned, float* workspace, void* stream);
int cim_roi_align_maskcat_fwd_ws(const float* feat, const float* rois, const float* masks, float* cat,
                                 int B, int C, int H, int W, int K, int P,
                                 float spatial_scale, int sampling_ratio, int aligned, float* workspace, void* stream);

/* Backward with tables_ready != 0: `workspace` still holds the tables a *_fwd_ws call on the SAME rois / geometry
 * built (the aggregated-weight forward and the backward share them); 0 rebuilds them.
 * scratch: cim_roi_align_bwd_scratch(K,B,C,H,W) bytes (0: none needed) for the per-ROI-group partial maps of the
 * region-form backward, which are then summed by one streaming pass; with scratch == NULL the groups meet in
 * grad_in through float atomics (device-scope float atomics run at ~90 G/s on MI355X: 2-3x slower end to end). */
long long cim_roi_align_bwd_scratch(int K, int B, int C, int H, int W);
int cim_roi_align_bwd_ws(const float* grad_out, const float* rois, float* grad_in,
                         int B, int C, int H, int W, int K, int P,
                         float spatial_scale, int sampling_ratio, int aligned, float* workspace, int tables_ready,
                         float* scratch, void* stream);
int cim_roi_align_maskcat_bwd_ws(const float* grad_cat, const float* rois, const float* masks, float* grad_in,
                                 int B, int C, int H, int W, int K, int P,
                                 float spatial_scale, int sampling_ratio, int aligned, float* workspace, int tables_ready,
                                 float* scratch, void* stream);

/* Fused ROIAlign + mask multiply + channel concat: the input of MaskFuse.mask_branch,
 * lib/modeling/resnet50.py:121-134 (vgg16.py:162-175, HRNet.py:615-628):
 *   cat[k,ph,pw,0:C]  = roi_align(feat)[k,ph,pw,:]
 *   cat[k,ph,pw,C:2C] = roi_align(feat)[k,ph,pw,:] * masks[k,ph,pw]
 * masks [K,P,P] f32; cat [K,P,P,2C] f32.  box_x / mask_x are never materialised. */
int cim_roi_align_maskcat_fwd(const float* feat, const float* rois, const float* masks, float* cat,
                              int B, int C, int H, int W, int K, int P,
                              float spatial_scale, int sampling_ratio, int aligned, void* stream);

/* ROIAlign + mask multiply + channel concat + the Winograd INPUT TRANSFORM of MaskFuse.mask_branch's 3 x 3 convolution in one
 * launch (+ the table launch): lib/modeling/resnet50.py:121-135 up to the convolution's contraction.  Writes the pair image
 *   V [121][Rs][2C]  (mixed 4 + 3 tiling of the 7 x 7 map, cim_wino7_input_pair's layout; rows K .. Rs-1 zero; `scale` [121] from
 *   cim_wino7_pair_scales(kind 0))
 * that cim_gemm_pair_batched contracts with the filter image - `cat` is never stored (the two calls cim_roi_align_maskcat_fwd_ws
 * + cim_wino7_input_pair wrote and re-read its 4 * K * 49 * 2C bytes).  Bit-identical to those two calls.  P == 7, C % 8 == 0,
 * H, W <= 64; workspace: cim_roi_align_bwd_workspace(K,P,H,W) bytes, left holding the tables for cim_roi_align_maskcat_bwd_ws
 * (tables_ready = 1). */
int cim_roi_align_wino7_pair_fwd(const float* feat, const float* rois, const float* masks, void* V, const float* scale,
                                 int B, int C, int H, int W, int K, int Rs, int P, float spatial_scale, int sampling_ratio,
                                 int aligned, float* workspace, void* stream);

int cim_roi_align_maskcat_bwd(const float* grad_cat, const float* rois, const float* masks, float* grad_in,
                              int B, int C, int H, int W, int K, int P,
                              float spatial_scale, int sampling_ratio, int aligned, float* workspace, void* stream);

/* ------------------------------------------------------------------ mask IoU / containment maps (a-7)
 * Replaces lib/utils/mask_utils.py:6-18 (mask_iou) and :20-32 (mask_asymmetric_iou) as driven
 * by tools/pre/create_cob_iou.py:43-49 / create_cob_asy_iou.py:43-53, and the per-step
 * pickle.load + H2D at lib/modeling/model_builder.py:147-159.
 *   iou[i,j] = |m_i & m_j| / |m_i | m_j|      asy[i,j] = |m_i & m_j| / |m_j|
 * with the reference's rounding chain int64 -> f64 divide -> f32 -> f16.
 */
/* masks_u8 [N, HW] (non-zero = inside) -> packed [words, N] uint64 (WORD-MAJOR: packed[w*N + n]),
 * words = ceil(HW/64), bit b of word w = pixel 64*w+b, tail bits zero. */
int cim_mask_pack(const uint8_t* masks_u8, uint64_t* packed, int N, int HW, void* stream);

/* packed [words,N] (word-major) -> area [N] int32, iou_f16 [N,N], asy_f16 [N,N]. */
int cim_mask_iou_pair(const uint64_t* packed, int N, int words, int32_t* area,
                      uint16_t* iou_f16, uint16_t* asy_f16, void* stream);

/* ------------------------------------------------------------------ CIM mining (a-4)
 * Replaces CIM_layer.CIM_label / MIST_label / instance_nms, lib/modeling/heads.py:237-407.
 * All index outputs are bit-identical to the reference CPU path.
 */
/* heads.py:338: flag[i] = (#{j : asy[i,j] > con_thr}) < 0.9*N     (flag: uint8 [N]) */
int cim_asy_flag(const uint16_t* asy_f16, int N, float con_thr, uint8_t* flag, void* stream);

/* Everything the mining needs from the containment map ALONE (an input of the step: the host runs this on a side stream under
 * the backbone forward): flags [n_slots, N] = cim_asy_flag for each of the n_slots distinct con_thr values (con_thr_host: HOST
 * array), and asy_t [N, ldt] = the map transposed, rows padded to ldt = 8 ceil(N / 8) entries (16-byte aligned rows, pad entries
 * zero; may be NULL) - the containment step reads whole COLUMNS of the map (heads.py:386: every proposal against one seed),
 * contiguous rows of the transposed copy. */
int cim_asy_prep(const uint16_t* asy_f16, int N, const float* con_thr_host, int n_slots, uint8_t* flags, uint16_t* asy_t,
                 void* stream);

/* The whole mining + assignment of ONE training step - all REFINE_TIMES CIM layers - without a host round trip:
 * 2 launches (seed selection + containment arg-max + arbitration + anti-noise sampling; assignment).
 * Replaces, per layer, CIM_layer.forward = CIM_label / MIST_label + instance_nms + the sampling loop + the assignment,
 * lib/modeling/heads.py:237-503, as called three times per image from lib/modeling/model_builder.py:170-187.
 *
 * The image's classes are read on the device: class c is active iff labels[c] != 0 (heads.py:340); per-class outputs
 * are indexed by the class id c itself.
 *
 * Per layer (cim_mining_layer):
 *   seed_score  top-K ranking score, element (i, c) at seed_score[i*seed_ld + seed_off + c]        (heads.py:354 / :279)
 *   det         containment arg-max score, element (i, c) at det[i*det_ld + det_off + c*det_cs]; det_cs = 0 selects the
 *               class-agnostic detector (heads.py:346-347); unused (may be NULL) when using_cim == 0 (MIST_label)
 *   wa, wb      arbitration weight (i, c) = wa[i*wa_ld + wa_off + c] * (wb ? wb[i*wb_ld + wb_off + c*wb_cs] : 1)
 *               (preds = cls * det, heads.py:330,397; MIST: the product tensor itself, wb = NULL, :306)
 *   nms_thr, cls_thr, iou_thr, con_thr   Python-float thresholds (rounded to binary16 inside)
 *   using_cim   1: CIM_label (seeds -> containing proposals), 0: MIST_label (seeds are the pseudo GTs)
 *   anti_noise  1: resample each class's pseudo GTs in proportion to their weight (heads.py:447-473)
 *   flag_slot   which of the `flags` arrays (one per distinct con_thr, written by cim_asy_flag beforehand) this layer uses
 * Outputs per layer:
 *   topk [C,K] i32 (keep_sort_idx), seeds [C,K] i32 (keep_nms_idx in selection order, -1 beyond n_seeds[c]),
 *   n_seeds [C] i32, res [C,K] i32 (res_idx per seed, -1: no containing proposal / no seed; CIM layers only)
 *   gt_class [N] i32 (0 / class id + 1), gt_weight [N] f32 (-1 where unset)             after arbitration
 *   pre_idx [N] i32, pre_keep [N] u8: the G pseudo GTs in ascending proposal order and the sampling keep-mask over them
 *   gt_idx [N] i32, gt_cls [N] i32, gt_w [N] f32: the G' survivors;  counts [2] i32 = { G, G' }
 *   pseudo_labels [N,C+1] f32, pseudo_iou [N] binary16 {0,1}, loss_weights [N] f32, max_idx [N] i32   (heads.py:477-501;
 *   NOT written when G == 0: the reference returns (None, None, None) and the model skips the layer, :429-430)
 * Step-wide:
 *   flags [n_slots, N] u8 (cim_asy_flag), uniforms [max_uniforms] f64 = the next doubles of the host's MT19937 stream
 *   (np.random.random_sample), consumed in the order of the reference's sequential calls; used [1] i32 = how many were
 *   consumed; layer_valid [R] i32 = (G > 0); status [1] i32 error bits (0 = ok; 1: class list longer than K, 2: uniforms
 *   exhausted; the losses launch ORs 4 / 8 into it, see cim_loss_args).
 * Limits: N <= 8192, K <= 1024, R <= CIM_MAX_LAYERS. */
#define CIM_MAX_LAYERS 4
typedef struct cim_mining_layer {
    const float* seed_score; int32_t seed_ld, seed_off;
    const float* det;        int32_t det_ld, det_off, det_cs;
    const float* wa;         int32_t wa_ld, wa_off;
    const float* wb;         int32_t wb_ld, wb_off, wb_cs;
    float nms_thr, cls_thr, iou_thr, con_thr;
    int32_t using_cim, anti_noise, flag_slot, reserved_;
    int32_t* topk; int32_t* seeds; int32_t* n_seeds; int32_t* res;
    int32_t* gt_class; float* gt_weight;
    int32_t* pre_idx; uint8_t* pre_keep;
    int32_t* gt_idx; int32_t* gt_cls; float* gt_w; int32_t* counts;
    float* pseudo_labels; uint16_t* pseudo_iou; float* loss_weights; int32_t* max_idx;
} cim_mining_layer;

typedef struct cim_mining_args {
    int32_t N, C, K, R;
    const float* labels;             /* [C] image labels */
    const uint16_t* iou;             /* [N,N] binary16 mask-IoU map */
    const uint16_t* asy;             /* [N,N] binary16 containment map */
    const uint16_t* asy_t;           /* [N, 8 ceil(N/8)] its transpose from cim_asy_prep (NULL: the map's columns are read with strided loads) */
    const uint8_t* flags;            /* [n_slots, N] from cim_asy_flag */
    const double* uniforms;          /* [max_uniforms] */
    int32_t max_uniforms, reserved_;
    int32_t* used;                   /* [1] */
    int32_t* status;                 /* [1] */
    int32_t* layer_valid;            /* [R] */
    cim_mining_layer layer[CIM_MAX_LAYERS];
} cim_mining_args;

/* Dynamic LDS of the mining launch (must be <= 156 KiB).
 * sync: cim_mining_sync_bytes() bytes of caller-owned device scratch, 8-byte aligned, ZERO-FILLED ONCE by the caller when it
 * is allocated: the workgroups of the mining launch meet through it (arrival counters of the (class, layer) workgroups - a
 * layer's last arrival runs its arbitration -, the list lengths the layers hand each other for the position in the uniform
 * stream) and the call leaves every word zero again, so the same scratch serves the next call on that stream and the launches
 * can be captured into a HIP graph.  Calls that may be in flight TOGETHER (two streams) need a scratch each. */
long long cim_mining_lds_bytes(int N, int K);
long long cim_mining_sync_bytes(void);
int cim_mining_step(const cim_mining_args* args, void* sync, void* stream);

/* ------------------------------------------------------------------ network-input image (f-3: the data side of the step)
 * Replaces prep_im_for_blob(flag="ToTensor"), lib/utils/blob.py:93-147, called from lib/roi_data/minibatch.py:109-150
 * (training) and lib/core/test.py:464-473 via get_image_blob (inference):
 *   src_bgr [h,w,3] uint8 as cv2.imread returns it (device memory) [-> horizontal flip, minibatch.py:121-122]
 *   -> float32 -> cv2.resize(fx = fy = im_scale, INTER_LINEAR) -> np.uint8 -> BGR2RGB -> /255 -> (x - mean) / std
 * dst: float32 planes [3][H][W] of an NCHW blob: element (c,y,x) at dst[c*plane_stride + y*row_stride + x]
 * (H = round(h*im_scale), W = round(w*im_scale); strides let the caller write into a padded batch blob,
 * lib/utils/blob.py:59-83).  inv_scale = 1 / im_scale (double, as cv::resize uses it).
 * mean_std6_host: HOST pointer to {mean_r, mean_g, mean_b, std_r, std_g, std_b}.
 * The resize arithmetic restates OpenCV 4.x's INTER_LINEAR float path; cv2 is not in the reference tree: unpinned. */
int cim_image_prep(const uint8_t* src_bgr, int h, int w, float* dst, int H, int W, long long plane_stride, int row_stride,
                   double inv_scale, int hflip, const float* mean_std6_host, void* stream);

/* ------------------------------------------------------------------ max-pooling / nearest up-sampling of the bodies (a-11)
 * Replaces nn.MaxPool2d(3, 2, 1) of the ResNet stem (lib/modeling/resnet50.py:29, torchvision resnet50.maxpool), nn.MaxPool2d(2, 2)
 * of VGG16 (lib/modeling/vgg16.py:43,50,60) and nn.Upsample(scale_factor = 2^k, mode = 'nearest') of HRNet's fuse layers
 * (lib/modeling/HRNet.py:201).  x: NC planes [H][W] fp32 (NCHW contiguous).  ATen semantics: floor output size
 * (cim_maxpool2d_out_size), no dilation, the window is scanned rows first, the first maximum wins, NaN propagates.
 *   cim_maxpool2d_fwd: y [NC][Ho][Wo]; idx (may be NULL: inference / frozen stem) receives the arg-max as h * W + w
 *   cim_maxpool2d_bwd: dx [NC][H][W] = for every input pixel the sum of dy over the windows whose arg-max it is (a gather in
 *                      window order: deterministic, no atomics)
 *   cim_upsample_nearest_fwd: y [NC][H*scale][W*scale] = x[.., oh / scale, ow / scale]; accumulate != 0: y += ... (the fuse sum)
 *   cim_upsample_nearest_bwd: dx [NC][H][W] = sum of the scale x scale block of dy, rows first */
int cim_maxpool2d_out_size(int in, int k, int stride, int pad);
int cim_maxpool2d_fwd(const float* x, float* y, int* idx, int NC, int H, int W, int k, int stride, int pad, void* stream);
int cim_maxpool2d_bwd(const float* dy, const int* idx, float* dx, int NC, int H, int W, int k, int stride, int pad, void* stream);
int cim_upsample_nearest_fwd(const float* x, float* y, int NC, int H, int W, int scale, int accumulate, void* stream);
int cim_upsample_nearest_bwd(const float* dy, float* dx, int NC, int H, int W, int scale, void* stream);

/* ------------------------------------------------------------------ backbone 1x1 convolutions (a-11)
 * The 1 x 1 convolutions of the ResNet bottlenecks (torchvision Bottleneck conv1 / conv3 / downsample.0 as wrapped by
 * lib/modeling/resnet50.py:17-91) and their backward products as small-tile fp32-MFMA GEMMs, C[M,N] = A . B:
 *   A: a_mcontig == 0: element (m,k) at A[m*lda + k];  1: at A[k*lda + m]
 *   B: b_kcontig == 0: element (k,n) at B[k*ldb + n];  1: at B[n*ldb + k]
 *   forward (NCHW, one image): A = weight [Cout][Cin], B = x [Cin][HW];  dX: A = weight (a_mcontig), B = dy;
 *   dW: A = dy [Cout][HW], B = x [Cin][HW] (b_kcontig).
 * Epilogue (splits == 1): x_raw (optional) receives the plain product (the convolution output the BatchNorm backward
 * needs); C = relu?(acc * a[m] + b[m] (+ residual)), a = gamma*rsqrt(var+eps), b = beta - mean*a per ROW m (= output
 * channel) when gamma/beta/mean/var are given - the frozen-statistics BatchNorm (+ identity) (+ ReLU) that follows the
 * convolution, lib/modeling/resnet50.py:53-77 - else C = relu?(acc (+ residual)).
 * splits > 1 (cim_gemm_small_splits): split-K through `workspace` [splits][M][N] + a fixed-order reduce pass that applies
 * the same epilogue. */
int cim_gemm_small_splits(int M, int N, int K);
int cim_gemm_small_f32(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                       int a_mcontig, int b_kcontig, float* x_raw, const float* gamma, const float* beta,
                       const float* mean, const float* var, float eps, const float* residual, int relu, int splits,
                       float* workspace, void* stream);

/* Backward of conv1x1 -> frozen BatchNorm (+ residual) (+ ReLU) for B images in one call: BatchNorm / ReLU backward
 * (cim_bn_act_bwd), dx [B,cin,hw] = W^T . dconv, dw [cout,cin] = sum_b dconv_b . x_b^T.  dy, y (post-activation output; may
 * be NULL without relu), x_raw (convolution output) are [B,cout,hw]; x is the convolution input [B,cin,hw]; w [cout,cin].
 * dres / dgamma+dbeta / dx / dw may be NULL when not needed.  workspace: cim_conv1x1_bwd_workspace(B,cin,cout,hw) bytes.
 * side_stream (may be NULL): a second HIP stream the weight-gradient GEMM is enqueued on, next to the data-gradient GEMM on
 * `stream`, forked after the BatchNorm backward through fork_event (a hipEvent_t of the CALLER, created with
 * hipEventDisableTiming, passed as void*; needed whenever side_stream is given - the library creates no events).  join != 0:
 * `stream` waits for it through join_event (the caller's as well) before the call returns (stream order is all the
 * caller needs); join == 0: the CALLER makes `stream` wait for side_stream before dw (and the buffers x, workspace) are used or
 * reused - the weight gradients of a whole backward pass then run beside the data-gradient chain.  Both events are recorded and
 * waited on inside the call: the caller may pass the same pair to its next call.
 * Chaining two layers' backward without a BatchNorm-backward launch in between (round 3):
 *   in_gamma, in_var, in_eps (in_gamma may be NULL): frozen BatchNorm of the layer that PRODUCED x as relu(bn(conv)) with no
 *     residual - dx is then written as  x > 0 ? dx * in_gamma rsqrt(in_var + in_eps) : 0,  i.e. already the gradient of that
 *     layer's convolution output (its BatchNorm + ReLU backward in this product's epilogue);
 *   dy_is_dconv != 0: dy IS such a gradient of this layer's convolution output (handed over by the layer after it): the
 *     BatchNorm backward launch is skipped (dres must be NULL).
 * With TRAINABLE gamma / beta in the producing layer (the reference's configuration: lib/modeling/resnet50.py:59-60 leaves the
 * affine parameters of the frozen-statistics BatchNorm layers trainable) (round 4):
 *   in_xr, in_mean, in_part (in_part may be NULL): that layer's convolution output [B,cin,hw], its running mean, and
 *     [B][2][ceil(hw/32)][cin] floats that receive, per 32-pixel group g of every channel c, sum dz and sum dz (in_xr - in_mean)
 *     over the group (dz = x > 0 ? dx : 0) - the per-channel sums of that layer's BatchNorm backward, written by the same epilogue;
 *   a layer called with dy_is_dconv gets its own dgamma / dbeta from the array its consumer filled: cim_bn_part_finish below
 *     (dgamma, dbeta must be NULL in that call). */
long long cim_conv1x1_bwd_workspace(int B, int cin, int cout, int hw);
int cim_conv1x1_bn_act_bwd(const float* dy, const float* y, const float* x_raw, const float* x, const float* w,
                           const float* gamma, const float* mean, const float* var, float eps, int relu,
                           float* dres, float* dgamma, float* dbeta, float* dx, float* dw, int B, int cin, int cout, int hw,
                           float* workspace, void* stream, void* side_stream, void* fork_event, void* join_event, int join,
                           int dy_is_dconv, const float* in_gamma, const float* in_var, float in_eps,
                           const float* in_xr, const float* in_mean, float* in_part, const float* dx_add, int dx_add_w);
/* dx_add (may be NULL; round 4): [B,cin,hw] added to dx in the data gradient's epilogue - the gradient that reaches x through a
 * SECOND branch (the identity path of a bottleneck: lib/modeling/resnet50.py:17-44 torchvision Bottleneck `out += identity`, or
 * the block's downsample convolution), so that autograd needs no separate add launch per block.
 * dx_add_w = 0: dx_add has dx's shape.  dx_add_w = W > 0 (hw = H W): dx_add is [B,cin,ceil(H/2),ceil(W/2)], the data gradient of
 * a STRIDE-2 1 x 1 convolution of the same x - added at the pixels (2i, 2j) it belongs to (replaces autograd's zero-filled
 * scatter of the strided slice: two fills, two copies and an add per downsample block).
 * The affine gradients of chained layers from those partial sums, for n layers in ceil(n / 24) launches (`descs` is a HOST array):
 * dbeta[c] = sum over images and groups of part[b][0][g][c], dgamma[c] = rsqrt(var[c] + eps) sum part[b][1][g][c], in index
 * order (deterministic).  dgamma or dbeta may be NULL.  One call at the end of a backward pass serves the whole body. */
typedef struct { const float* part; const float* var; float eps; float* dgamma; float* dbeta; int images; int parts; int channels; } cim_bn_part_desc;
int cim_bn_part_finish(const cim_bn_part_desc* descs, int n, void* stream);

/* 3 x 3 convolution (padding 1, stride 1 or 2, no bias, groups 1) -> frozen BatchNorm (+ residual) (+ ReLU) of the
 * bottlenecks (torchvision Bottleneck.conv2 / bn2, lib/modeling/resnet50.py:17-44,53-77), NCHW fp32, ONE image per call, as an
 * implicit GEMM on the small-tile fp32-MFMA kernel above (the im2col matrix is never built): x [cin][H][W], w [cout][cin][3][3]
 * (the nn.Conv2d weight as it is), y / x_raw / residual [cout][Ho][Wo], Ho = (H-1)/stride + 1.  Epilogue and split-K as
 * cim_gemm_small_f32 (workspace [splits][cout][Ho Wo] floats when cim_conv3x3_nchw_splits(...) > 1).  cin % 4 == 0.
 * Replaces ATen -> MIOpen (miopenSp3AsmConv / Im2d2Col + rocBLAS) for those layers. */
int cim_conv3x3_nchw_splits(int cin, int cout, int H, int W, int stride);
int cim_conv3x3_nchw_f32(const float* x, const float* w, float* y, int cin, int cout, int H, int W, int stride, int dilation,
                         float* x_raw, const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                         const float* residual, int relu, int splits, float* workspace, void* stream);
/* dilation >= 1 with padding = dilation ("same"; dilation > 1 needs stride 1): the dilated conv5 of VGG16
 * (lib/modeling/vgg16.py:70-78).  A convolution with a bias and no BatchNorm is the same call with gamma = 1, beta = bias,
 * mean = 0, var = 1, eps = 0 (a = 1 exactly); a bias in front of a BatchNorm is folded into mean (mean - bias).  The forward
 * takes any cin (the frozen RGB stems); the backward needs cin % 4 == 0. */
/* The stem of the ResNet body: 7 x 7 convolution (padding 3, stride 1 or 2, no bias, any cin) -> frozen BatchNorm (+ ReLU),
 * forward only (torchvision ResNet conv1 / bn1 / relu, lib/modeling/resnet50.py:20,53-77, frozen by FREEZE_AT): the same
 * implicit GEMM with 49 taps.  x [cin][H][W], w [cout][cin][7][7], y [cout][Ho][Wo].  Replaces ATen -> MIOpen + 2 launches. */
int cim_conv7x7_nchw_f32(const float* x, const float* w, float* y, int cin, int cout, int H, int W, int stride,
                         const float* gamma, const float* beta, const float* mean, const float* var, float eps, int relu,
                         void* stream);
/* Backward of cim_conv3x3_nchw_f32 for B images in one call: BatchNorm / ReLU backward (cim_bn_act_bwd), dx [B,cin,H,W] (transposed
 * convolution as an implicit GEMM over (cout, tap)), dw [cout,cin,3,3] = sum over images and output pixels (split-K).
 * dy, y (may be NULL without relu), x_raw are [B,cout,Ho,Wo]; dres / dgamma+dbeta / dx / dw may be NULL when not needed.
 * workspace: cim_conv3x3_nchw_bwd_workspace(...) bytes.  cin % 4 == 0, cout % 4 == 0.  side_stream, fork_event, join_event, join,
 * dy_is_dconv and in_gamma / in_var / in_eps as in cim_conv1x1_bn_act_bwd. */
long long cim_conv3x3_nchw_bwd_workspace(int B, int cin, int cout, int H, int W, int stride);
/* cim_conv3x3_dx_parts (ABI 14): the number of 32-pixel groups `in_part` ([B][2][parts][cin]) must provide for this geometry - stride 1:
 * ceil(H W / 32); stride 2 (dilation 1): the data gradient runs by parity classes of the input pixels (a class only contracts over
 * the taps that reach it: 9 of the 36 (tap, class) pairs) and numbers its groups class by class, tile padded. */
int cim_conv3x3_dx_parts(int H, int W, int stride);
int cim_conv3x3_nchw_bn_act_bwd(const float* dy, const float* y, const float* x_raw, const float* x, const float* w,
                                const float* gamma, const float* mean, const float* var, float eps, int relu, float* dres,
                                float* dgamma, float* dbeta, float* dx, float* dw, int B, int cin, int cout, int H, int W,
                                int stride, int dilation, float* workspace, void* stream, void* side_stream, void* fork_event,
                                void* join_event, int join, int dy_is_dconv, const float* in_gamma, const float* in_var, float in_eps,
                                const float* in_xr, const float* in_mean, float* in_part, const float* wt_ready);
/* wt_ready (may be NULL): the weight transposed to [cout][3][3][cin] - the data gradient's A operand - when the caller made it
 * ahead of the pass; else the call makes it itself (one small launch per layer in front of the data gradient).
 * cim_conv3x3_wt_multi: those transposes for n layers in ceil(n / 16) launches (`descs` is a HOST array; weights only change at
 * the optimizer step, so a body's transposes are made once per step beside the forward pass). */
typedef struct { const float* w; float* wt; int cin; int cout; } cim_wt_desc;
int cim_conv3x3_wt_multi(const cim_wt_desc* descs, int n, void* stream);

/* ------------------------------------------------------------------ backbone BatchNorm chains (a-11)
 * Frozen-statistics BatchNorm (+ residual) (+ ReLU) of the ResNet / HRNet bodies, lib/modeling/resnet50.py:17-44,53-77
 * (every BN in eval(): running statistics, trainable affine), one launch each way instead of 2-3 / 3-4 ATen passes.
 * NCHW fp32; x, res, y, dy, dx, dres are [N,C,HW]; gamma, beta, mean, var, dgamma, dbeta are [C].
 *   fwd: y = relu?(x*a + b (+ res)),  a = gamma*rsqrt(var+eps),  b = beta - mean*a        (res may be NULL)
 *   bwd: dz = dy*(y>0 if relu); dx = dz*a; dres = dz; dgamma = sum dz*(x-mean)*rstd; dbeta = sum dz
 *        (dx, dres, and the dgamma/dbeta pair may be NULL when not needed; y may be NULL without relu).
 *        When cim_bn_act_bwd_chunks(N,C,HW) > 1 the per-channel sums are accumulated with atomicAdd: the caller
 *        zeroes dgamma / dbeta first. */
int cim_bn_act_fwd(const float* x, const float* res, const float* gamma, const float* beta, const float* mean,
                   const float* var, float eps, float* y, int N, int C, int HW, int relu, void* stream);
int cim_bn_act_bwd_chunks(int N, int C, int HW);
int cim_bn_act_bwd(const float* dy, const float* y, const float* x, const float* gamma, const float* mean,
                   const float* var, float eps, float* dx, float* dres, float* dgamma, float* dbeta,
                   int N, int C, int HW, int relu, void* stream);

/* ------------------------------------------------------------------ MaskFuse contractions (a-2)
 * The dense contractions behind MaskFuse, lib/modeling/resnet50.py:104-110,135-136 (Conv2d(2C,C,3,pad=1) in the mixed 4 + 3
 * Winograd tiling of the 7 x 7 ROI map, Linear(49C,4096), Linear(4096,4096)) and their autograd backward: fp32 in, fp32 out, on
 * ONE arithmetic engine and ONE convolution algorithm.  (Rounds 1-3 carried four engines and four algorithms behind environment
 * switches; the superseded ones are test infrastructure now: experiments/include/cim_exp.h.)
 */
/* The pair engine ("f16x2p"): every fp32 operand x is scaled by ONE power of two per matrix and split into two fp16 terms,
 * x * s = h + l (22 significant bits); a product is evaluated as h*l + l*h + h*h on v_mfma_f32_32x32x16_f16 with fp32
 * accumulation and rescaled exactly in the epilogue (error class of an fp32 GEMM, tests/test_gpu_gemm_pair.py).  The operands were split by
 * their PRODUCERS.  A "pair image" of a logical fp32 matrix [rows][cols] is [rows][ld / 8][h: 8 x f16 | l: 8 x f16]
 * (ld logical elements per row, a multiple of 8; 4 bytes per element) with x * s = h + l for ONE power-of-two scale s per
 * matrix (per batch entry).  The same image serves the product that contracts over its columns (K-contiguous use) and the
 * one that contracts over its rows (through the hardware transpose read of LDS), so it is written once.  The GEMM moves
 * 16-byte chunks HBM -> LDS by LDS-DMA; its loop holds MFMAs and LDS reads only.
 *   cim_gemm_pair[_batched]: C[M,N] = A . B * 1 / (a_scale * b_scale) (+ bias)(ReLU), fp32 C.  a_mcontig = 0: A element (m,k) at A[m*lda + k], 1: at A[k*lda + m]; b_kcontig = 0: B element (k,n) at B[k*ldb + n], 1: at B[n*ldb + k];
 *     K % 32 == 0 (zero-filled rows / columns pad the images), lda / ldb % 8 == 0, an M-contiguous A needs M % 8 == 0, an
 *     N-contiguous B needs N % 8 == 0.  a_scale / b_scale: [batch] device floats, the scales the images were written with.
 *     c_amax (optional): receives max |C| as an IEEE bit pattern through atomicMax (caller zeroes) - the scale source
 *     of the next split.
 *   cim_pair_scales: scale[i] = 2^(14 - exponent(amax[min(i, n_amax - 1)] * factor[i]))  (factor may be NULL); reduce_all != 0:
 *                    every scale from the MAXIMUM of the n_amax words (row / column maxima of a weight -> its one scale)
 *   cim_pair_split : fp32 X [batch][rows][ld] -> pair image [batch][rows_pad][ldp], rows >= `rows` zero-filled; relu_y
 *                    (optional, laid out as X): elements with relu_y <= 0 are written as 0 (a fused ReLU backward mask)
 *   cim_pair_amax  : max |x| bit pattern of n floats (atomicMax into a caller-zeroed word; X 16-byte aligned, any n)
 *   cim_pair_masked_stats: the ReLU backward of a fully connected layer in front of its split, dz = y > 0 ? dy : 0 on [rows][cols]
 *                    (never stored: cim_pair_split applies the same mask): max |dz| as cim_pair_amax and, when `part`
 *                    ([ceil(rows / 64)][cols], may be NULL) is given, the bias gradient's partial sums over chunks of 64 rows, each
 *                    added up in row order - the caller sums the chunks (lib/modeling/resnet50.py:107-110 seg_fc's ReLUs) */
/* max_workgroups (0 = one launch over all tiles): caps the launches of THIS product - it then goes out as consecutive launches
 * of at most that many workgroups.  A workgroup of this engine owns its CU (128 KB of LDS), so a capped product never holds
 * more CUs than that and leaves the rest of the chip to concurrent streams - used for the MaskFuse weight-gradient products
 * that run beside the backbone's backward (cim_amd/ops/maskfuse_pair.py).  Same tiles, same arithmetic, same bits. */
/* products: 3 = the fp32-class evaluation h*l + l*h + h*h (the training step's arithmetic); 1 = the h*h product alone - operands
 * with 11 significant bits, fp32 accumulation: the arithmetic class of TF32, which the reference's own convolutions / matmuls
 * run in on its hardware (torch 1.10 defaults; tools/train.py:153-154 touches cudnn.deterministic / benchmark only).  Reported
 * beside the headline as bench.py's extra.tf32_class, never used by default. */
/* form: 0 = 256 x 256 tiles, eight waves, 128 KB of LDS - a workgroup owns its CU (every product of the forward / data-gradient
 * chains); 1 = 128 x 256 tiles, four waves (one per SIMD, <= 256 registers each), 96 KB of LDS: half of the CU's registers and 64 KB of
 * its LDS stay free, so workgroups of kernels on OTHER streams run on the same CU beside it - the form of the MaskFuse weight
 * gradients that run beside the backbone's backward (a_mcontig = 1, b_kcontig = 0 only).  Same products in the same order per
 * output element: same bits as form 0. */
int cim_gemm_pair_splits(int M, int N, int K);
int cim_gemm_pair(const void* A, const void* B, float* C, const float* bias, int M, int N, int K,
                  int lda, int ldb, int ldc, int a_mcontig, int b_kcontig, int relu, int splits, float* workspace,
                  const float* a_scale, const float* b_scale, uint32_t* c_amax, int max_workgroups, int products, int form,
                  void* stream);
int cim_gemm_pair_batched(const void* A, const void* B, float* C, int M, int N, int K,
                          int lda, int ldb, int ldc, int a_mcontig, int b_kcontig,
                          int batch, long long a_bs, long long b_bs, long long c_bs,
                          const float* a_scale, const float* b_scale, int max_workgroups, int products, int form, void* stream);
int cim_pair_scales(const uint32_t* amax, int n_amax, const float* factor, float* scale, int n, int reduce_all, void* stream);
int cim_pair_split(const float* X, void* P, int rows, int rows_pad, int cols, int ld, int ldp, int batch,
                   long long x_bs, long long p_bs, const float* scale, const float* relu_y, void* stream);
int cim_pair_amax(const float* X, long long n, uint32_t* amax, void* stream);
int cim_pair_masked_stats(const float* dy, const float* y, int rows, int cols, float* part, uint32_t* amax, void* stream);

/* Producers of pair images for the MaskFuse convolution in the mixed 4 + 3 Winograd tiling (121 positions, P = 7) and for seg_fc.0's input:
 *   cim_wino7_pair_scales: scale[121] from max |d| of the untransformed tensor (the maximum of the n_amax bit patterns `amax`, times
 *        max(1, amax_mul[0]) when amax_mul is given: the {0, 1} masks of MaskFuse's concat): per position the bound
 *        (abs row sum)_i (abs row sum)_j max|d| of the transform.  kind 0: input (B^T), 1: filter (G), 2: dy for the weight
 *        gradient (GD), 3: dy for the adjoint data gradient (A)
 *   cim_wino7_input_pair : x [R,7,7,C] fp32 -> V [121][Rs][C] pair image (Rs >= R rows per position, rows >= R zeroed)
 *   cim_wino7_filter_pair: W [Cout,Cin,3,3] -> U' [121][Cout][Cin] pair image (ci contiguous)
 *   cim_wino7_dy_pair    : dy [R,7,7,C] -> D (adjoint = 0: GD dy GD^T) or E (adjoint = 1: A dy A^T) [121][Rs][C]
 *   cim_wino7_flatten_bwd_dy_pair: the backward of the (c, h, w) flatten + mask_branch's ReLU + BOTH transforms above in one launch:
 *        dX [R][C*49] (seg_fc.0's data gradient), relu_y [R,7,7,C] (saved conv output; NULL: no mask) -> E (adjoint = 1 image, may be
 *        NULL) and D (adjoint = 0 image, may be NULL), bit-identical to cim_flatten_chw_bwd_bias + cim_wino7_dy_pair x 2 without the
 *        masked gradient ever being stored; bias_partial [R][C] (may be NULL) as cim_flatten_chw_bwd_bias.  C % 256 == 0
 *   cim_wino7_output_amax: M [121][R][C] -> y [R,7,7,C] = A^T m A per tile (+ bias, ReLU); also reports max |y| (atomicMax, caller zeroes)
 *   cim_flatten_chw_pair : the (c, h, w) flatten of the NCHW `.view(N, -1)` on channels-last data, into a pair image [Rs][C*PP] (lib/modeling/resnet50.py:135) */
int cim_wino7_pair_scales(const uint32_t* amax, int n_amax, const uint32_t* amax_mul, int kind, float* scale, void* stream);
int cim_wino7_input_pair(const float* x, void* V, const float* scale, int R, int Rs, int C, void* stream);
int cim_wino7_filter_pair(const float* W, void* U, const float* scale, int Cout, int Cin, void* stream);
int cim_wino7_dy_pair(const float* dy, void* D, const float* scale, int R, int Rs, int C, int adjoint, void* stream);
int cim_wino7_flatten_bwd_dy_pair(const float* dX, const float* relu_y, void* E, const float* scale_e, void* D, const float* scale_d,
                                  float* bias_partial, int R, int Rs, int C, void* stream);
int cim_wino7_output_amax(const float* M, const float* bias, float* y, int R, int C, int relu, uint32_t* y_amax, void* stream);
int cim_flatten_chw_pair(const float* src, void* dst, const float* scale, int R, int Rs, int PP, int C, void* stream);
/* the flatten's backward: src [R][C][PP] -> dst [R][PP][C], zeroed where relu_y [R][PP][C] <= 0 (the conv's ReLU mask); also writes bias_partial [R][C] = sum over the PP pixels of the masked gradient (the
 * conv's bias gradient is its sum over R); bias_partial may be NULL */
int cim_flatten_chw_bwd_bias(const float* src, const float* relu_y, float* dst, float* bias_partial, int R, int PP, int C,
                             void* stream);


/* fp32 stages of the same tiling that the pair engine's results pass through (tile must be 7):
 *   cim_wino_wgrad_output     : dU [121][Cin][Cout] (fp32, from cim_gemm_pair_batched) -> dW [Cout,Cin,3,3]
 *   cim_wino_dx_adjoint_output: Md [121][R][Cin] -> dx [R,7,7,Cin]   (overlap-add of B Md B^T: the adjoint of the forward) */
int cim_wino_wgrad_output(const float* dU, float* dW, int Cout, int Cin, int tile, void* stream);
int cim_wino_dx_adjoint_output(const float* M, float* dx, int R, int P, int C, int tile, void* stream);
/* cim_wino_dx_adjoint_output (tile = 7) with the backward of MaskFuse's prologue folded in: Md [121][R][2 Cb] is the Winograd-domain
 * gradient of cat = [box, box * mask] (lib/modeling/resnet50.py:131-134); written is  dbox [R,7,7,Cb] = dcat[..., :Cb] + mask *
 * dcat[..., Cb:]  (masks [R,7,7]) - the input of the plain ROIAlign backward (cim_roi_align_bwd_ws), half the bytes of dcat. */
int cim_wino7_dx_maskfold(const float* M, const float* masks, float* dbox, int R, int Cb, void* stream);

/* ------------------------------------------------------------------ losses (a-8, a-9, a-10)
 * Replaces cls_iou_loss + loss_weight_bag_loss (per refinement layer), mil_bag_loss and PCL_loss,
 * lib/modeling/heads.py:10-166, as summed by lib/modeling/model_builder.py:170-204.
 * One launch: R (= REFINE_TIMES <= 3) + 2 workgroups.  All score tensors are [N, C1] f32.
 *   part [R+2][4]  per-job partial losses in the order {bag, pcl, cls, iou}: rows 0..R-1 = the
 *                  refinement layers (iou NOT yet multiplied by 3), row R = mil_bag, row R+1 = PCL
 *   grad [3 + 4R][N][C1]  gradient components for a unit upstream gradient:
 *        0: d bag(mil)/d predict_cls   1: d pcl/d predict_cls   2: d bag(mil)/d predict_det
 *        3+4i: d cls_i/d refine_cls[i]   4+4i: d bag_i/d refine_cls[i]
 *        5+4i: d iou_i/d refine_iou[i]   6+4i: d bag_i/d refine_iou[i]
 * The PCL cluster structure of `mat` (heads.py:14-21: distinct non-zero ids ascending, member rows, the column of each
 * row's entry, the background cluster = the id found in column 0) is derived inside the launch by the PCL workgroup:
 * no host copy of `mat`, no per-image plan.  layer_valid comes straight from cim_mining_step. */
#define CIM_PCL_MAX_CLUSTERS 256
typedef struct cim_loss_args {
    const float* pc;                 /* predict_cls */
    const float* pd;                 /* predict_det */
    const float* rc[3];              /* refine_cls[i] */
    const float* ri[3];              /* refine_iou[i] */
    const float* pseudo_labels[3];   /* [N,C1] one-hot / zero rows (CIM_layer output) */
    const uint16_t* pseudo_iou_f16[3]; /* [N] binary16 {0,1} */
    const float* loss_weights[3];    /* [N] (unscaled) */
    float weight_scale[3];           /* lmda: 3 for layer 0, else 1 (model_builder.py:172) */
    const int32_t* layer_valid;      /* DEVICE [R]: 0 = CIM_layer found no pseudo GT -> layer skipped (cim_mining_step) */
    const float* labels;             /* [C1-1] image labels */
    const float* mat;                /* [N,C1] PRM cluster matrix (heads.py:10-41): <= 1 non-zero (= cluster id) per row */
    int32_t* status;                 /* DEVICE [1] error bits, OR-ed: 4 = a row of mat has several non-zeros (use the
                                        general ATen formulation), 8 = several distinct ids in column 0 (heads.py:20),
                                        16 = more than CIM_PCL_MAX_CLUSTERS clusters */
    int N, C1, R;
    float* part;
    float* grad;
    int ld;                          /* row stride (floats) of pc / pd / rc / ri: C1 (or 0) for dense [N,C1] tensors, 8 C1 when
                                        they are the column blocks of the heads' fused score matrix [N, (2 + 2R) C1] */
} cim_loss_args;

int cim_losses_fwd(const cim_loss_args* args, void* stream);

/* ------------------------------------------------------------------ head activations (a-3)
 * Replaces the softmax / sigmoid epilogue of cls_iou_model.forward, lib/modeling/heads.py:199-217.
 * logits / scores [N, (2+2R)*C1] with column blocks [classifier | detector | refine_cls[R] | refine_iou[R]]:
 * softmax over classes (classifier, refine_cls), softmax over PROPOSALS (detector), sigmoid (refine_iou).
 * colstat / coldot: [2*C1] / [C1] f32 scratch for the detector's column reductions. */
/* The linear part of the eight heads on the small-tile fp32-MFMA GEMM (no library GEMM on the path):
 *   cim_linear_bias_f32: Y[M][N] = X[M][K] . W[N][K]^T + bias[N]   (splits: cim_gemm_small_splits(M, N, K))
 * cim_loss_finish: cim_losses_fwd's partial sums part[rows][4] -> out[6] = (bag, pcl, cls, iou, 3 iou, total) on the device, with
 *   total = ((bag + pcl) + cls) + 3 iou: the IoU loss's weight of lib/modeling/model_builder.py:199 and the sum the driver
 *   differentiates (lib/utils/training_stats.py:72-83 `total_loss += loss` in dictionary order) in the same launch.
 * cim_loss_grad_combine: the gradient of the losses w.r.t. the fused score matrix [N][(2 + 2R) C1] from cim_losses_fwd's `grad`
 *   components and the upstream gradients of cim_loss_finish's six outputs (device scalars; NULL = zero). */
int cim_linear_bias_f32(const float* X, const float* W, const float* bias, float* Y, int M, int N, int K, int splits,
                        float* workspace, void* stream);
int cim_loss_finish(const float* part, int rows, float* out, void* stream);
int cim_loss_grad_combine(const float* G, const float* g_bag, const float* g_pcl, const float* g_cls, const float* g_iou,
                          const float* g_iou3, const float* g_total, float* out, int N, int C1, int R, void* stream);
int cim_head_act_fwd(const float* logits, float* scores, float* colstat, int N, int C1, int R, void* stream);
int cim_head_act_bwd(const float* scores, const float* grad_scores, float* grad_logits, float* coldot,
                     int N, int C1, int R, void* stream);

/* ------------------------------------------------------------------ optimizer step
 * Fused multi-tensor SGD with momentum and weight decay: torch.optim.SGD's update (dampening 0, no Nesterov) as
 * constructed at tools/train.py:282-311 and stepped at :438, ONE launch for all tensors:
 *     g' = g + wd*p ;  buf = momentum*buf + g' ;  p = p - lr*buf        (buf zero-initialised by the caller)
 * `tensors` and `chunks` are DEVICE arrays: one record per tensor (refreshed whenever a pointer, lr or wd changes) and one
 * per chunk of a tensor (the host uses 16384-element chunks; offsets are multiples of 4 elements); a workgroup streams
 * one chunk.  16-byte accesses are used when the three pointers of a tensor are 16-byte aligned. */
typedef struct cim_sgd_tensor {
    uint64_t p;        /* float* parameter (device address) */
    uint64_t g;        /* const float* gradient */
    uint64_t buf;      /* float* momentum buffer */
    int64_t n;         /* elements */
    float lr, wd;
    /* matrix mode (cols > 0): the tensor is a [rows][cols] matrix (cols % 4 == 0, 16-byte aligned pointers), its chunks are
     * 64-row x 1024-column tiles (chunk.offset = first row, chunk.n = first column) and the launch also accumulates
     * max |w_new| per row / per column (IEEE bit patterns, atomicMax) into the caller-zeroed arrays row_amax [rows] /
     * col_amax [cols] - the operand scales of the f16x2 contraction engine (cim_amax_rowcol) as a by-product. */
    int32_t rows, cols;
    uint64_t row_amax, col_amax;
} cim_sgd_tensor;
typedef struct cim_sgd_chunk {
    int32_t tensor;    /* index into `tensors` */
    int32_t n;         /* elements of this chunk (clamped to the tensor's end in the kernel) */
    int64_t offset;    /* first element */
} cim_sgd_chunk;
/* max_workgroups (ABI 14): 0 = one workgroup per chunk; > 0: at most that many workgroups, each walking over chunks b, b + grid, ...
 * (an update that runs beside another stream's latency-bound launches holds one or two workgroup slots per CU instead of all). */
int cim_sgd_multi(const cim_sgd_tensor* tensors, const cim_sgd_chunk* chunks, int n_chunks, float momentum, int max_workgroups,
                  void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CIM_HIP_H */
