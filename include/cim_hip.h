/*
 * cim_hip.h - C ABI of libcim_hip.so, the MI355X (gfx950) kernel library behind the
 * per-image training step of ZechengLi19/CIM.
 *
 * Conventions (every entry point):
 *   - returns 0 on success, a positive hipError_t on a HIP failure, -1 on bad arguments;
 *     cim_last_error() returns a thread-local description of the last failure;
 *   - every pointer is a caller-owned DEVICE pointer to a dense array of the stated dtype
 *     (the Python host passes torch.Tensor.data_ptr()); nothing is allocated or freed,
 *     no host synchronisation happens, no global state is kept (re-entrant);
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
 * built (the aggregated-weight forward and the gather backward share them); 0 rebuilds them. */
int cim_roi_align_bwd_ws(const float* grad_out, const float* rois, float* grad_in,
                         int B, int C, int H, int W, int K, int P,
                         float spatial_scale, int sampling_ratio, int aligned, float* workspace, int tables_ready, void* stream);
int cim_roi_align_maskcat_bwd_ws(const float* grad_cat, const float* rois, const float* masks, float* grad_in,
                                 int B, int C, int H, int W, int K, int P,
                                 float spatial_scale, int sampling_ratio, int aligned, float* workspace, int tables_ready, void* stream);

/* Fused ROIAlign + mask multiply + channel concat: the input of MaskFuse.mask_branch,
 * lib/modeling/resnet50.py:121-134 (vgg16.py:162-175, HRNet.py:615-628):
 *   cat[k,ph,pw,0:C]  = roi_align(feat)[k,ph,pw,:]
 *   cat[k,ph,pw,C:2C] = roi_align(feat)[k,ph,pw,:] * masks[k,ph,pw]
 * masks [K,P,P] f32; cat [K,P,P,2C] f32.  box_x / mask_x are never materialised. */
int cim_roi_align_maskcat_fwd(const float* feat, const float* rois, const float* masks, float* cat,
                              int B, int C, int H, int W, int K, int P,
                              float spatial_scale, int sampling_ratio, int aligned, void* stream);

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

/* Step 1 (heads.py:354-380 / 279-304): for each of the n_cls image classes (class ids in
 * `classes`, ascending), stable-descending top-K of seed_score[:, c] and greedy NMS over
 * the K x K sub-block of iou (keep dst iff iou < nms_thr, fp16 compare).
 *   seed_score: f32, element (i, c) at seed_score[i*score_ld + score_off + c]
 *   topk_idx  [n_cls, K] int32  (keep_sort_idx)
 *   seeds     [n_cls, K] int32  (keep_nms_idx, selection order; entries >= n_seeds are -1)
 *   n_seeds   [n_cls]    int32
 * Limits: N <= 8192, K <= 1024. */
int cim_seed_select(const float* seed_score, int score_ld, int score_off, const uint16_t* iou_f16, int N,
                    const int32_t* classes, int n_cls, int K, float nms_thr,
                    int32_t* topk_idx, int32_t* seeds, int32_t* n_seeds, void* stream);

/* Step 2 (heads.py:386-395): for every seed s of class c, among proposals i with
 * asy[i,s] > con_thr and flag[i], the first index maximising det[i,c]
 * (det: element (i,c) at det[i*det_ld + det_off + c*det_cstride]; det_cstride = 0 selects the
 * class-agnostic detector, heads.py:346-347).  res_idx [n_cls,K] int32, -1 where the seed
 * has no containing proposal (column dropped at heads.py:392) or s >= n_seeds. */
int cim_contain_argmax(const uint16_t* asy_f16, const uint8_t* flag, const float* det, int det_ld, int det_off,
                       int det_cstride, int N, const int32_t* classes, int n_cls, int K, float con_thr,
                       const int32_t* seeds, const int32_t* n_seeds, int32_t* res_idx, void* stream);

/* Cross-class arbitration + compaction (heads.py:397-405 for CIM, :306-314 for MIST).
 * cand [n_cls,K] int32 candidate proposals per class (-1 = none; duplicates allowed),
 * weight element (i,c) = wa[i*wa_ld + wa_off + c] * (wb ? wb[i*wb_ld + wb_off + c*wb_cstride] : 1).
 * Classes are applied sequentially in ascending order with the strict '>' rule.
 * Outputs: gt_class [N] int32 (0 = not a pseudo GT, else class id + 1), gt_weight [N] f32
 * (-1 where unset), and gt_pack [1 + 3N] int32 = { G, idx[N], class[N], weight_bits[N] } with
 * the G pseudo GTs compacted in ascending proposal order (one D2H copy feeds the host-side
 * anti-noise sampling of heads.py:447-473). */
int cim_arbitrate(const int32_t* cand, const int32_t* classes, int n_cls, int K, int N,
                  const float* wa, int wa_ld, int wa_off,
                  const float* wb, int wb_ld, int wb_off, int wb_cstride,
                  int32_t* gt_class, float* gt_weight, int32_t* gt_pack, void* stream);

/* ------------------------------------------------------------------ assignment (a-6)
 * Replaces lib/modeling/heads.py:435,477-501.  gt_idx [G] int32 ascending proposal indices of
 * the surviving pseudo GTs (after anti-noise sampling, done on the host with NumPy's global
 * RNG exactly as heads.py:447-473), gt_cls [G] int32 (class id + 1), gt_w [G] f32.
 * Outputs: pseudo_labels [N, C1] f32 one-hot rows (C1 = classes + 1), pseudo_iou_f16 [N]
 * ({0,1} in binary16, like the reference), loss_weights [N] f32, max_idx [N] int32. */
int cim_assign(const uint16_t* iou_f16, int N, const int32_t* gt_idx, const int32_t* gt_cls, const float* gt_w,
               int G, int C1, float cls_thr, float iou_thr,
               float* pseudo_labels, uint16_t* pseudo_iou_f16, float* loss_weights, int32_t* max_idx,
               void* stream);

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
 * fp32-in / fp32-out MFMA GEMMs replacing the ATen/cuDNN calls behind MaskFuse,
 * lib/modeling/resnet50.py:104-110,135-136 (Conv2d(2C,C,3,pad=1), Linear(49C,4096),
 * Linear(4096,4096)) and their autograd backward.  Two arithmetic engines, same results class:
 *   engine 1 (default): every fp32 operand is split exactly into three bf16 terms in the kernel and the
 *     product evaluated as six v_mfma_f32_32x32x16_bf16 products with fp32 accumulation
 *     (dropped terms < 2^-23 |a*b|: two orders below the fp32 accumulation rounding itself);
 *   engine 0: v_mfma_f32_32x32x2_f32 (f32 multiplies).
 * cim_gemm_set_engine() selects process-wide (host code sets it from CIM_GEMM_ENGINE = bf16x3 | fp32).
 *
 * C[M,N] = A . B (+ bias[N]) (ReLU optional), row-major C with leading dimension ldc.
 *   a_mcontig = 0: A element (m,k) at A[m*lda + k];  1: at A[k*lda + m]
 *   b_kcontig = 0: B element (k,n) at B[k*ldb + n];  1: at B[n*ldb + k]   (nn.Linear weight)
 * splits > 1: split-K through `workspace` (splits*M*N floats), reduced in a fixed order
 * (deterministic).  cim_gemm_f32_splits() returns the split count the library would choose. */
int cim_gemm_set_engine(int engine);
int cim_gemm_get_engine(void);
int cim_gemm_f32_splits(int M, int N, int K);
int cim_gemm_f32(const float* A, const float* B, float* C, const float* bias, int M, int N, int K,
                 int lda, int ldb, int ldc, int a_mcontig, int b_kcontig, int relu,
                 int splits, float* workspace, void* stream);

/* 3x3 / stride 1 / pad 1 convolution on R independent P x P maps as an implicit GEMM
 * (no im2col buffer): X [R,P,P,Cin] (NHWC), Whwio [3,3,Cin,Cout], Y [R,P,P,Cout].
 * The data gradient is the same call on dY with the spatially flipped, in/out-swapped weights. */
int cim_conv3x3_f32(const float* X, const float* Whwio, const float* bias, float* Y,
                    int R, int P, int Cin, int Cout, int relu, void* stream);

/* Weight gradient: dWhwio [3,3,Cin,Cout] = im2col(X)^T . dY, dY [R,P,P,Cout]. */
int cim_conv3x3_wgrad_f32(const float* X, const float* dY, float* dWhwio,
                          int R, int P, int Cin, int Cout, int splits, float* workspace, void* stream);

/* `batch` independent GEMMs of identical shape in one launch (strides in elements between
 * consecutive problems); used for the 16 positions of the Winograd-domain convolution. */
int cim_gemm_f32_batched(const float* A, const float* B, float* C, int M, int N, int K,
                         int lda, int ldb, int ldc, int a_mcontig, int b_kcontig,
                         int batch, long long a_bs, long long b_bs, long long c_bs, void* stream);

/* f16x2 engine (the host's default, CIM_GEMM_ENGINE=f16x2): the same contractions as cim_gemm_f32 /
 * cim_gemm_f32_batched with every fp32 operand scaled by a power of two per A row / per B column and split
 * into TWO fp16 terms (x*s = h + l, 23 significant bits), evaluated as the three products hl + lh + hh on
 * v_mfma_f32_32x32x16_f16 with fp32 accumulation and rescaled exactly in the epilogue.  Error bound of an
 * fp32 GEMM relative to |a_row|*|b_col| (dropped l*l term <= 2^-22, rms 2^-25.6, per product); half the MFMA
 * work of the bf16x3 engine.
 *
 * cim_amax_rowcol: X is a stored [batch][rows][ld] fp32 matrix (cols used).  row_amax [batch*rows] /
 * col_amax [batch*cols] (either may be NULL) receive max |x| as IEEE bit patterns through atomicMax, so the
 * CALLER ZEROES them first.  One pass over X serves both orientations of the operand.
 * a_amax [batch][M]: the row array of a K-contiguous A, the column array of the stored matrix of an
 * M-contiguous A;  b_amax [batch][N]: the column array of an N-contiguous B, the row array of a K-contiguous B. */
int cim_amax_rowcol(const float* X, int rows, int cols, int ld, int batch, long long bs,
                    uint32_t* row_amax, uint32_t* col_amax, void* stream);
int cim_gemm_f16x2_splits(int M, int N, int K);
int cim_gemm_f16x2(const float* A, const float* B, float* C, const float* bias, int M, int N, int K,
                   int lda, int ldb, int ldc, int a_mcontig, int b_kcontig, int relu,
                   int splits, float* workspace, const uint32_t* a_amax, const uint32_t* b_amax, void* stream);
int cim_gemm_f16x2_batched(const float* A, const float* B, float* C, int M, int N, int K,
                           int lda, int ldb, int ldc, int a_mcontig, int b_kcontig,
                           int batch, long long a_bs, long long b_bs, long long c_bs,
                           const uint32_t* a_amax, const uint32_t* b_amax, void* stream);

/* Winograd F(2x2,3x3) evaluation of the same 3x3 / stride 1 / pad 1 convolution (fp32 throughout,
 * 1.72x fewer multiplies at P = 7): T = ceil(P/2) tiles per side, 16 transform positions.
 *   cim_wino_input_transform : x [R,P,P,C]            -> V [16][R*T*T][C]       (B^T d B)
 *   cim_wino_filter_transform: W [Cout,Cin,3,3]       -> U [16][K][N]           (G g G^T)
 *        mode 0: K = Cin, N = Cout (forward);  mode 1: K = Cout, N = Cin, taps rotated (data gradient)
 *   (16 GEMMs  M[pos] = V[pos] . U[pos]  through cim_gemm_f32_batched)
 *   cim_wino_output_transform: M [16][R*T*T][C], bias -> y [R,P,P,C]            (A^T m A, +bias, ReLU)
 *   weight gradient: cim_wino_dy_transform: dy [R,P,P,C] -> D [16][R*T*T][C]    (G2 dy G2^T)
 *        16 GEMMs dU[pos] = V[pos]^T . D[pos];  cim_wino_wgrad_output: dU [16][Cin][Cout] -> dW [Cout,Cin,3,3]
 * `tile` = 2: F(2x2,3x3), 16 positions (default);  `tile` = 4: F(4x4,3x3) on the points {0,1,-1,2,-1/2,inf},
 * 36 positions, T = ceil(P/4), 3.1x fewer multiplies than direct at P = 7, fp32 error ~7e-6. */
int cim_wino_input_transform(const float* x, float* V, int R, int P, int C, int tile, void* stream);
/* (c, h, w) flatten between the conv and seg_fc.0 (resnet50.py:135, `.view(N, -1)` of an NCHW tensor) on channels-last
 * data.  backward = 0: src [R][PP][C] -> dst [R][C][PP].  backward = 1: src [R][C][PP] -> dst [R][PP][C], zeroed where
 * relu_y [R][PP][C] <= 0 (the conv's ReLU mask; NULL = no mask).  PP <= 64, C % 64 == 0. */
int cim_flatten_chw(const float* src, const float* relu_y, float* dst, int R, int PP, int C, int backward, void* stream);

/* f16x2 engine helpers (tile = 4 only):
 * cim_wino_input_transform_amax: the input transform that also stores row_amax [36][R*T*T], an upper bound of
 *   max |V[pos][m][:]| (bit patterns; f_pos * max |x| over the tile's patch and all channels; plain stores, nothing
 *   to zero) - the per-row operand scales of the forward / data-gradient GEMMs.
 * cim_wino_scale_bounds: per-column scale BOUNDS of a transformed operand from the |max| of the untransformed
 *   tensor, bounds [36][n] = f_pos * max_{t<group} amax_in[n*group + t], f_pos = product of the absolute row sums
 *   of the transform matrix (kind 0: B^T (input), 1: G (filter), 2: G4 (output gradient)). */
int cim_wino_input_transform_amax(const float* x, float* V, uint32_t* row_amax, int R, int P, int C, int tile, void* stream);
int cim_wino_scale_bounds(const uint32_t* amax_in, uint32_t* bounds, int n, int group, int kind, int tile, void* stream);
/* Data gradient of the mixed tiling (tile = 7) as the ADJOINT of the forward - reuses the forward's U, no transform of a
 * rotated filter:  E = A dy A^T per tile (cim_wino_dy_adjoint_transform; row_amax [121][R] optional: row-scale bounds of E),
 * 121 GEMMs Md[pos] = E[pos] . U[pos]^T (U [121][Cin][Cout] read K-contiguously), dx = overlap-add of B Md B^T
 * (cim_wino_dx_adjoint_output).  E [121][R][Cout], Md [121][R][Cin], dx [R,7,7,Cin]. */
int cim_wino_dy_adjoint_transform(const float* dy, float* E, uint32_t* row_amax, int R, int P, int C, int tile, void* stream);
int cim_wino_dx_adjoint_output(const float* M, float* dx, int R, int P, int C, int tile, void* stream);
int cim_wino_filter_transform(const float* W, float* U, int Cout, int Cin, int mode, int tile, void* stream);
int cim_wino_output_transform(const float* M, const float* bias, float* y, int R, int P, int C, int relu, int tile, void* stream);
int cim_wino_dy_transform(const float* dy, float* D, int R, int P, int C, int tile, void* stream);
int cim_wino_wgrad_output(const float* dU, float* dW, int Cout, int Cin, int tile, void* stream);

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
 * PCL cluster plan (host-built from `mat`, heads.py:14-21): row_cluster [N] = index of the row's
 * cluster (-1: none), row_col [N] = column of its non-zero entry, cluster_size [K], bg_cluster =
 * index of the background cluster (-1: none). */
typedef struct cim_loss_args {
    const float* pc;                 /* predict_cls */
    const float* pd;                 /* predict_det */
    const float* rc[3];              /* refine_cls[i] */
    const float* ri[3];              /* refine_iou[i] */
    const float* pseudo_labels[3];   /* [N,C1] one-hot / zero rows (CIM_layer output) */
    const uint16_t* pseudo_iou_f16[3]; /* [N] binary16 {0,1} */
    const float* loss_weights[3];    /* [N] (unscaled) */
    float weight_scale[3];           /* lmda: 3 for layer 0, else 1 (model_builder.py:172) */
    int layer_valid[3];              /* 0: CIM_layer returned None -> layer skipped */
    const float* labels;             /* [C1-1] image labels */
    const int32_t* row_cluster;
    const int32_t* row_col;
    const int32_t* cluster_size;
    int K, bg_cluster;
    int N, C1, R;
    float* part;
    float* grad;
} cim_loss_args;

int cim_losses_fwd(const cim_loss_args* args, void* stream);

/* ------------------------------------------------------------------ head activations (a-3)
 * Replaces the softmax / sigmoid epilogue of cls_iou_model.forward, lib/modeling/heads.py:199-217.
 * logits / scores [N, (2+2R)*C1] with column blocks [classifier | detector | refine_cls[R] | refine_iou[R]]:
 * softmax over classes (classifier, refine_cls), softmax over PROPOSALS (detector), sigmoid (refine_iou).
 * colstat / coldot: [2*C1] / [C1] f32 scratch for the detector's column reductions. */
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
int cim_sgd_multi(const cim_sgd_tensor* tensors, const cim_sgd_chunk* chunks, int n_chunks, float momentum, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CIM_HIP_H */
