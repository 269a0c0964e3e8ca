/*
 * cim_exp.h - entry points of experiments/libcim_exp.so: the SUPERSEDED arithmetic engines and convolution algorithms of the MaskFuse
 * contractions (rounds 1-3), kept as test infrastructure only - tests/test_gpu_tolerance.py measures how far each of them is from
 * the reference, tests/test_gpu_experiments.py keeps them honest.  Nothing under cim_amd/ loads this library; the product has ONE
 * engine (cim_gemm_pair*, include/cim_hip.h) and ONE algorithm (the mixed 4 + 3 Winograd tiling).
 * Conventions as in include/cim_hip.h.
 */
#ifndef CIM_EXP_H
#define CIM_EXP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ MaskFuse contractions (a-2)
 * fp32-in / fp32-out MFMA GEMMs replacing the ATen/cuDNN calls behind MaskFuse,
 * lib/modeling/resnet50.py:104-110,135-136 (Conv2d(2C,C,3,pad=1), Linear(49C,4096),
 * Linear(4096,4096)) and their autograd backward.  Two arithmetic engines, same results class:
 *   engine 1 (default): every fp32 operand is split exactly into three bf16 terms in the kernel and the
 *     product evaluated as six v_mfma_f32_32x32x16_bf16 products with fp32 accumulation
 *     (dropped terms < 2^-23 |a*b|: two orders below the fp32 accumulation rounding itself);
 *   engine 0: v_mfma_f32_32x32x2_f32 (f32 multiplies).
 * `engine` is an ARGUMENT of every such call (0 or 1): the library keeps no engine switch.
 *
 * C[M,N] = A . B (+ bias[N]) (ReLU optional), row-major C with leading dimension ldc.
 *   a_mcontig = 0: A element (m,k) at A[m*lda + k];  1: at A[k*lda + m]
 *   b_kcontig = 0: B element (k,n) at B[k*ldb + n];  1: at B[n*ldb + k]   (nn.Linear weight)
 * splits > 1: split-K through `workspace` (splits*M*N floats), reduced in a fixed order
 * (deterministic).  cim_gemm_f32_splits() returns the split count the library would choose. */
int cim_gemm_f32_splits(int M, int N, int K, int engine);
int cim_gemm_f32(const float* A, const float* B, float* C, const float* bias, int M, int N, int K,
                 int lda, int ldb, int ldc, int a_mcontig, int b_kcontig, int relu,
                 int splits, float* workspace, int engine, void* stream);

/* 3x3 / stride 1 / pad 1 convolution on R independent P x P maps as an implicit GEMM
 * (no im2col buffer): X [R,P,P,Cin] (NHWC), Whwio [3,3,Cin,Cout], Y [R,P,P,Cout].
 * The data gradient is the same call on dY with the spatially flipped, in/out-swapped weights. */
int cim_conv3x3_f32(const float* X, const float* Whwio, const float* bias, float* Y,
                    int R, int P, int Cin, int Cout, int relu, int engine, void* stream);

/* Weight gradient: dWhwio [3,3,Cin,Cout] = im2col(X)^T . dY, dY [R,P,P,Cout]. */
int cim_conv3x3_wgrad_f32(const float* X, const float* dY, float* dWhwio,
                          int R, int P, int Cin, int Cout, int splits, float* workspace, int engine, void* stream);

/* `batch` independent GEMMs of identical shape in one launch (strides in elements between
 * consecutive problems); used for the 16 positions of the Winograd-domain convolution. */
int cim_gemm_f32_batched(const float* A, const float* B, float* C, int M, int N, int K,
                         int lda, int ldb, int ldc, int a_mcontig, int b_kcontig,
                         int batch, long long a_bs, long long b_bs, long long c_bs, int engine, void* stream);

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

#ifdef __cplusplus
}
#endif
#endif /* CIM_EXP_H */
