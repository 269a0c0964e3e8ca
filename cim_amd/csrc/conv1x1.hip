// Backbone 1x1 convolutions as small-tile fp32-MFMA GEMMs with the BatchNorm / residual / ReLU chain in the epilogue.
//
// Replaces, for the ResNet-50 C4 body of /root/reference/lib/modeling/resnet50.py:17-91 (torchvision Bottleneck:
// conv1 / conv3 / downsample.0 are 1 x 1 convolutions, each followed by a BatchNorm kept in eval mode :53-77), the
// ATen -> MIOpen -> rocBLAS path those layers took (31 GEMM launches + 31 BatchNorm launches forward at cfg2, on
// 64 x 32 ... 128 x 128 Tensile tiles at ~47 TF forward / ~26 TF backward) by ONE launch per layer and direction.
//
// In NCHW a 1 x 1 convolution of one image is  Y[Cout][HW] = W[Cout][Cin] . X[Cin][HW]:
//   forward          A = W   (row-major, K-contiguous)        B = X  ([K][N], N-contiguous)
//   data gradient    A = W^T (element (m,k) at W[k*Cin + m])  B = dY ([K][N], N-contiguous)
//   weight gradient  A = dY  (K = HW contiguous)              B = X^T (element (k,n) at X[n*HW + k]: K-contiguous)
// The contraction sizes are small (M, K = 64 ... 1024, N = HW = 1.4k ... 22k), so the tile is 64 x 64 x 16 with four
// waves of one 32 x 32 MFMA tile each (v_mfma_f32_32x32x2_f32: true fp32 products, no operand splitting), k-major LDS
// slabs ([k][m]: a lane's MFMA operand is one conflict-free ds_read_b32), double buffered, the next slab's global loads
// in flight while the current one is multiplied; up to 8 workgroups per CU hide the rest.  Short-K / long-K products
// (weight gradients: K = HW) take split-K through a workspace + one reduce pass (no float atomics: they run at
// ~90 G/s on MI355X).
// Epilogue (forward): x = acc (stored when the backward needs the convolution output), y = relu?(x * a[m] + b[m] (+ res)),
// a = gamma * rsqrt(var + eps), b = beta - mean * a: the BatchNorm (+ identity) (+ ReLU) of the bottleneck costs no pass.
#include "common.h"
#include "../../include/cim_hip.h"

namespace {

constexpr int SBM = 64, SBN = 64, SBK = 32, SNT = 256;
constexpr int SLD = 68;                      // padded row stride of a k-major slab (floats); % 4 == 0 for b128 stores
typedef float f32x16 __attribute__((ext_vector_type(16)));
// 16-byte global loads from 4-byte aligned addresses: rows of an NCHW activation [C][HW] start wherever HW puts them
// (HW = 1419, 5590 at cfg2), and gfx9's global_load_dwordx4 only needs dword alignment (unaligned access mode is on under
// ROCm); the type tells the compiler not to assume more.  The first version fell back to four 4-byte loads per lane for
// such operands and ran 2x slower than the aligned one.
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ float4 ld4(const float* p) {
    const f32x4u v = *reinterpret_cast<const f32x4u*>(p);
    return make_float4(v.x, v.y, v.z, v.w);
}

struct SmallArgs {
    const float* A; const float* B; float* C; float* Xraw;
    const float* gamma; const float* beta; const float* mean; const float* var; const float* res;
    float eps;
    int M, N, K, lda, ldb, ldc;
    int a_mcontig, b_kcontig, relu, bn;
    int splits; float* ws;
};

// Tile loaders.  An operand tile is ROWS x 32 (k) floats per slab, moved as 16-byte pieces: piece index p ->
//   K-contiguous operand (element (r, k) at P[r*ld + k]):  r = p % ROWS, k = (p / ROWS) * 4   (lanes along the rows: the
//     transposing LDS stores are conflict-free; with lanes along k they were 4-way conflicted)
//   row-contiguous operand (element (r, k) at P[k*ld + r]): k = p / (ROWS/4), r = (p % (ROWS/4)) * 4
// LDS slabs are k-major ([k][row], stride SLD): row-contiguous pieces are one ds_write_b128, K-contiguous ones transpose.
template <bool RC, int ROWS>
__device__ __forceinline__ float4 tile_load(const float* __restrict__ P, int ld, int rows, int kend, int r0, int k0, int p) {
    int r, k;
    if (RC) { k = k0 + p / (ROWS / 4); r = r0 + (p % (ROWS / 4)) * 4; } else { r = r0 + p % ROWS; k = k0 + (p / ROWS) * 4; }
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (RC) {
        if (k < kend) {
            const float* q = P + (size_t)k * ld + r;
            if (r + 3 < rows) v = ld4(q);
            else { if (r < rows) v.x = q[0]; if (r + 1 < rows) v.y = q[1]; if (r + 2 < rows) v.z = q[2]; }
        }
    } else if (r < rows) {
        const float* q = P + (size_t)r * ld + k;
        if (k + 3 < kend) v = ld4(q);
        else { if (k < kend) v.x = q[0]; if (k + 1 < kend) v.y = q[1]; if (k + 2 < kend) v.z = q[2]; }
    }
    return v;
}

template <bool RC, int ROWS>
__device__ __forceinline__ void tile_store(float* __restrict__ S, float4 v, int p) {
    if (RC) {
        const int k = p / (ROWS / 4), r = (p % (ROWS / 4)) * 4;
        *reinterpret_cast<float4*>(S + k * SLD + r) = v;
    } else {
        const int r = p % ROWS, k = (p / ROWS) * 4;      // a wave's lanes hold consecutive rows: conflict-free transposing stores
        S[(k + 0) * SLD + r] = v.x;
        S[(k + 1) * SLD + r] = v.y;
        S[(k + 2) * SLD + r] = v.z;
        S[(k + 3) * SLD + r] = v.w;
    }
}

// WN = waves along N: 2 -> 64 x 64 tile, 256 threads; 1 -> 64 x 32 tile, 128 threads (twice the workgroups for the
// smallest problems).  Tried and measured slower (tools/bench_gemm_small.py, 7 layer shapes, us forward / dX / dW:
// 158 / 137 / 183 with this kernel): 128- and 256-row tiles (4 waves of 1 x 2 / 2 x 2 MFMA tiles, 134-170 VGPRs) to re-read
// the activation operand less often: 251 / 207 / 290 - these products are bound by the latency chain of a tile's few
// slabs and by how many tiles are in flight, not by operand traffic.
template <bool AM, bool BKc, int WN>
__global__ __launch_bounds__(128 * WN) void gemm_small_kernel(const SmallArgs g) {
    constexpr int NT = 128 * WN, BNT = 32 * WN;
    constexpr int PA = SBM * 8 / NT, PB = BNT * 8 / NT;          // 16-byte pieces per thread and slab: A 4 / 2, B 2 / 2
    __shared__ __attribute__((aligned(16))) float As[2][SBK * SLD];
    __shared__ __attribute__((aligned(16))) float Bs[2][SBK * SLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    // tile order: N fastest (neighbouring workgroups share the A = weight panel through L2)
    const int tiles_n = (g.N + BNT - 1) / BNT;
    const int tile = blockIdx.x;
    const int m0 = (tile / tiles_n) * SBM, n0 = (tile % tiles_n) * BNT;
    const int split = blockIdx.y;
    const int kper = ((g.K + g.splits - 1) / g.splits + SBK - 1) / SBK * SBK;      // (a multiple of the slab depth)
    const int kbeg = split * kper, kend = min(g.K, kbeg + kper);

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;

    float4 ra0[PA], rb0[PB];
#define SM_GLOAD(RA, RB, K0)                                                                                       \
    {                                                                                                              \
        _Pragma("unroll") for (int i = 0; i < PA; ++i) RA[i] = tile_load<AM, SBM>(g.A, g.lda, g.M, kend, m0, K0, tid + i * NT);   \
        _Pragma("unroll") for (int i = 0; i < PB; ++i) RB[i] = tile_load<!BKc, BNT>(g.B, g.ldb, g.N, kend, n0, K0, tid + i * NT); \
    }
#define SM_PUT(RA, RB, BUF)                                                                                        \
    {                                                                                                              \
        _Pragma("unroll") for (int i = 0; i < PA; ++i) tile_store<AM, SBM>(As[BUF], RA[i], tid + i * NT);           \
        _Pragma("unroll") for (int i = 0; i < PB; ++i) tile_store<!BKc, BNT>(Bs[BUF], RB[i], tid + i * NT);         \
    }
#define SM_MMA(BUF)                                                                                                \
    {                                                                                                              \
        const float* __restrict__ a = As[BUF] + (lane >> 5) * SLD + wm * 32 + (lane & 31);                         \
        const float* __restrict__ b = Bs[BUF] + (lane >> 5) * SLD + wn * 32 + (lane & 31);                         \
        _Pragma("unroll") for (int kk = 0; kk < SBK; kk += 2)                                                      \
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk * SLD], b[kk * SLD], acc, 0, 0, 0);                    \
    }
    if (kbeg < kend) {
        // the next slab's global loads are in flight while the current one is multiplied.  (Two slabs ahead in a second
        // register set measured slower: 100+ VGPRs; so did 64 x 32 tiles for everything below 1024 workgroups.)
        const int nslab = (kend - kbeg + SBK - 1) / SBK;
        SM_GLOAD(ra0, rb0, kbeg)
        SM_PUT(ra0, rb0, 0)
        __syncthreads();
        for (int s = 0; s < nslab; ++s) {
            const int more = s + 1 < nslab;
            if (more) SM_GLOAD(ra0, rb0, kbeg + (s + 1) * SBK)
            if (s & 1) { SM_MMA(1) } else { SM_MMA(0) }
            if (more) { if (s & 1) { SM_PUT(ra0, rb0, 0) } else { SM_PUT(ra0, rb0, 1) } }
            __syncthreads();
        }
    }
#undef SM_GLOAD
#undef SM_PUT
#undef SM_MMA

    // ---- epilogue: lane holds rows 8*(r/4) + 4*(lane/32) + r%4, column lane%32 of its wave's 32 x 32 tile
    const int col = n0 + wn * 32 + (lane & 31);
    if (col >= g.N) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
        if (row >= g.M) continue;
        const float v = acc[r];
        if (g.splits > 1) {
            g.ws[((size_t)split * g.M + row) * g.N + col] = v;
            continue;
        }
        const size_t o = (size_t)row * g.ldc + col;
        if (g.Xraw) g.Xraw[o] = v;
        float y = v;
        if (g.bn) {
            const float a = g.gamma[row] * rsqrtf(g.var[row] + g.eps);
            y = fmaf(v, a, g.beta[row] - g.mean[row] * a);
        }
        if (g.res) y += g.res[o];
        if (g.relu) y = fmaxf(y, 0.0f);
        g.C[o] = y;
    }
}

// split-K: sum of the partial products in a fixed order (deterministic) + the same epilogue as the single-pass kernel
__global__ __launch_bounds__(256) void small_splitk_reduce_kernel(const SmallArgs g) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t mn = (size_t)g.M * g.N;
    if (i >= mn) return;
    float v = g.ws[i];
    for (int k = 1; k < g.splits; ++k) v += g.ws[(size_t)k * mn + i];
    const int row = (int)(i / g.N);
    const size_t o = (size_t)row * g.ldc + (i % g.N);
    if (g.Xraw) g.Xraw[o] = v;
    float y = v;
    if (g.bn) {
        const float a = g.gamma[row] * rsqrtf(g.var[row] + g.eps);
        y = fmaf(v, a, g.beta[row] - g.mean[row] * a);
    }
    if (g.res) y += g.res[o];
    if (g.relu) y = fmaxf(y, 0.0f);
    g.C[o] = y;
}

template <bool AM, bool BKc>
static void launch_small(const SmallArgs& g, int splits, hipStream_t st, bool narrow) {
    const long long tm = (g.M + SBM - 1) / SBM;
    if (narrow)
        hipLaunchKernelGGL((gemm_small_kernel<AM, BKc, 1>), dim3((unsigned)(tm * ((g.N + 31) / 32)), (unsigned)splits), dim3(128), 0, st, g);
    else
        hipLaunchKernelGGL((gemm_small_kernel<AM, BKc, 2>), dim3((unsigned)(tm * ((g.N + 63) / 64)), (unsigned)splits), dim3(256), 0, st, g);
}

}  // namespace

extern "C" int cim_gemm_small_splits(int M, int N, int K) {
    // A workgroup walks its K range slab by slab, one global-load latency (~1.5 us) per 32-k slab when it is alone on its CU:
    // long-K products with few output tiles (res4 conv1: 92 tiles x 32 slabs = 41 us) are cut until there are ~2
    // workgroups per CU, never below 4 slabs per workgroup.
    const long long tiles = (long long)((M + SBM - 1) / SBM) * ((N + SBN - 1) / SBN);
    int s = 1;
    while (tiles * s < 512 && K / (s * 2) >= 4 * SBK && s < 64) s *= 2;
    return s;
}

extern "C" int cim_gemm_small_f32(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                                  int a_mcontig, int b_kcontig, float* x_raw, const float* gamma, const float* beta,
                                  const float* mean, const float* var, float eps, const float* residual, int relu, int splits,
                                  float* workspace, void* stream) {
    CIM_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0 && lda > 0 && ldb > 0 && ldc >= N && splits >= 1);
    CIM_CHECK_ARG((gamma == nullptr) == (beta == nullptr) && (gamma == nullptr) == (mean == nullptr) && (gamma == nullptr) == (var == nullptr));
    CIM_CHECK_ARG(splits == 1 || workspace);
    SmallArgs g;
    g.A = A; g.B = B; g.C = C; g.Xraw = x_raw;
    g.gamma = gamma; g.beta = beta; g.mean = mean; g.var = var; g.res = residual; g.eps = eps;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.a_mcontig = a_mcontig; g.b_kcontig = b_kcontig; g.relu = relu; g.bn = gamma != nullptr;
    g.splits = splits; g.ws = workspace;
    const long long tiles = (long long)((M + SBM - 1) / SBM) * ((N + SBN - 1) / SBN);
    CIM_CHECK_ARG(tiles * 2 < (1ll << 31) && splits <= 65535);
    const bool narrow = tiles * splits < 128;           // 64 x 32 tiles only for problems that cannot fill the chip otherwise
    hipStream_t st = cim::as_stream(stream);
    if (a_mcontig) {
        if (b_kcontig) launch_small<true, true>(g, splits, st, narrow); else launch_small<true, false>(g, splits, st, narrow);
    } else {
        if (b_kcontig) launch_small<false, true>(g, splits, st, narrow); else launch_small<false, false>(g, splits, st, narrow);
    }
    if (splits > 1) {
        const size_t n = (size_t)M * N;
        hipLaunchKernelGGL(small_splitk_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, g);
    }
    CIM_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// The whole backward of conv1x1 -> BatchNorm (+ residual) (+ ReLU) for a batch of images in ONE call: BatchNorm / ReLU
// backward (bn_act.hip), data gradient and weight gradient GEMMs with their split-K reduces - 3 to 5 launches enqueued by
// one host call instead of a dozen Python-level operations (the backbone backward was host-bound: 1.07 ms of idle GPU
// time in 3.6 ms at cfg2 with the operations issued one by one).
extern "C" long long cim_conv1x1_bwd_workspace(int B, int cin, int cout, int hw) {
    const long long dx = (long long)cim_gemm_small_splits(cin, hw, cout) * cin * hw;
    const long long dw = (long long)cim_gemm_small_splits(cout, cin, hw) * cout * cin + (B > 1 ? (long long)cout * cin : 0);
    const long long dconv = (long long)B * cout * hw;
    return (long long)sizeof(float) * (dconv + (dx > dw ? dx : dw));
}

namespace {
__global__ __launch_bounds__(256) void small_axpy_kernel(float* __restrict__ y, const float* __restrict__ x, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] += x[i];
}
}  // namespace

extern "C" int cim_conv1x1_bn_act_bwd(const float* dy, const float* y, const float* x_raw, const float* x, const float* w,
                                      const float* gamma, const float* mean, const float* var, float eps, int relu,
                                      float* dres, float* dgamma, float* dbeta, float* dx, float* dw, int B, int cin, int cout,
                                      int hw, float* workspace, void* stream) {
    CIM_CHECK_ARG(dy && x_raw && x && w && gamma && mean && var && workspace && B > 0 && cin > 0 && cout > 0 && hw > 0);
    CIM_CHECK_ARG((y != nullptr || !relu) && ((dgamma == nullptr) == (dbeta == nullptr)));
    float* dconv = workspace;                                  // [B][cout][hw]: dz * a, the gradient of the convolution output
    float* ws = workspace + (size_t)B * cout * hw;
    int rc = cim_bn_act_bwd(dy, y, x_raw, gamma, mean, var, eps, dconv, dres, dgamma, dbeta, B, cout, hw, relu, stream);
    if (rc) return rc;
    for (int b = 0; b < B && dx; ++b) {                        // dX[cin, hw] = W^T . dconv
        rc = cim_gemm_small_f32(w, dconv + (size_t)b * cout * hw, dx + (size_t)b * cin * hw, cin, hw, cout, cin, hw, hw, 1, 0,
                                nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, nullptr, 0, cim_gemm_small_splits(cin, hw, cout), ws, stream);
        if (rc) return rc;
    }
    for (int b = 0; b < B && dw; ++b) {                        // dW[cout, cin] = dconv . X^T  (K = hw)
        const int sp = cim_gemm_small_splits(cout, cin, hw);
        float* out = b == 0 ? dw : ws + (size_t)sp * cout * cin;
        rc = cim_gemm_small_f32(dconv + (size_t)b * cout * hw, x + (size_t)b * cin * hw, out, cout, cin, hw, hw, hw, cin, 0, 1,
                                nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, nullptr, 0, sp, ws, stream);
        if (rc) return rc;
        if (b) {
            const size_t n = (size_t)cout * cin;
            hipLaunchKernelGGL(small_axpy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, cim::as_stream(stream), dw, out, n);
        }
    }
    CIM_CHECK_LAUNCH();
    return 0;
}
