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
#include <mutex>

namespace {

#ifndef CIM_SMALL_BK
#define CIM_SMALL_BK 32
#endif
constexpr int SBM = 64, SBN = 64, SBK = CIM_SMALL_BK, SNT = 256;
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
    const float* colbias;    // [N] added per output COLUMN (nn.Linear bias: C = X . W^T + b); null for the convolutions
    unsigned* tile_cnt;      // split-K: per-tile arrival counters (zero between launches) - the last split of a tile combines
    // data gradients only: the BatchNorm + ReLU backward of the layer that PRODUCED this product's input, applied to the
    // result - C(row, col) = mask(row, col) > 0 ? C * mgamma[row] rsqrt(mvar[row] + meps) : 0 with mask = that layer's output
    // (this layer's input, same layout as C): the producer's own backward then starts from the gradient of its convolution
    const float* mask; const float* mgamma; const float* mvar; float meps;
};
struct InputBn { const float* y; const float* gamma; const float* var; float eps; };       // (the four fields above as arguments)

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

// Split-K combine.  0 (default): a separate reduce launch per split-K product (84 launches per step at cfg2).
// 1 / 2: combined INSIDE the launch by the tile's last arriving workgroup (below) - built and measured in round 3 because
// it removes those launches, and kept as an experiment only: whole step at cfg2 (same box, interleaved) 14.87 ms with the
// separate reduce, 17.92 ms with variant 1 (agent-scope release fence per workgroup + acquire in the last arrival:
// every fence writes back / invalidates the XCD's L2 under ~500 workgroups per launch), 16.41 ms with variant 2 (sc1
// partial stores / loads, no fences).  The weight-gradient products run 16-64 splits: one workgroup then reads 0.25-1 MB of
// partial tiles serially where the reduce launch spreads them over the chip.
#ifndef CIM_SMALL_FUSED_REDUCE
#define CIM_SMALL_FUSED_REDUCE 0
#endif
// Epilogue of one output value (shared by the single-pass kernels, the in-kernel split-K combine and the reduce kernel)
__device__ __forceinline__ void small_finish(const SmallArgs& g, int row, int col, float v) {
    const size_t o = (size_t)row * g.ldc + col;
    if (g.Xraw) g.Xraw[o] = v;
    float y = v;
    if (g.bn) {
        const float a = g.gamma[row] * rsqrtf(g.var[row] + g.eps);
        y = fmaf(v, a, g.beta[row] - g.mean[row] * a);
    }
    if (g.res) y += g.res[o];
    if (g.colbias) y += g.colbias[col];
    if (g.relu) y = fmaxf(y, 0.0f);
    if (g.mask) y = g.mask[o] > 0.0f ? y * (g.mgamma[row] * rsqrtf(g.mvar[row] + g.meps)) : 0.0f;
    g.C[o] = y;
}

// Split-K inside the launch: every split stores its partial tile to the workspace and takes a ticket on the tile's counter;
// the LAST arrival sums the tile's partials in split order (the order the reduce kernel used: same bits) and applies the
// epilogue.  Publication follows the chip's rules (the 8 XCDs' L2s are not coherent): plain stores, every wave drains
// vmcnt, barrier, ONE agent-scope release + relaxed agent-scope fetch_add by one lane; the last arrival does one
// agent-scope acquire, then plain loads.  The counter is left at zero for the next launch that uses the slot.
// `flag` = a word of the kernel's own LDS array (a second __shared__ object would de-pipeline the k-loop).
// CIM_SMALL_FUSED_REDUCE = 2: the partial tiles travel past the (per-XCD, non-coherent) L2s by themselves - agent-scope relaxed
// atomic stores / loads compile to sc1 accesses - so the hand-off needs no cache write-back / invalidate fences at all.
__device__ __forceinline__ void small_put_partial(float* p, float v) {
#if CIM_SMALL_FUSED_REDUCE == 2
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    *p = v;
#endif
}
__device__ __forceinline__ float small_get_partial(const float* p) {
#if CIM_SMALL_FUSED_REDUCE == 2
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    return *p;
#endif
}

template <int TN>
__device__ __forceinline__ void small_splitk_combine(const SmallArgs& g, int tile, int m0, int n0, int tid, int nthreads,
                                                     volatile int* flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
#if CIM_SMALL_FUSED_REDUCE != 2
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        const unsigned t = __hip_atomic_fetch_add(g.tile_cnt + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = t == (unsigned)g.splits - 1u;
        if (last) {
            __hip_atomic_store(g.tile_cnt + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#if CIM_SMALL_FUSED_REDUCE != 2
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
        }
        *flag = last;
    }
    __syncthreads();
    if (!*flag) return;
    const size_t mn = (size_t)g.M * g.N;
    for (int e = tid; e < SBM * TN; e += nthreads) {
        const int r = e / TN, c = e - r * TN;
        const int row = m0 + r, col = n0 + c;
        if (row >= g.M || col >= g.N) continue;
        const size_t i = (size_t)row * g.N + col;
        float v = small_get_partial(g.ws + i);
        for (int k = 1; k < g.splits; ++k) v += small_get_partial(g.ws + (size_t)k * mn + i);
        small_finish(g, row, col, v);
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
    if (col < g.N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
            if (row >= g.M) continue;
            if (g.splits > 1) small_put_partial(g.ws + ((size_t)split * g.M + row) * g.N + col, acc[r]);
            else small_finish(g, row, col, acc[r]);
        }
    }
    if (g.splits > 1 && g.tile_cnt != nullptr)
        small_splitk_combine<BNT>(g, tile, m0, n0, tid, NT, reinterpret_cast<volatile int*>(&As[0][0]));
}

// split-K: sum of the partial products in a fixed order (deterministic) + the same epilogue as the single-pass kernel
__global__ __launch_bounds__(256) void small_splitk_reduce_kernel(const SmallArgs g) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t mn = (size_t)g.M * g.N;
    if (i >= mn) return;
    float v = g.ws[i];
    for (int k = 1; k < g.splits; ++k) v += g.ws[(size_t)k * mn + i];
    small_finish(g, (int)(i / g.N), (int)(i % g.N), v);
}

// ------------------------------------------------------------------------------------------------------------------
// 3 x 3 convolutions (padding 1, stride 1 or 2) of the bottlenecks as IMPLICIT GEMMs on the same tile, MFMA loop and
// epilogue: the im2col matrix is never built, the B loader (and, for the data gradient, the A loader) computes the
// shifted / strided address and the zero-padding predicate per element.  In NCHW, one image, X [Cin][H][W] -> Y
// [Cout][Ho][Wo], W [Cout][Cin][3][3] (the module's own tensor, no repacking), k-order (channel, tap):
//   forward          M = Cout, N = Ho Wo, K = 9 Cin   A = W as [Cout][9 Cin] (K-contiguous)    B(k, n) = X[ci][s yo + dy][s xo + dx]
//   data gradient    M = Cin,  N = H W,   K = 9 Cout  A(m, k) = W[co][m][tap], from a transposed copy  B(k, n) = dY[co][(yi - dy) / s][(xi - dx) / s]
//   weight gradient  M = Cout, N = 9 Cin, K = Ho Wo   A = dY as [Cout][Ho Wo] (K-contiguous)   B(k, n) = X[ci][s yo + dy][s xo + dx]
// (tap = 3 (dy + 1) + (dx + 1); the weight gradient's C is the weight tensor's own layout).  Gathered elements are 4-byte
// loads, 8 per thread and slab, issued together; interior float4 pieces of stride-1 convolutions take one 16-byte load.
enum { CONV_FWD = 0, CONV_DX = 1, CONV_DW = 2 };
struct ConvGeom { int H, W, Ho, Wo, stride, cin, cout; float inv_w; int dil; };      // dil: dilation (padding = dil * (k / 2): "same")

// element (k, n) of the implicit B operand; `src` = X (forward, weight gradient) or dY (data gradient)
template <int MODE, int KS>
__device__ __forceinline__ bool conv_src(const ConvGeom& c, int ch, int tap, int y, int x, int& off) {
    const int dy = (tap / KS - KS / 2) * c.dil, dx = (tap - (tap / KS) * KS - KS / 2) * c.dil;      // padding dil * (KS / 2)
    if (MODE == CONV_DX) {                 // (y, x) = input pixel; the output pixel that reads it through this tap
        const int ty = y - dy, tx = x - dx;
        if (c.stride == 2 && ((ty | tx) & 1)) return false;
        const int yo = c.stride == 2 ? ty >> 1 : ty, xo = c.stride == 2 ? tx >> 1 : tx;
        off = (ch * c.Ho + yo) * c.Wo + xo;
        return ty >= 0 && tx >= 0 && yo < c.Ho && xo < c.Wo;
    }
    const int yi = y * c.stride + dy, xi = x * c.stride + dx;   // (y, x) = output pixel
    off = (ch * c.H + yi) * c.W + xi;
    return (unsigned)yi < (unsigned)c.H && (unsigned)xi < (unsigned)c.W;
}

#ifndef CIM_CONV3_BK
#define CIM_CONV3_BK 32            // slab depth of the 3 x 3 kernel (K = 9 C is long: 64 halves the barriers and load round trips per k)
#endif
constexpr int CBK = CIM_CONV3_BK;
template <int MODE, int KS = 3>
__global__ __launch_bounds__(256) void conv3x3_small_kernel(const SmallArgs g, const ConvGeom c) {
    constexpr int TAPS = KS * KS;
    constexpr int NT = 256, PA = SBM * (CBK / 4) / NT, PB = SBN * (CBK / 4) / NT;        // sixteen-byte pieces per thread and slab
    extern __shared__ __attribute__((aligned(16))) float c3_smem[];
    float (*As)[CBK * SLD] = reinterpret_cast<float (*)[CBK * SLD]>(c3_smem);
    float (*Bs)[CBK * SLD] = reinterpret_cast<float (*)[CBK * SLD]>(c3_smem + 2 * CBK * SLD);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (g.N + SBN - 1) / SBN;
    const int m0 = (blockIdx.x / tiles_n) * SBM, n0 = (blockIdx.x % tiles_n) * SBN;
    const int split = blockIdx.y;
    const int kper = ((g.K + g.splits - 1) / g.splits + CBK - 1) / CBK * CBK;
    const int kbeg = split * kper, kend = min(g.K, kbeg + kper);
    const int wsrc = (MODE == CONV_DX) ? c.Wo : c.W;                     // row length of the gathered tensor
    (void)wsrc;

    // ---- per-thread constants of the B pieces
    // forward / data gradient: B is row-contiguous in n (pixels): piece p -> k = k0 + p / 16, n = n0 + (p % 16) * 4: the
    // four pixels are fixed for the whole K loop, (channel, tap) changes per slab.
    // weight gradient: B is K-contiguous: piece p -> n = n0 + p % 64 (a (channel, tap) pair, fixed), k = k0 + (p / 64) * 4 pixels.
    int py[PB][4], px[PB][4];            // forward / dX: pixel coordinates of the piece's 4 columns (-1: column >= N)
    int bch[PB], btap[PB];               // dW: (channel, tap) of the piece's row (-1: row >= N)
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        const int p = tid + i * NT;
        if (MODE == CONV_DW) {
            const int n = n0 + p % SBN;
            bch[i] = n < g.N ? n / TAPS : -1;
            btap[i] = n - (n / TAPS) * TAPS;
        } else {
            const int wrow = (MODE == CONV_DX) ? c.W : c.Wo;             // n runs over input pixels (dX) / output pixels (forward)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + (p % 16) * 4 + j;
                const int y = (int)(((float)n + 0.5f) * c.inv_w);       // n / wrow (exact: see the launcher)
                py[i][j] = n < g.N ? y : -1;
                px[i][j] = n - y * wrow;
            }
        }
    }

    auto load_b = [&](int k0, float4 (&rb)[PB]) {
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const int p = tid + i * NT;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (MODE == CONV_DW) {
                const int k = k0 + (p / SBN) * 4;                        // 4 consecutive output pixels
                if (bch[i] >= 0) {
                    int off[4];
                    bool ok[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int po = k + j;
                        const int yo = (int)(((float)po + 0.5f) * c.inv_w), xo = po - yo * c.Wo;
                        ok[j] = po < kend && conv_src<MODE, KS>(c, bch[i], btap[i], yo, xo, off[j]);
                    }
                    if (c.stride == 1 && ok[0] && ok[1] && ok[2] && ok[3] && off[3] == off[0] + 3) {     // all inside: 16-byte load
                        const float4 t = ld4(g.B + off[0]);
                        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (ok[j]) v[j] = g.B[off[j]];
                    }
                }
            } else {
                const int k = k0 + p / 16;
                if (k < kend) {
                    const int ch = k / TAPS, tap = k - ch * TAPS;
                    int off[4];
                    bool ok[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) ok[j] = py[i][j] >= 0 && conv_src<MODE, KS>(c, ch, tap, py[i][j], px[i][j], off[j]);
                    if (c.stride == 1 && ok[0] && ok[1] && ok[2] && ok[3] && off[3] == off[0] + 3) {     // all inside: 16-byte load
                        const float4 t = ld4(g.B + off[0]);
                        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (ok[j]) v[j] = g.B[off[j]];
                    }
                }
            }
            rb[i] = make_float4(v[0], v[1], v[2], v[3]);
        }
    };
    auto load_a = [&](int k0, float4 (&ra)[PA]) {
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int p = tid + i * NT;
            if (MODE == CONV_DX) {       // A = the weight transposed to [(co, tap)][ci] by conv3x3_wt_kernel: M-contiguous rows
                ra[i] = tile_load<true, SBM>(g.A, g.lda, g.M, kend, m0, k0, p);
            } else {
                ra[i] = tile_load<false, SBM>(g.A, g.lda, g.M, kend, m0, k0, p);
            }
        }
    };

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    float4 ra0[PA], rb0[PB];
#define C3_PUT(BUF)                                                                                                 \
    {                                                                                                               \
        _Pragma("unroll") for (int i = 0; i < PA; ++i) tile_store<MODE == CONV_DX, SBM>(As[BUF], ra0[i], tid + i * NT); \
        _Pragma("unroll") for (int i = 0; i < PB; ++i) tile_store<MODE != CONV_DW, SBN>(Bs[BUF], rb0[i], tid + i * NT); \
    }
#define C3_MMA(BUF)                                                                                                 \
    {                                                                                                               \
        const float* __restrict__ a = As[BUF] + (lane >> 5) * SLD + wm * 32 + (lane & 31);                          \
        const float* __restrict__ b = Bs[BUF] + (lane >> 5) * SLD + wn * 32 + (lane & 31);                          \
        _Pragma("unroll") for (int kk = 0; kk < CBK; kk += 2)                                                       \
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk * SLD], b[kk * SLD], acc, 0, 0, 0);                     \
    }
    if (kbeg < kend) {
        const int nslab = (kend - kbeg + CBK - 1) / CBK;
        load_a(kbeg, ra0);
        load_b(kbeg, rb0);
        C3_PUT(0)
        __syncthreads();
        for (int s = 0; s < nslab; ++s) {
            const int more = s + 1 < nslab;
            if (more) { load_a(kbeg + (s + 1) * CBK, ra0); load_b(kbeg + (s + 1) * CBK, rb0); }
            if (s & 1) { C3_MMA(1) } else { C3_MMA(0) }
            if (more) { if (s & 1) { C3_PUT(0) } else { C3_PUT(1) } }
            __syncthreads();
        }
    }
#undef C3_PUT
#undef C3_MMA
    // ---- epilogue (as gemm_small_kernel)
    const int col = n0 + wn * 32 + (lane & 31);
    if (col < g.N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
            if (row >= g.M) continue;
            if (g.splits > 1) small_put_partial(g.ws + ((size_t)split * g.M + row) * g.N + col, acc[r]);
            else small_finish(g, row, col, acc[r]);
        }
    }
    if (g.splits > 1 && g.tile_cnt != nullptr)
        small_splitk_combine<SBN>(g, (int)blockIdx.x, m0, n0, tid, NT, reinterpret_cast<volatile int*>(c3_smem));
}

// w [Cout][Cin][9] -> wt [Cout][9][Cin] (the data gradient's A operand, M-contiguous): one workgroup per output channel,
// through LDS so that both sides are coalesced.  (Gathering W[co][ci][tap] in the GEMM's loader - 36-byte strides between
// lanes - made the data gradient 1.7x slower than the forward: 69 vs 40 us.)
__global__ __launch_bounds__(256) void conv3x3_wt_kernel(const float* __restrict__ w, float* __restrict__ wt, int cin) {
    extern __shared__ float wt_s[];
    const int co = blockIdx.x, n = cin * 9;
    for (int i = threadIdx.x; i < n; i += 256) wt_s[i] = w[(size_t)co * n + i];
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 256) {
        const int tap = i / cin, ci = i - tap * cin;
        wt[(size_t)co * n + i] = wt_s[ci * 9 + tap];
    }
}

// Arrival counters of the in-kernel split-K combine: a ring of zeroed words per device, allocated on first use and never
// freed (4 MiB).  A launch takes the next `tiles` words; every tile's last arrival puts its word back to zero, so a slot is
// reusable once its launch has finished - with 2^20 words and at most a few thousand tiles per launch, launches that are in
// flight together (two streams, a few dozen queued kernels) never share a word.
[[maybe_unused]] constexpr long long CNT_RING = 1ll << 20;
unsigned* splitk_counters(long long tiles) {
#if !CIM_SMALL_FUSED_REDUCE
    return nullptr;
#else
    static std::mutex mu;
    static unsigned* ring[64] = {nullptr};
    static long long next[64] = {0};
    if (tiles > CNT_RING / 4) return nullptr;          // (an oversize launch keeps the separate reduce pass)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (ring[dev] == nullptr) {
        if (hipMalloc(reinterpret_cast<void**>(&ring[dev]), sizeof(unsigned) * CNT_RING) != hipSuccess) { ring[dev] = nullptr; return nullptr; }
        if (hipMemset(ring[dev], 0, sizeof(unsigned) * CNT_RING) != hipSuccess) return nullptr;       // (synchronous: before any launch uses it)
    }
    if (next[dev] + tiles > CNT_RING) next[dev] = 0;
    unsigned* p = ring[dev] + next[dev];
    next[dev] += tiles;
    return p;
#endif
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
    // (sweep, tools/bench_gemm_small.py, us over 7 layer shapes forward / dX / dW: 157 / 136 / 182 with these limits; 2 slabs
    // per workgroup or up to 768-1024 workgroups: 188 / 178 / 219 - more partial products than the latency chain gains)
    static const int min_k = getenv("CIM_SMALL_MINK") ? atoi(getenv("CIM_SMALL_MINK")) : 4 * SBK;      // sweep switches
    static const int want = getenv("CIM_SMALL_WGS") ? atoi(getenv("CIM_SMALL_WGS")) : 512;
    int s = 1;
    while (tiles * s < want && K / (s * 2) >= min_k && s < 64) s *= 2;
    return s;
}

static int gemm_small_impl(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                           int a_mcontig, int b_kcontig, float* x_raw, const float* gamma, const float* beta,
                           const float* mean, const float* var, float eps, const float* residual, int relu, int splits,
                           float* workspace, void* stream, const InputBn* in_bn);

extern "C" int cim_gemm_small_f32(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                                  int a_mcontig, int b_kcontig, float* x_raw, const float* gamma, const float* beta,
                                  const float* mean, const float* var, float eps, const float* residual, int relu, int splits,
                                  float* workspace, void* stream) {
    return gemm_small_impl(A, B, C, M, N, K, lda, ldb, ldc, a_mcontig, b_kcontig, x_raw, gamma, beta, mean, var, eps, residual, relu,
                           splits, workspace, stream, nullptr);
}

static int gemm_small_impl(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                           int a_mcontig, int b_kcontig, float* x_raw, const float* gamma, const float* beta,
                           const float* mean, const float* var, float eps, const float* residual, int relu, int splits,
                           float* workspace, void* stream, const InputBn* in_bn) {
    CIM_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0 && lda > 0 && ldb > 0 && ldc >= N && splits >= 1);
    CIM_CHECK_ARG((gamma == nullptr) == (beta == nullptr) && (gamma == nullptr) == (mean == nullptr) && (gamma == nullptr) == (var == nullptr));
    CIM_CHECK_ARG(splits == 1 || workspace);
    SmallArgs g{};
    g.A = A; g.B = B; g.C = C; g.Xraw = x_raw;
    g.gamma = gamma; g.beta = beta; g.mean = mean; g.var = var; g.res = residual; g.eps = eps;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.a_mcontig = a_mcontig; g.b_kcontig = b_kcontig; g.relu = relu; g.bn = gamma != nullptr;
    g.splits = splits; g.ws = workspace; g.colbias = nullptr;
    if (in_bn) { g.mask = in_bn->y; g.mgamma = in_bn->gamma; g.mvar = in_bn->var; g.meps = in_bn->eps; }
    const long long tiles = (long long)((M + SBM - 1) / SBM) * ((N + SBN - 1) / SBN);
    CIM_CHECK_ARG(tiles * 2 < (1ll << 31) && splits <= 65535);
    const bool narrow = tiles * splits < 128;           // 64 x 32 tiles only for problems that cannot fill the chip otherwise
    g.tile_cnt = splits > 1 ? splitk_counters(narrow ? (long long)((M + SBM - 1) / SBM) * ((N + 31) / 32) : tiles) : nullptr;
    hipStream_t st = cim::as_stream(stream);
    if (a_mcontig) {
        if (b_kcontig) launch_small<true, true>(g, splits, st, narrow); else launch_small<true, false>(g, splits, st, narrow);
    } else {
        if (b_kcontig) launch_small<false, true>(g, splits, st, narrow); else launch_small<false, false>(g, splits, st, narrow);
    }
    if (splits > 1 && g.tile_cnt == nullptr) {          // (no counters: the separate, equally ordered reduce pass)
        const size_t n = (size_t)M * N;
        hipLaunchKernelGGL(small_splitk_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, g);
    }
    CIM_CHECK_LAUNCH();
    return 0;
}

// Y[M][N] = X[M][K] . W[N][K]^T + bias[N]  (nn.Linear on the small-tile fp32-MFMA GEMM): the eight scoring heads of
// lib/modeling/heads.py:194-219 as ONE product against their concatenated weights (N = 8 (C + 1) = 168 / 648 columns).
extern "C" int cim_linear_bias_f32(const float* X, const float* W, const float* bias, float* Y, int M, int N, int K, int splits,
                                   float* workspace, void* stream) {
    CIM_CHECK_ARG(X && W && Y && M > 0 && N > 0 && K > 0 && splits >= 1 && splits <= 65535 && (splits == 1 || workspace));
    SmallArgs g{};
    g.A = X; g.B = W; g.C = Y; g.Xraw = nullptr;
    g.gamma = g.beta = g.mean = g.var = g.res = nullptr; g.eps = 0.f;
    g.M = M; g.N = N; g.K = K; g.lda = K; g.ldb = K; g.ldc = N;
    g.a_mcontig = 0; g.b_kcontig = 1; g.relu = 0; g.bn = 0;
    g.splits = splits; g.ws = workspace; g.colbias = bias;
    const long long tiles = (long long)((M + SBM - 1) / SBM) * ((N + SBN - 1) / SBN);
    g.tile_cnt = splits > 1 ? splitk_counters(tiles) : nullptr;
    hipStream_t st = cim::as_stream(stream);
    launch_small<false, true>(g, splits, st, false);
    if (splits > 1 && g.tile_cnt == nullptr) {
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
    return (long long)sizeof(float) * (dconv + dx + dw);      // (dX and dW run side by side: separate split-K areas)
}

namespace {
// Fork / join between the caller's stream and a side stream inside one host call: the weight-gradient GEMM of a layer
// runs on the side stream next to the data-gradient GEMM (both only read the BatchNorm backward's output; each fills a
// fraction of the chip - they are latency bound).  Events come from a small round-robin pool (recorded and waited on at
// once, so reuse is safe).
struct ForkJoin {
    hipStream_t main, side;
    hipEvent_t ev[2];
    bool on;
    ForkJoin(hipStream_t m, hipStream_t s, bool want) : main(m), side(s), on(want && s != nullptr && s != m) {
        static std::mutex mu;
        static hipEvent_t pool[64];
        static int next = -1;            // -1: pool not created yet
        if (!on) return;
        std::lock_guard<std::mutex> lock(mu);
        if (next < 0) {
            for (int i = 0; i < 64; ++i)
                if (hipEventCreateWithFlags(&pool[i], hipEventDisableTiming) != hipSuccess) { on = false; return; }
            next = 0;
        }
        ev[0] = pool[next];
        ev[1] = pool[next + 1];
        next = (next + 2) % 64;
    }
    hipStream_t fork() {                 // side stream, ordered after everything enqueued on main so far
        if (!on) return main;
        (void)hipEventRecord(ev[0], main);
        (void)hipStreamWaitEvent(side, ev[0], 0);
        return side;
    }
    void join() {                        // main waits for the side stream's work
        if (!on) return;
        (void)hipEventRecord(ev[1], side);
        (void)hipStreamWaitEvent(main, ev[1], 0);
    }
};

__global__ __launch_bounds__(256) void small_axpy_kernel(float* __restrict__ y, const float* __restrict__ x, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] += x[i];
}
}  // namespace

extern "C" int cim_conv1x1_bn_act_bwd(const float* dy, const float* y, const float* x_raw, const float* x, const float* w,
                                      const float* gamma, const float* mean, const float* var, float eps, int relu,
                                      float* dres, float* dgamma, float* dbeta, float* dx, float* dw, int B, int cin, int cout,
                                      int hw, float* workspace, void* stream, void* side_stream, int join,
                                      int dy_is_dconv, const float* in_gamma, const float* in_var, float in_eps) {
    CIM_CHECK_ARG(dy && x_raw && x && w && gamma && mean && var && workspace && B > 0 && cin > 0 && cout > 0 && hw > 0);
    CIM_CHECK_ARG((y != nullptr || !relu) && ((dgamma == nullptr) == (dbeta == nullptr)));
    CIM_CHECK_ARG(!dy_is_dconv || (dres == nullptr && dgamma == nullptr));       // (the consumer's data gradient already applied this layer's BatchNorm + ReLU backward)
    CIM_CHECK_ARG((in_gamma == nullptr) == (in_var == nullptr));
    float* dconv = dy_is_dconv ? const_cast<float*>(dy) : workspace;             // [B][cout][hw]: dz * a, the gradient of the convolution output
    float* ws_dx = workspace + (size_t)B * cout * hw;
    float* ws_dw = ws_dx + (size_t)cim_gemm_small_splits(cin, hw, cout) * cin * hw;
    int rc = dy_is_dconv ? 0 : cim_bn_act_bwd(dy, y, x_raw, gamma, mean, var, eps, dconv, dres, dgamma, dbeta, B, cout, hw, relu, stream);
    if (rc) return rc;
    ForkJoin fj(cim::as_stream(stream), cim::as_stream(side_stream), dx && dw);
    void* st_dw = fj.fork();                                   // the weight gradient next to the data gradient
    for (int b = 0; b < B && dw; ++b) {                        // dW[cout, cin] = dconv . X^T  (K = hw)
        const int sp = cim_gemm_small_splits(cout, cin, hw);
        float* out = b == 0 ? dw : ws_dw + (size_t)sp * cout * cin;
        rc = cim_gemm_small_f32(dconv + (size_t)b * cout * hw, x + (size_t)b * cin * hw, out, cout, cin, hw, hw, hw, cin, 0, 1,
                                nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, nullptr, 0, sp, ws_dw, st_dw);
        if (rc) return rc;
        if (b) {
            const size_t n = (size_t)cout * cin;
            hipLaunchKernelGGL(small_axpy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, cim::as_stream(st_dw), dw, out, n);
        }
    }
    for (int b = 0; b < B && dx; ++b) {                        // dX[cin, hw] = W^T . dconv  (x the BatchNorm + ReLU backward of the layer that made x)
        const InputBn ib{x + (size_t)b * cin * hw, in_gamma, in_var, in_eps};
        rc = gemm_small_impl(w, dconv + (size_t)b * cout * hw, dx + (size_t)b * cin * hw, cin, hw, cout, cin, hw, hw, 1, 0,
                             nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, nullptr, 0, cim_gemm_small_splits(cin, hw, cout), ws_dx, stream,
                             in_gamma ? &ib : nullptr);
        if (rc) return rc;
    }
    if (join) fj.join();                                       // else the caller joins the side stream before the weight gradient is used
    CIM_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// 3 x 3 convolution (padding 1, stride 1 / 2, no bias) + BatchNorm (eval statistics) (+ residual) (+ ReLU), one image per
// launch; see conv3x3_small_kernel.
namespace {
int conv3x3_launch(int mode, const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldc, const ConvGeom& c,
                   float* x_raw, const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                   const float* residual, int relu, int splits, float* ws, hipStream_t st, int ksize = 3, const InputBn* in_bn = nullptr) {
    SmallArgs g{};
    g.A = A; g.B = B; g.C = C; g.Xraw = x_raw;
    g.gamma = gamma; g.beta = beta; g.mean = mean; g.var = var; g.res = residual; g.eps = eps;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = 0; g.ldc = ldc;
    g.a_mcontig = 0; g.b_kcontig = mode == CONV_DW; g.relu = relu; g.bn = gamma != nullptr;
    g.splits = splits; g.ws = ws; g.colbias = nullptr;
    if (in_bn) { g.mask = in_bn->y; g.mgamma = in_bn->gamma; g.mvar = in_bn->var; g.meps = in_bn->eps; }
    const dim3 grid((unsigned)(((M + SBM - 1) / SBM) * ((N + SBN - 1) / SBN)), (unsigned)splits);
    g.tile_cnt = splits > 1 ? splitk_counters((long long)grid.x) : nullptr;
    const size_t lds = sizeof(float) * 4 * CBK * SLD;
    auto kern = ksize == 7 ? conv3x3_small_kernel<CONV_FWD, 7>
                : mode == CONV_FWD ? conv3x3_small_kernel<CONV_FWD> : mode == CONV_DX ? conv3x3_small_kernel<CONV_DX> : conv3x3_small_kernel<CONV_DW>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, g, c);
    if (splits > 1 && g.tile_cnt == nullptr) {
        const size_t n = (size_t)M * N;
        hipLaunchKernelGGL(small_splitk_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, g);
    }
    return 0;
}
ConvGeom conv_geom(int cin, int cout, int H, int W, int stride, int mode, int dil = 1) {
    ConvGeom c;
    c.H = H; c.W = W; c.stride = stride; c.cin = cin; c.cout = cout; c.dil = dil;
    c.Ho = (H - 1) / stride + 1; c.Wo = (W - 1) / stride + 1;            // odd kernel k, padding k / 2
    c.inv_w = 1.0f / (float)(mode == CONV_DX ? W : c.Wo);
    return c;
}
}  // namespace

#define CONV3_ARGS_OK(CIN_MULT)                                                                                      \
    CIM_CHECK_ARG(cin > 0 && cout > 0 && H > 0 && W > 0 && (stride == 1 || stride == 2) && W <= 4096 &&               \
                  dilation >= 1 && dilation <= 8 && (dilation == 1 || stride == 1) &&                                 \
                  (long long)H * W < (1ll << 20) && (long long)cin * H * W < (1ll << 31) && (long long)cout * H * W < (1ll << 31) && \
                  (long long)cin * cout * 9 < (1ll << 31) && cin % (CIN_MULT) == 0)

extern "C" int cim_conv3x3_nchw_splits(int cin, int cout, int H, int W, int stride) {
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    return cim_gemm_small_splits(cout, Ho * Wo, 9 * cin);
}

extern "C" int cim_conv3x3_nchw_f32(const float* x, const float* w, float* y, int cin, int cout, int H, int W, int stride,
                               int dilation, float* x_raw, const float* gamma, const float* beta, const float* mean, const float* var,
                               float eps, const float* residual, int relu, int splits, float* workspace, void* stream) {
    CIM_CHECK_ARG(x && w && y && splits >= 1 && splits <= 65535 && (splits == 1 || workspace));
    CONV3_ARGS_OK(1);                 // (forward: any cin - the RGB stems of VGG16 / HRNet are frozen, forward only)
    CIM_CHECK_ARG((gamma == nullptr) == (beta == nullptr) && (gamma == nullptr) == (mean == nullptr) && (gamma == nullptr) == (var == nullptr));
    const ConvGeom c = conv_geom(cin, cout, H, W, stride, CONV_FWD, dilation);
    conv3x3_launch(CONV_FWD, w, x, y, cout, c.Ho * c.Wo, 9 * cin, 9 * cin, c.Ho * c.Wo, c, x_raw, gamma, beta, mean, var, eps,
                   residual, relu, splits, workspace, cim::as_stream(stream));
    CIM_CHECK_LAUNCH();
    return 0;
}

// The stem: 7 x 7 convolution (padding 3, stride 1 / 2, no bias, any cin) -> frozen BatchNorm (+ ReLU), forward only (the
// reference freezes it: FREEZE_AT >= 1); the same implicit GEMM with 49 taps (K = 49 cin; A = the weight as it is).
extern "C" int cim_conv7x7_nchw_f32(const float* x, const float* w, float* y, int cin, int cout, int H, int W, int stride,
                                    const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                                    int relu, void* stream) {
    CIM_CHECK_ARG(x && w && y && cin > 0 && cout > 0 && H > 0 && W > 0 && (stride == 1 || stride == 2) && W <= 4096);
    CIM_CHECK_ARG((long long)H * W < (1ll << 20) && (long long)cin * H * W < (1ll << 31) && (long long)cout * H * W < (1ll << 31));
    CIM_CHECK_ARG((gamma == nullptr) == (beta == nullptr) && (gamma == nullptr) == (mean == nullptr) && (gamma == nullptr) == (var == nullptr));
    const ConvGeom c = conv_geom(cin, cout, H, W, stride, CONV_FWD);
    conv3x3_launch(CONV_FWD, w, x, y, cout, c.Ho * c.Wo, 49 * cin, 49 * cin, c.Ho * c.Wo, c, nullptr, gamma, beta, mean, var, eps,
                   nullptr, relu, 1, nullptr, cim::as_stream(stream), 7);
    CIM_CHECK_LAUNCH();
    return 0;
}

extern "C" long long cim_conv3x3_nchw_bwd_workspace(int B, int cin, int cout, int H, int W, int stride) {
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const long long dx = (long long)cim_gemm_small_splits(cin, H * W, 9 * cout) * cin * H * W;
    const long long dw = (long long)cim_gemm_small_splits(cout, 9 * cin, Ho * Wo) * cout * cin * 9 + (B > 1 ? (long long)cout * cin * 9 : 0);
    const long long dconv = (long long)B * cout * Ho * Wo;
    return (long long)sizeof(float) * (dconv + (long long)cout * cin * 9 + dx + dw);
}

// The whole backward of conv3x3 -> BatchNorm (+ residual) (+ ReLU): BatchNorm / ReLU backward (bn_act.hip), data gradient
// and weight gradient implicit GEMMs with their split-K reduces, enqueued by one host call (as cim_conv1x1_bn_act_bwd).
extern "C" int cim_conv3x3_nchw_bn_act_bwd(const float* dy, const float* y, const float* x_raw, const float* x, const float* w,
                                      const float* gamma, const float* mean, const float* var, float eps, int relu,
                                      float* dres, float* dgamma, float* dbeta, float* dx, float* dw, int B, int cin, int cout,
                                      int H, int W, int stride, int dilation, float* workspace, void* stream, void* side_stream,
                                      int join, int dy_is_dconv, const float* in_gamma, const float* in_var, float in_eps) {
    CIM_CHECK_ARG(dy && x_raw && x && w && gamma && mean && var && workspace && B > 0);
    CONV3_ARGS_OK(4);
    CIM_CHECK_ARG((y != nullptr || !relu) && ((dgamma == nullptr) == (dbeta == nullptr)) && cout % 4 == 0);
    CIM_CHECK_ARG(!dy_is_dconv || (dres == nullptr && dgamma == nullptr));
    CIM_CHECK_ARG((in_gamma == nullptr) == (in_var == nullptr));
    const ConvGeom cx = conv_geom(cin, cout, H, W, stride, CONV_DX, dilation), cw = conv_geom(cin, cout, H, W, stride, CONV_DW, dilation);
    const int hwo = cx.Ho * cx.Wo, hw = H * W;
    hipStream_t st = cim::as_stream(stream);
    float* dconv = dy_is_dconv ? const_cast<float*>(dy) : workspace;      // [B][cout][Ho Wo]: the gradient of the convolution output
    float* wt = workspace + (size_t)B * cout * hwo;            // [cout][9][cin]
    float* ws_dx = wt + (size_t)cout * cin * 9;
    float* ws_dw = ws_dx + (size_t)cim_gemm_small_splits(cin, hw, 9 * cout) * cin * hw;
    CIM_CHECK_ARG((size_t)cin * 9 * sizeof(float) <= 64 * 1024);
    int rc = dy_is_dconv ? 0 : cim_bn_act_bwd(dy, y, x_raw, gamma, mean, var, eps, dconv, dres, dgamma, dbeta, B, cout, hwo, relu, stream);
    if (rc) return rc;
    ForkJoin fj(st, cim::as_stream(side_stream), dx && dw);
    hipStream_t st_dw = fj.fork();                             // the weight gradient next to the data gradient
    for (int b = 0; b < B && dw; ++b) {                        // dW[cout][cin 9] = dconv . im2col(X)^T  (K = Ho Wo)
        const int sp = cim_gemm_small_splits(cout, 9 * cin, hwo);
        float* out = b == 0 ? dw : ws_dw + (size_t)sp * cout * cin * 9;
        conv3x3_launch(CONV_DW, dconv + (size_t)b * cout * hwo, x + (size_t)b * cin * hw, out, cout, 9 * cin, hwo, hwo, 9 * cin, cw,
                       nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, nullptr, 0, sp, ws_dw, st_dw);
        if (b) {
            const size_t n = (size_t)cout * cin * 9;
            hipLaunchKernelGGL(small_axpy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st_dw, dw, out, n);
        }
    }
    if (dx) hipLaunchKernelGGL(conv3x3_wt_kernel, dim3(cout), dim3(256), sizeof(float) * cin * 9, st, w, wt, cin);
    for (int b = 0; b < B && dx; ++b) {                        // dX[cin][H W] = sum over (co, tap) W[co][ci][tap] dconv[co][shifted]
        const InputBn ib{x + (size_t)b * cin * hw, in_gamma, in_var, in_eps};
        conv3x3_launch(CONV_DX, wt, dconv + (size_t)b * cout * hwo, dx + (size_t)b * cin * hw, cin, hw, 9 * cout, cin, hw, cx, nullptr,
                       nullptr, nullptr, nullptr, nullptr, 0.f, nullptr, 0, cim_gemm_small_splits(cin, hw, 9 * cout), ws_dx, st, 3,
                       in_gamma ? &ib : nullptr);
    }
    if (join) fj.join();                                       // else the caller joins the side stream before the weight gradient is used
    CIM_CHECK_LAUNCH();
    return 0;
}
